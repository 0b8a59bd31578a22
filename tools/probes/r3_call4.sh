# round 3, call 4: suite, the default bench line (all sub-objects), x3 step again, FABind+ training trace
O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -6 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]|split" $O/tests.log > $O/tests_gaps.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3d/bench_default.json"))
print("headline", round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],4))
for k,v in d.items():
    if isinstance(v,dict) and "value" in v and k not in ("roofline",): print(k, round(v["value"],1), v.get("unit"), round(v.get("ms_per_step",0),1), v.get("steps"))
    elif isinstance(v,dict) and "error" in v: print(k, v)
PY
python bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('x3', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_pt_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/profpt
head -36 $O/plus_train_kernel_stats.txt
