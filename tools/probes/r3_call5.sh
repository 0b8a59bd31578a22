# round 3, call 5: fused cross attention (unit test, forward bench, matrix-core utilisation counters), whole suite, default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_attn_mfma.py -m gpu -q -s -x > $O/tests_attn.log 2>&1; tail -5 $O/tests_attn.log; grep "fused attention" $O/tests_attn.log
for f in 1 0; do
  FABIND_ATTN_FUSED=$f python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fwd fused=$f', round(d['value'],1), round(d['ms_per_step'],2))"
  FABIND_ATTN_FUSED=$f python bench.py --mode fwd --n-iter 8 --no-cpu-baseline --no-extras --steps 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fwd n_iter8 fused=$f', round(d['value'],1), round(d['ms_per_step'],2))"
done
FABIND_BENCH_DUMP_PROFILE=$O/fwd_launch_groups.txt python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 5 > /dev/null 2>&1; head -12 $O/fwd_launch_groups.txt
python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -6 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]|split|fused" $O/tests.log > $O/tests_gaps.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3e/bench_default.json"))
print("headline", round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],4))
for k,v in d.items():
    if isinstance(v,dict) and "value" in v and k not in ("roofline",): print(k, round(v["value"],1), v.get("unit"), round(v.get("ms_per_step",0),1), v.get("steps"))
    elif isinstance(v,dict) and "error" in v: print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/proffwd -o fwd -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --no-cpu-baseline --no-extras --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/proffwd/fwd_results.db > $O/fwd_kernel_stats.txt 2>&1
python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1) > $O/fwd_pmc_util.txt 2>&1
rm -rf $O/proffwd $O/pmc_u
head -24 $O/fwd_kernel_stats.txt; grep -i "cross_attn\|fused" $O/fwd_pmc_util.txt | head
