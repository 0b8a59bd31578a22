# round 3, call 7: new unit tests, forward bench with the fused attention, FABind+ training, suite, default bench line, utilisation counters
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_attn_mfma.py tests/test_gpu_kernels.py -m gpu -q -s -x -k "fused_attention or layernorm_rows or dropout_under_autograd" > $O/tests_new.log 2>&1; tail -4 $O/tests_new.log; grep "fused attention" $O/tests_new.log
for f in 1 0; do
  FABIND_ATTN_FUSED=$f python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fwd fused=$f', round(d['value'],1), round(d['ms_per_step'],2))"
done
FABIND_BENCH_DUMP_PROFILE=$O/fwd_launch_groups.txt python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 5 > /dev/null 2>&1; grep cross_attn $O/fwd_launch_groups.txt
for e in 1 0; do FABIND_EPI_DROP_GRAD=$e python bench.py --no-cpu-baseline --mode plus_train 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plus_train epi_drop=$e', d['value'], d['ms_per_step'])"; done
python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -6 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]|split|fused" $O/tests.log > $O/tests_gaps.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/bench_default.json"))
print("headline", round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],4), d["roofline"].get("avg_us"))
for k,v in d.items():
    if isinstance(v,dict) and "value" in v and k not in ("roofline",): print(k, round(v["value"],1), v.get("unit"), round(v.get("ms_per_step",0),1), v.get("steps"))
    elif isinstance(v,dict) and "error" in v: print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1) > $O/fwd_pmc_util.txt 2>&1
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/pmc_u $O/profpt
cat $O/fwd_pmc_util.txt; head -28 $O/plus_train_kernel_stats.txt
