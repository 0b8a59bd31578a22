O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_attn_mfma.py -m gpu -q -s -x -k "fused_attention" > $O/tests_new.log 2>&1; tail -3 $O/tests_new.log; grep "fused attention" $O/tests_new.log
for r in 1 2; do for f in 1 0; do
  FABIND_ATTN_FUSED=$f python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fwd round $r fused=$f', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
FABIND_BENCH_DUMP_PROFILE=$O/fwd_launch_groups.txt python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 5 > /dev/null 2>&1; grep cross_attn $O/fwd_launch_groups.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1) > $O/fwd_pmc_util.txt 2>&1
rm -rf $O/pmc_u
grep -i "cross_attn" $O/fwd_pmc_util.txt
