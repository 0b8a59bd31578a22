# bf16 residual epilogue + 32-padded narrow contractions + narrow-row LayerNorm: tests, FABind+ parity, plus_train (3 runs), kernel stats
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3l}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -6 > $O/tests_kernels.txt
python -m pytest tests/test_gpu_plus.py tests/test_gpu_production.py tests/test_gpu_plus_dense_api.py -q -m gpu -k "plus" 2>&1 | tail -8 > $O/tests_plus.txt
for i in 1 2 3; do
python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/profpt
FABIND_BENCH_DUMP_PROFILE=$O/pt_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
cat $O/tests_kernels.txt $O/tests_plus.txt $O/ab.txt; head -45 $O/plus_train_kernel_stats.txt | cut -c1-160
