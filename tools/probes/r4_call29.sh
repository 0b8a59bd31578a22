O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c29}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or norm" > $O/tests_ln.log 2>&1; tail -3 $O/tests_ln.log
timeout 1500 python -m pytest tests/test_gpu_plus.py tests/test_gpu_production.py -x -q > $O/tests_plus.log 2>&1; tail -3 $O/tests_plus.log
for i in 1 2; do
python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('plus_train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1; rm -rf $O/prof
grep "layernorm" $O/plus_train_kernel_stats.txt | cut -c1-170; head -1 $O/plus_train_kernel_stats.txt
