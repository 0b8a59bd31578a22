O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c3}; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "tn_multi or queued_weight or gemm_tn or mlp2 or relu_mlp or rowdot" > $O/tests_tn.log 2>&1; tail -3 $O/tests_tn.log
python -m pytest tests/test_gpu_headline.py tests/test_gpu_stack.py tests/test_gpu_model.py tests/test_gpu_param_pack.py -x -q > $O/tests_stack.log 2>&1; tail -3 $O/tests_stack.log
python tools/probes/pocket_hostprof2.py > $O/hostprof2.txt 2>&1; head -8 $O/hostprof2.txt
for i in 1 2; do
for d in 0 1; do
FABIND_TN_DEFER=$d python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket defer=$d', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_PREFETCH=1 FABIND_TN_DEFER=$d python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket prefetch defer=$d', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done; done
python -m pytest tests/test_gpu_bench.py -x -q > $O/tests_bench.log 2>&1; tail -5 $O/tests_bench.log
