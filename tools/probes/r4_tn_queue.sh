# Round 4: the queued weight-gradient contractions (fabind_gemm_tn_multi) -- kernel tests, then same-box A/B of the pocket-sized and headline steps.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4tnq}; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "tn_multi or queued_weight or gemm_tn" > $O/tests_tn.log 2>&1; tail -3 $O/tests_tn.log
for i in 1 2; do
for d in 0 1; do
FABIND_TN_DEFER=$d python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket defer=$d', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done; done
for d in 0 1; do
FABIND_TN_DEFER=$d python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline defer=$d', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_TN_DEFER=$d python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model defer=$d', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python -m pytest tests/test_gpu_headline.py tests/test_gpu_stack.py tests/test_gpu_model.py -x -q > $O/tests_stack.log 2>&1; tail -3 $O/tests_stack.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
rm -rf $O/prof
head -12 $O/pocket_kernel_stats.txt
