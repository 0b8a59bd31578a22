# saved-forward edge backward (fused_edge_bwd4.hip): kernel parity, stack parity, then the same-box A/B of the headline step
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c19}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_edge" > $O/tests_k.log 2>&1; tail -4 $O/tests_k.log
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_stack.py -x -q > $O/tests_s.log 2>&1; tail -4 $O/tests_s.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline saved-forward edge bwd', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_EDGE_SAVE_FWD=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline recompute edge bwd   ', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket saved-forward', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_EDGE_SAVE_FWD=0 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket recompute    ', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fb_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
rm -rf $O/prof
head -12 $O/fwdbwd_kernel_stats.txt | cut -c1-170
