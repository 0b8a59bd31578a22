# (1) pair-bias finishing pass after the register-accumulation rewrite: kernel time + tests that cover it; (2) pocket-realistic shape:
# kernel time against wall time, idle gaps by following kernel
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3j}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py -q -m gpu -x -k "grad" 2>&1 | tail -4 > $O/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profp -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_pocket_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fb_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profp/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
python tools/prof_gaps.py $O/profp/pk_results.db 15 > $O/pocket_gaps.txt 2>&1
rm -rf $O/prof $O/profp
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_pocket.json 2>/dev/null
cat $O/tests.txt; cut -c1-330 $O/bench.json; cut -c1-330 $O/bench_pocket.json; grep -n "pair_bias_finish\|kernels," $O/fwdbwd_kernel_stats.txt | cut -c1-160; head -30 $O/pocket_gaps.txt; head -30 $O/pocket_kernel_stats.txt | cut -c1-170
