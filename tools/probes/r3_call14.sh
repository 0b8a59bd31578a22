# idle gaps of the headline step's timed region (which kernels follow the gaps), and the kernel time / wall time ratio
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3o}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_gaps.py $O/prof/fb_results.db 10 edges_kernelILb1 2 > $O/gaps.txt 2>&1
python tools/prof_summary.py $O/prof/fb_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
rm -rf $O/prof
cat $O/bench_prof.log | cut -c1-200 | tail -2; cat $O/gaps.txt
