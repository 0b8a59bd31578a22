#!/bin/bash
# round 5: kernel trace of the FABind+ sampling step (n_iter 8, 20 poses per complex)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c33; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o ps -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_sampling --n-iter 8 --no-cpu-baseline --no-extras --steps 1 --warmup 1 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/ps_results.db > $O/plus_sampling_kernel_stats.txt 2>&1
rm -rf $O/prof
head -45 $O/plus_sampling_kernel_stats.txt | cut -c1-100,108-170
tail -2 $O/bench_prof.log | cut -c1-300
