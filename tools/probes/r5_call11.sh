#!/bin/bash
# round 5, call 11: kernel trace of the FABind+ training step with the folded edge LayerNorm
O=$GRAFT_REPO_ROOT/gpurun_out/r5c11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/prof
head -70 $O/plus_train_kernel_stats.txt
