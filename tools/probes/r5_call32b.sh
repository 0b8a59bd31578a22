#!/bin/bash
# round 5: kernel trace of the full model in the reference's training configuration (train mode, n_iter 8)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c37; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o mt -- python3 $GRAFT_REPO_ROOT/bench.py --mode model --train-mode --n-iter 8 --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/mt_results.db > $O/model_train_kernel_stats.txt 2>&1
rm -rf $O/prof
head -40 $O/model_train_kernel_stats.txt | cut -c1-100,108-170
grep -o '"value": [0-9.]*, "unit": "complexes/s", "n_gpus": 1, "steps": 4, "warmup": 2, "ms_per_step": [0-9.]*' $O/bench_prof.log
