O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c28}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1; rm -rf $O/prof
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_pt.json 2>/dev/null
head -40 $O/plus_train_kernel_stats.txt | cut -c1-170
