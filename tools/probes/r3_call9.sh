# FABind+ training step: live launch groups + kernel trace summary; pocket-realistic shape of the headline bench for reference
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3i}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/profpt
FABIND_BENCH_DUMP_PROFILE=$O/pt_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_pt.json 2>$O/bench_pt.err
python bench.py --n-prot 100 --no-cpu-baseline --no-extras > $O/bench_pocket.json 2>/dev/null
cat $O/bench_pt.json $O/bench_pocket.json | cut -c1-400; head -40 $O/pt_launch_groups.txt
