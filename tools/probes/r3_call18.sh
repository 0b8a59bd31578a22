# FABind+ sampling-mode inference: kernel trace summary + live launch groups + idle gaps
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3v}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o ps -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_sampling --poses 4 --no-cpu-baseline --no-extras --steps 1 --warmup 1 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/ps_results.db > $O/plus_sampling_kernel_stats.txt 2>&1
python tools/prof_gaps.py $O/prof/ps_results.db 20 > $O/gaps.txt 2>&1
rm -rf $O/prof
FABIND_BENCH_DUMP_PROFILE=$O/ps_launch_groups.txt python bench.py --mode plus_sampling --poses 4 --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $O/bench_ps.json 2>/dev/null
cut -c1-300 $O/bench_ps.json; head -40 $O/plus_sampling_kernel_stats.txt | cut -c1-165; head -14 $O/gaps.txt; head -25 $O/ps_launch_groups.txt
