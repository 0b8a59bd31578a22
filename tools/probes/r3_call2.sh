# round 3, call 2: GPU suite with the split-bf16 mode (unit test of the x3 GEMM, gate tests in bf16x3), then the headline step in that
# mode: bench line, live launch groups, rocprofv3 kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -15 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]|split-bf16" $O/tests.log > $O/tests_gaps.txt
FABIND_BENCH_DUMP_PROFILE=$O/x3_launch_groups.txt python bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 5 > $O/bench_x3.json 2> $O/bench_x3.err
cat $O/bench_x3.json; tail -3 $O/bench_x3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profx3 -o x3 -- python3 $GRAFT_REPO_ROOT/bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_x3_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profx3/x3_results.db > $O/x3_kernel_stats.txt 2>&1
rm -rf $O/profx3
head -30 $O/x3_kernel_stats.txt
head -30 $O/x3_launch_groups.txt
