# round 3, call 3: split-bf16 GEMM microbenchmark (both tiles, four epilogues), x3-related unit tests, the whole suite, and the x3 step
O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/probes/gemm_x3_bench.py > $O/gemm_x3_bench.txt 2>&1; cat $O/gemm_x3_bench.txt | grep X3GEMM
python -m pytest tests -m gpu -q -s -x -k "split or x3 or gemm" > $O/tests_x3.log 2>&1; tail -4 $O/tests_x3.log
python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -12 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]|split" $O/tests.log > $O/tests_gaps.txt
FABIND_BENCH_DUMP_PROFILE=$O/x3_launch_groups.txt python bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 5 > $O/bench_x3.json 2> $O/bench_x3.err
cat $O/bench_x3.json; tail -3 $O/bench_x3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profx3 -o x3 -- python3 $GRAFT_REPO_ROOT/bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_x3_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profx3/x3_results.db > $O/x3_kernel_stats.txt 2>&1
rm -rf $O/profx3
head -34 $O/x3_kernel_stats.txt
head -24 $O/x3_launch_groups.txt
