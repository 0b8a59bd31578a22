# round 3, call 1: the whole GPU suite incl. the production-size parity tests (printed gaps -> gpurun_out/r3a/tests.log) and a kernel
# trace of the headline step in fp32 mode (where do its 610 ms go: planning input for the split-bf16 mode)
O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -5 $O/tests.log
grep -E "RMSD|gap|gradients|loss|moved|per-tensor|^    [0-9]" $O/tests.log > $O/tests_gaps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof32 -o fp32 -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp32 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_fp32.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof32/fp32_results.db > $O/fp32_kernel_stats.txt 2>&1
rm -rf $O/prof32
head -40 $O/fp32_kernel_stats.txt
