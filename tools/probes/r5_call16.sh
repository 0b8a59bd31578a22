# round 5, call 16: the full GPU suite + smoke on the round's tree
O=$GRAFT_REPO_ROOT/gpurun_out/r5c16; mkdir -p $O
timeout 3300 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
python __graft_entry__.py --smoke 2>&1 | grep smoke | tee $O/smoke.txt
