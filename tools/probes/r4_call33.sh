O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c33}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -s > $O/tests_train.log 2>&1; tail -25 $O/tests_train.log | cut -c1-260
