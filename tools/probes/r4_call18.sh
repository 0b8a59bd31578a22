O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c18}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -m gpu -s > $O/tests_h.log 2>&1; tail -3 $O/tests_h.log
grep "complex 0 in\|worst tensors" $O/tests_h.log | cut -c1-520
