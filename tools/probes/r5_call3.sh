# round 5, call 3: new tests (x3 saving forward, config 3 read literally, two-rank DP on ragged pocket batches, the tightened bf16 gate), gate-mode A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_edge" > $O/tests_k.log 2>&1; tail -3 $O/tests_k.log
timeout 1500 python -m pytest tests/test_gpu_production.py -x -q -s -k "config3" > $O/tests_c3.log 2>&1; grep -E "config 3|bf16:|six-term|passed|failed|Error" $O/tests_c3.log | tail -8
timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q -s > $O/tests_dp.log 2>&1; grep -E "rank|passed|failed|Error" $O/tests_dp.log | tail -8
timeout 1800 python -m pytest tests/test_gpu_headline.py -x -q -s > $O/tests_h.log 2>&1; grep -E "headline shape|passed|failed" $O/tests_h.log | tail -14
tools/ab.sh r5c3/x3save 2 "FABIND_X3_EDGE_SAVE_FWD=1" "FABIND_X3_EDGE_SAVE_FWD=0" --precision bf16x3 --steps 8 --warmup 2
tools/ab.sh r5c3/x3save_n8 1 "FABIND_X3_EDGE_SAVE_FWD=1" "FABIND_X3_EDGE_SAVE_FWD=0" --precision bf16x3 --n-iter 8 --steps 3 --warmup 1
