#!/bin/bash
# round 5, call 10: LN-fold tests again (mask-aligned reference), the FABind+ production-size tests, plus_train launch groups with the fold
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plus.py -x -q -s  > $O/tests_plus.log 2>&1; grep -n "edge LN-fold\|worst\|passed\|failed" $O/tests_plus.log
timeout 1500 python -m pytest tests/test_gpu_production.py tests/test_gpu_plus_dense_api.py tests/test_gpu_training.py tests/test_gpu_bench.py -x -q -s -k "plus" > $O/tests_prod.log 2>&1; tail -5 $O/tests_prod.log
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_plus_groups.json 2>/dev/null

