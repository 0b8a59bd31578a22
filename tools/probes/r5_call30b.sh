#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c35; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_plus.py tests/test_gpu_plus_dense_api.py tests/test_gpu_production.py -x -q -k "plus" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('plus_train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt; done
