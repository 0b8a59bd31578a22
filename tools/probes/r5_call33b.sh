#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c38; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_stack.py tests/test_gpu_model.py tests/test_gpu_training.py tests/test_gpu_dp.py tests/test_gpu_kernels.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do python bench.py --mode model --train-mode --n-iter 8 --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model train n_iter8', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt; done
