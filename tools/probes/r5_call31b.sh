#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c36; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "drop" > $O/tests_k.log 2>&1; tail -2 $O/tests_k.log
timeout 2400 python -m pytest tests/test_gpu_stack.py tests/test_gpu_model.py tests/test_gpu_training.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
python bench.py --train-mode --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('train_mode', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('eval_mode', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
done
