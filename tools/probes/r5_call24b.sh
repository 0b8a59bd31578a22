#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c30; mkdir -p $O
timeout 600 python tools/probes/model_repeat.py 2>&1 | grep -v amdgpu.ids | tee $O/model_repeat.txt
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_plus.py tests/test_gpu_production.py tests/test_gpu_dp.py tests/test_gpu_training.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
