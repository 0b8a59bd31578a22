#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c32; mkdir -p $O
timeout 600 python tools/probes/plus_repeat.py 2>&1 | grep -v amdgpu.ids | tee $O/plus_repeat.txt
timeout 900 python -m pytest tests/test_gpu_data.py tests/test_gpu_plus.py -x -q 2>&1 | tail -2
