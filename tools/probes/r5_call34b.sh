#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c39; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_gpu_dp.py -x -q -s 2>&1 | grep -v amdgpu.ids | grep "rank\|passed\|failed" | tee -a $O/dp_repeat.txt; done
