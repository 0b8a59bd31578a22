#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c31; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "plan or reproducible" 2>&1 | tail -1 | tee -a $O/repeat.txt; done
