#!/bin/bash
# round 5: the weight-gradient contraction's layouts SINGLE-PROCESS: is any of them nondeterministic without contention?
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c49; mkdir -p $O
for w in 16 4 8; do
  echo "== bf16, FABIND_TN_WAVES=$w, one process"
  FABIND_TN_WAVES=$w timeout 900 python tools/probes/stack_repeat.py 25 solo bf16 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-260 | tail -4
done
