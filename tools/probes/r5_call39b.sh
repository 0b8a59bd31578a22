#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c44; mkdir -p $O
for c in A B C D E F; do (timeout 900 python tools/probes/model_repeat3.py 70 $c 2>&1 | grep -v amdgpu.ids | grep "passes differ" > $O/d_$c.txt) & done
wait; cat $O/d_*.txt; ls -la $GRAFT_REPO_ROOT/gpurun_out/las_case_*.pt 2>/dev/null | head
