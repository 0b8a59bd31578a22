#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c41; mkdir -p $O
for c in A B C D E F; do (timeout 900 python tools/probes/model_repeat3.py 70 $c 2>&1 | grep -v amdgpu.ids > $O/r3_$c.txt) & done
wait
cat $O/r3_*.txt | cut -c1-900
