#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c42; mkdir -p $O
echo "== alone"; timeout 300 python tools/probes/las_repeat.py 3000 alone 2>&1 | grep -v amdgpu.ids | tee $O/las_alone.txt
echo "== six processes"
for c in A B C D E F; do (timeout 600 python tools/probes/las_repeat.py 3000 $c 2>&1 | grep -v amdgpu.ids > $O/las_$c.txt) & done
wait
cat $O/las_[A-F].txt
