#!/bin/bash
# round 5: the headline step under contention (four copies at once), bf16 and bf16x3
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c46; mkdir -p $O
for prec in bf16 bf16x3; do
  echo "== $prec, four processes"
  for c in A B C D; do (timeout 1200 python tools/probes/stack_repeat.py 40 $c $prec 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-400 > $O/${prec}_$c.txt) & done
  wait; cat $O/${prec}_*.txt
done
