#!/bin/bash
# round 5: the headline step under contention with the weight-gradient queue OFF (every contraction its own launch pair): does the rare
# single-gradient mismatch stay?
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c47; mkdir -p $O
echo "== bf16, FABIND_TN_DEFER=0, four processes"
for c in A B C D; do (FABIND_TN_DEFER=0 timeout 1200 python tools/probes/stack_repeat.py 60 $c bf16 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-300 > $O/q0_$c.txt) & done
wait; cat $O/q0_*.txt
echo "== bf16, default (queue on), four processes"
for c in A B C D; do (timeout 1200 python tools/probes/stack_repeat.py 60 $c bf16 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-300 > $O/q1_$c.txt) & done
wait; cat $O/q1_*.txt
