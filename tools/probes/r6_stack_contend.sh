#!/bin/bash
# round 6: the whole stack training step (B = 64, 1500 / 40, bf16, fwd + bwd) N passes in FOUR concurrent processes, outputs and every parameter
# gradient compared bit for bit with the process's first pass (tools/probes/stack_repeat.py) -- round 5 saw 2-12 % of the passes differ in one
# weight gradient (profiles/r05_contention.txt); with the contraction's loop fixed (csrc/gemm.hip) none may.  Also the 8-wave layout.
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r6sc; mkdir -p $O
for w in 16 8; do
  echo "== bf16, FABIND_TN_WAVES=$w, four processes"
  for c in A B C D; do (FABIND_TN_WAVES=$w timeout 1200 python tools/probes/stack_repeat.py ${1:-40} $c bf16 2>&1 | grep "DIFFERS\|passes differ" | cut -c1-300 > $O/w${w}_$c.txt) & done
  wait; cat $O/w${w}_*.txt
done
