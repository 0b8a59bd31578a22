#!/bin/bash
# round 5: the headline step under contention with the weight-gradient contraction's other work-group layouts (FABIND_TN_WAVES = 4: 256 x 128
# tile, 4 waves, two work-groups per CU; 8: 8 waves): does the rare single-gradient mismatch depend on the kernel configuration?
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c48; mkdir -p $O
for w in 4 8; do
  echo "== bf16, FABIND_TN_WAVES=$w, four processes"
  for c in A B C D; do (FABIND_TN_WAVES=$w timeout 1200 python tools/probes/stack_repeat.py 60 $c bf16 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-300 > $O/w${w}_$c.txt) & done
  wait; cat $O/w${w}_*.txt
done
