#!/bin/bash
# round 5: after the LAS fix -- full-model and FABind+ steps under six processes sharing the device, the DP test ten times
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c45; mkdir -p $O
echo "== IaBNet step, six processes"
for c in A B C D E F; do (timeout 900 python tools/probes/model_repeat3.py 70 $c 2>&1 | grep -v amdgpu.ids | grep "DIFFERS\|passes differ" | cut -c1-300 > $O/m_$c.txt) & done
wait; cat $O/m_*.txt
echo "== FABind+ step, six processes"
for c in A B C D E F; do (timeout 900 python tools/probes/plus_repeat.py 40 $c 2>&1 | grep -v amdgpu.ids | grep "pass \|passes differ" | cut -c1-300 > $O/p_$c.txt) & done
wait; cat $O/p_*.txt
echo "== DP test x 10"
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 600 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -1; done
