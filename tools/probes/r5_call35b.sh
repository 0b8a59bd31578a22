#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c40; mkdir -p $O
echo "== alone" | tee $O/repeat2.txt
timeout 600 python tools/probes/model_repeat2.py 10 alone 2>&1 | grep -v amdgpu.ids | tee -a $O/repeat2.txt
echo "== three copies at once" | tee -a $O/repeat2.txt
(timeout 600 python tools/probes/model_repeat2.py 10 A 2>&1 | grep -v amdgpu.ids >> $O/repeat2_A.txt) &
(timeout 600 python tools/probes/model_repeat2.py 10 B 2>&1 | grep -v amdgpu.ids >> $O/repeat2_B.txt) &
timeout 600 python tools/probes/model_repeat2.py 10 C 2>&1 | grep -v amdgpu.ids | tee -a $O/repeat2.txt
wait
cat $O/repeat2_A.txt $O/repeat2_B.txt | tee -a $O/repeat2.txt
