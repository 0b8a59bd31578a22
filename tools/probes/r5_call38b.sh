#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c43; mkdir -p $O
echo "== default build"
for c in A B C D E F; do (timeout 900 python tools/probes/model_repeat3.py 70 $c 2>&1 | grep -v amdgpu.ids | grep "passes differ" > $O/d_$c.txt) & done
wait; cat $O/d_*.txt
echo "== LAS accumulates with scalar FMAs (no v_pk_fma)"
for c in A B C D E F; do (FABIND_LIB=fabind_amd/_ab/libfabind_lasnp.so timeout 900 python tools/probes/model_repeat3.py 70 $c 2>&1 | grep -v amdgpu.ids | grep "passes differ" > $O/v_$c.txt) & done
wait; cat $O/v_*.txt
