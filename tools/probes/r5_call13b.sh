#!/bin/bash
# round 5: lnfold kernels with U edges in flight per wave: stand-alone times of four builds, then the plus tests on the default build
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c20; mkdir -p $O
for L in "" fabind_amd/_ab/libfabind_lf_u1.so fabind_amd/_ab/libfabind_lf_u4.so; do
  FABIND_LIB=$L timeout 300 python tools/probes/lnfold_time.py 2>&1 | tee -a $O/lnfold_time.txt
done
timeout 900 python -m pytest tests/test_gpu_plus.py tests/test_gpu_attn_mfma.py -x -q  > $O/tests.log 2>&1; tail -3 $O/tests.log
