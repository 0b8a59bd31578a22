#!/bin/bash
# round 5: does a late start of every CU's second GEMM work-group (main loop against the neighbour's epilogue) shorten the node-level GEMMs?
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c23; mkdir -p $O
for L in "" fabind_amd/_ab/libfabind_gs1.so fabind_amd/_ab/libfabind_gs3.so fabind_amd/_ab/libfabind_gs6.so ""; do
  FABIND_LIB=$L timeout 300 python tools/probes/gemm_stagger.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm_stagger.txt
done
