O=$GRAFT_REPO_ROOT/gpurun_out/r5c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_edge" > $O/tests_k.log 2>&1; tail -3 $O/tests_k.log
timeout 1500 python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py -x -q -k "grad or backward or train" > $O/tests_g.log 2>&1; tail -3 $O/tests_g.log
for L in "" fabind_amd/_ab/libfabind_ntl.so fabind_amd/_ab/libfabind_nt.so; do
  echo "== edge kernels, lib=[$L]" | tee -a $O/edge_nt.txt
  FABIND_LIB=$L python tools/probes/edge_bwd4_time.py 6 new 2>/dev/null | tee -a $O/edge_nt.txt
done
tools/ab.sh r5c2/defer 2 "FABIND_DEFER_DX=1" "FABIND_DEFER_DX=0" --steps 10 --warmup 3
tools/ab.sh r5c2/nt 2 "FABIND_LIB=fabind_amd/_ab/libfabind_nt.so" "FABIND_LIB=" --steps 10 --warmup 3
tools/ab.sh r5c2/defer_pocket 2 "FABIND_DEFER_DX=1" "FABIND_DEFER_DX=0" --n-prot 100 --steps 30 --warmup 5
