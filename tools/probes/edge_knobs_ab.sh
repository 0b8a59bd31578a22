# Same-box A/B of the fused edge kernel knobs on the headline bench (one gpurun call), interleaved PASSES times:
#   backward variant (0 = two LDS tiles / one work-group per CU, 1 = single in-place tile / two per CU)
#   x library (default build = s_setprio around the MFMA clusters; libfabind_hip_noprio.so = built with -DFE_NO_SETPRIO)
for pass in $(seq 1 ${PASSES:-3}); do
  for cfg in "0 prio" "1 prio" "0 noprio" "1 noprio"; do
    set -- $cfg
    lib=""; [ "$2" = "noprio" ] && lib="$PWD/fabind_amd/libfabind_hip_noprio.so"
    FABIND_LIB=$lib FABIND_EDGE_BWD_VARIANT=$1 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KNOB-AB pass $pass variant=$1 lib=$2', round(d['value'],2), round(d['ms_per_step'],2), 'dominant avg_us', round(d['roofline']['avg_us'],1))"
  done
done
