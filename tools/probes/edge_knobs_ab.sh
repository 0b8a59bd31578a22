# Same-box A/B of the fused edge backward variant on the headline bench (one gpurun call), interleaved PASSES times:
# 0 = two LDS tiles / one work-group per CU (default), 1 = single in-place tile / two per CU.
for pass in $(seq 1 ${PASSES:-3}); do
  for v in ${VARIANTS:-0 1}; do
    FABIND_EDGE_BWD_VARIANT=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KNOB-AB pass $pass variant=$v', round(d['value'],2), round(d['ms_per_step'],2), 'dominant avg_us', round(d['roofline']['avg_us'],1))"
  done
done
