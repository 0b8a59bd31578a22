# Same-box A/B of the fused edge kernel knobs on the headline bench (one gpurun call): backward variant (0 = two LDS tiles,
# 1 = single in-place tile) x XCD-aware tile order (0 / 1), interleaved PASSES times.
for pass in $(seq 1 ${PASSES:-3}); do
  for cfg in "1 1" "0 1" "1 0" "0 0"; do
    set -- $cfg
    FABIND_EDGE_BWD_VARIANT=$1 FABIND_EDGE_XCD=$2 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KNOB-AB pass $pass variant=$1 xcd=$2', round(d['value'],2), round(d['ms_per_step'],2), 'bwd avg_us', round(d['roofline']['avg_us'],1))"
  done
done
