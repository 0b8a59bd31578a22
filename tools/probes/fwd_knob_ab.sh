# Same-box A/B of the fused edge FORWARD variant on the headline bench and the forward-only bench, interleaved PASSES times.
for pass in $(seq 1 ${PASSES:-3}); do
  for v in 0 1; do
    FABIND_EDGE_FWD_VARIANT=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FWD-AB pass $pass variant=$v fwdbwd', round(d['value'],2), round(d['ms_per_step'],2))"
    FABIND_EDGE_FWD_VARIANT=$v python bench.py --no-cpu-baseline --no-extras --mode fwd 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FWD-AB pass $pass variant=$v fwd   ', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
