# Same-box A/B of the Layout cache (FABIND_LAYOUT_CACHE=0 rebuilds the layout every step) on the headline bench, interleaved.
for pass in $(seq 1 ${PASSES:-3}); do
  for v in 0 1; do
    FABIND_LAYOUT_CACHE=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LAYOUT-AB pass $pass cache=$v', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
