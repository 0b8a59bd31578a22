# Same-box A/B of the one-launch parameter pack (FABIND_PARAM_PACK=0: torch ops) on the headline bench and the pocket shape, interleaved.
for pass in $(seq 1 ${PASSES:-3}); do
  for v in 0 1; do
    FABIND_PARAM_PACK=$v python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PACK-AB pass $pass pack=$v headline', round(d['value'],2), round(d['ms_per_step'],2))"
    FABIND_PARAM_PACK=$v python bench.py --n-prot 100 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PACK-AB pass $pass pack=$v pocket  ', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
