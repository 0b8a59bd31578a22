for pass in 1 2 3 4; do
  for tree in . _ab_prev; do
    (cd $tree && python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB pass $pass tree=$tree fwdbwd', round(d['value'],2), round(d['ms_per_step'],2))")
    (cd $tree && python bench.py --no-cpu-baseline --no-extras --mode fwd 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB pass $pass tree=$tree fwd   ', round(d['value'],2), round(d['ms_per_step'],2))")
  done
done
