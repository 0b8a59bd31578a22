# Same-box interleaved A/B of the forward-only bench (--mode fwd): current tree vs the side worktree _ab_prev.
for pass in 1 2 3; do
  for tree in . _ab_prev; do
    (cd $tree && python bench.py --no-cpu-baseline --mode fwd 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FWD-AB pass $pass tree=$tree', round(d['value'],1), round(d['ms_per_step'],2))")
  done
done
