# Same-box interleaved A/B of the headline (default) bench: current tree vs the side worktree _ab_prev (an older commit).
for pass in $(seq 1 ${PASSES:-3}); do
  for tree in . _ab_prev; do
    (cd $tree && python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD-AB pass $pass tree=$tree', round(d['value'],2), round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4))")
  done
done
