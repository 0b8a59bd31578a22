# Same-box interleaved A/B of the headline (default) bench: current tree vs the side worktree _ab_prev (an older commit,
# `git worktree add _ab_prev <commit>` + build there).  LEGACY=1: both time round 1's batch construction
# (FABIND_BENCH_LEGACY_BATCH=1: 4 geometries tiled 16x, plain init) -- needed when _ab_prev is the round-1 tree, whose bench.py
# builds only that; otherwise both trees build the current workload.
for pass in $(seq 1 ${PASSES:-3}); do
  for tree in . _ab_prev; do
    extra="--no-extras"; [ "${LEGACY:-0}" = "1" ] && [ "$tree" = "_ab_prev" ] && extra=""
    (cd $tree && FABIND_BENCH_LEGACY_BATCH=${LEGACY:-0} python bench.py --no-cpu-baseline $extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD-AB pass $pass tree=$tree', round(d['value'],2), round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'avg_us', round(d['roofline']['avg_us'],1), d['roofline']['kernel'][:32])")
  done
done
