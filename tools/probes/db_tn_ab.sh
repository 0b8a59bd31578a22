# Same-box A/B of the bias gradient riding along with the TN contraction (FABIND_FUSE_DB_TN=0 restores the separate column sums).
run() { n=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras ${ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
  ARGS="" run big_off_$r FABIND_FUSE_DB_TN=0
  ARGS="" run big_on_$r FABIND_FUSE_DB_TN=1
done
for r in 1 2 3; do
  ARGS="--n-prot 100" run pocket_off_$r FABIND_FUSE_DB_TN=0
  ARGS="--n-prot 100" run pocket_on_$r FABIND_FUSE_DB_TN=1
done
