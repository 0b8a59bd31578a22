# Same-box A/B of the small-M tile switch of fabind_gemm (FABIND_GEMM_SMALL_M=0 disables it): pocket shape and headline shape.
run() { n=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras ${ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
  ARGS="--n-prot 100" run pocket_off_$r FABIND_GEMM_SMALL_M=0
  ARGS="--n-prot 100" run pocket_on_$r FABIND_GEMM_SMALL_M=192
  ARGS="--n-prot 100" run pocket_on320_$r FABIND_GEMM_SMALL_M=320
done
for r in 1 2; do
  ARGS="" run big_off_$r FABIND_GEMM_SMALL_M=0
  ARGS="" run big_on_$r FABIND_GEMM_SMALL_M=192
done
