# Same-box A/B of the atomic-free pair-Hadamard adjoint (FABIND_PAIRHAD_ROWS=0 restores the float-atomics kernel).
run() { n=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras ${ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', round(d['value'],1), round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
  ARGS="" run big_off_$r FABIND_PAIRHAD_ROWS=0
  ARGS="" run big_on_$r FABIND_PAIRHAD_ROWS=1
done
for r in 1 2 3; do
  ARGS="--n-prot 100" run pocket_off_$r FABIND_PAIRHAD_ROWS=0
  ARGS="--n-prot 100" run pocket_on_$r FABIND_PAIRHAD_ROWS=1
done
