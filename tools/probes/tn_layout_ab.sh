# Same-box A/B of the TN contraction layouts inside the default bench step: headline value + the TN launch groups.
O=gpurun_out/tnab; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" FABIND_BENCH_DUMP_PROFILE=$O/groups_$n.txt python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', d['value'], d['ms_per_step'])"
}
for r in 1 2 3; do
  run w4_$r FABIND_TN_WAVES=4
  run w16_$r FABIND_TN_WAVES=16
  run w20_$r FABIND_TN_WAVES=20
done
for n in w4_1 w16_1 w20_1; do echo == $n; grep "gemm_tn" $O/groups_$n.txt | head -9; done
