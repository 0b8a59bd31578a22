# Same-box A/B of the FABind+ inference-only kernels (env switches), interleaved, two passes.
for pass in 1 2; do
  for cfg in "1 1" "0 1" "1 0" "0 0"; do
    set -- $cfg
    FABIND_PLUS_FUSE_PAIR=$1 FABIND_PLUS_FOLD_EDGE_LN=$2 python bench.py --mode plus_sampling --batch 64 --n-iter 8 --poses 4 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB pass $pass fuse_pair=$1 ln_folds=$2', round(d['value'],1), round(d['ms_per_step'],1))"
  done
done
