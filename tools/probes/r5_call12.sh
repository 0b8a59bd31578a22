# round 5, call 12: split-precision sites of the bf16 mode (config.set_split_sites): the real gap at n_iter 1 / 2 / 8 and what it costs
O=$GRAFT_REPO_ROOT/gpurun_out/r5c12; mkdir -p $O
for lvl in 1 0; do
  echo "== FABIND_SPLIT_SITES=$lvl" | tee -a $O/gap.txt
  FABIND_SPLIT_SITES=$lvl timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k "bf16_gap" 2>&1 | grep -E "headline shape|passed|failed" | tee -a $O/gap.txt
done
timeout 1800 python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py tests/test_gpu_model.py tests/test_gpu_training.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
tools/ab.sh r5c12/split 2 "FABIND_SPLIT_SITES=1" "FABIND_SPLIT_SITES=0" --steps 10 --warmup 3
tools/ab.sh r5c12/split_fwd 2 "FABIND_SPLIT_SITES=1" "FABIND_SPLIT_SITES=0" --mode fwd --steps 10 --warmup 3
tools/ab.sh r5c12/split_n8 1 "FABIND_SPLIT_SITES=1" "FABIND_SPLIT_SITES=0" --n-iter 8 --steps 4 --warmup 2
tools/ab.sh r5c12/split_pocket 1 "FABIND_SPLIT_SITES=1" "FABIND_SPLIT_SITES=0" --n-prot 100 --steps 30 --warmup 5
