#!/bin/bash
# round 6: split-precision sites level 3 (hidden layers of node MLP / protein Transition in split precision) against level 2:
# parity gaps at the headline shape (n_iter 1 / 2 / 8), full IaBNet at production size, config 3 read literally; and what it costs.
O=${GRAFT_REPO_ROOT:-.}/gpurun_out; mkdir -p $O
for l in 2 3; do
  echo "== FABIND_SPLIT_SITES=$l"
  FABIND_SPLIT_SITES=$l python -m pytest tests/test_gpu_headline.py -k "bf16_gap" -s -q 2>&1 | grep "headline shape\|passed\|failed"
  FABIND_SPLIT_SITES=$l python -m pytest tests/test_gpu_production.py -k "iabnet_production_size_matches or config3" -s -q 2>&1 | grep "bf16:\|IaBNet 512\|config 3 read\|passed\|failed"
done
tools/ab.sh r6_split3 2 "FABIND_SPLIT_SITES=2" "FABIND_SPLIT_SITES=3" --mode fwdbwd --steps 10 --warmup 3
tools/ab.sh r6_split3 1 "FABIND_SPLIT_SITES=2" "FABIND_SPLIT_SITES=3" --mode fwdbwd --n-iter 8 --steps 4 --warmup 2
tools/ab.sh r6_split3 1 "FABIND_SPLIT_SITES=2" "FABIND_SPLIT_SITES=3" --steps 8 --warmup 2
