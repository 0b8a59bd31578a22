#!/bin/bash
# round 6: level 3 with the ligand-side Transition / output projection and the model's embedding Linears added
for l in 3; do
  echo "== FABIND_SPLIT_SITES=$l"
  FABIND_SPLIT_SITES=$l python -m pytest tests/test_gpu_headline.py -k "bf16_gap" -s -q 2>&1 | grep "headline shape\|passed\|failed"
  FABIND_SPLIT_SITES=$l python -m pytest tests/test_gpu_production.py -k "iabnet_production_size_matches or config3" -s -q 2>&1 | grep "bf16:\|IaBNet 512\|config 3 read\|passed\|failed"
done
