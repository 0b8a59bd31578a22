#!/bin/bash
# round 5, call 9: FABind+ LayerNorm-folded first edge Linear under autograd -- tests, then same-box A/B of plus_train
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plus.py -x -q -s > $O/tests_plus.log 2>&1; tail -25 $O/tests_plus.log
timeout 600 bash tools/ab.sh r5c9/lnfold 2 "FABIND_PLUS_FOLD_EDGE_LN_TRAIN=1" "FABIND_PLUS_FOLD_EDGE_LN_TRAIN=0" --mode plus_train --steps 3 --warmup 2
