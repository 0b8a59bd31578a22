#!/bin/bash
# Same-box interleaved A/B of bench.py under two environments (one gpurun call; box-to-box spread is +-3 %, so only pairs from one call compare).
#   tools/ab.sh <out-name> <rounds> "<env A>" "<env B>" [bench.py args ...]
# e.g. tools/ab.sh edge_save 2 "FABIND_EDGE_SAVE_FWD=1" "FABIND_EDGE_SAVE_FWD=0" --steps 10 --warmup 3
# Prints and appends to gpurun_out/<out-name>.txt one line per run: label, value, ms per step.
O=${GRAFT_REPO_ROOT:-.}/gpurun_out; mkdir -p $O
NAME=$1; R=$2; EA=$3; EB=$4; shift 4
for i in $(seq 1 $R); do
  for E in "$EA" "$EB"; do
    env $E python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-60s' % sys.argv[1], round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step')" "[$E] $*" | tee -a $O/$NAME.txt
  done
done
