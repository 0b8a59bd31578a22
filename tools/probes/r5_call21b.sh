#!/bin/bash
# round 5: same-box pairs, this tree vs the tree before the torch-glue changes (_ab_prev = ed6633a, same library): model, pocket, headline, plus_train
O=$GRAFT_REPO_ROOT/gpurun_out/r5c27; mkdir -p $O
run() { (cd $1 && python bench.py $3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$2 tree=$1', round(d['value'],1), round(d['ms_per_step'],2))") | tee -a $O/ab.txt; }
for pass in 1 2 3; do
  for tree in . _ab_prev; do
    run $tree model "--mode model --steps 6 --warmup 2"
    run $tree pocket "--n-prot 100 --steps 30 --warmup 5"
  done
done
for pass in 1 2; do
  for tree in . _ab_prev; do
    run $tree headline "--steps 10 --warmup 3"
    run $tree plus_train "--mode plus_train --steps 3 --warmup 2"
  done
done
