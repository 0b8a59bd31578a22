#!/bin/bash
# round 5: layout counts without atomics, atomics-free row-gather adjoints: full GPU suite + model / plus_train / headline steps
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c26; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
for i in 1 2; do
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
done
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
