#!/bin/bash
# round 5: full GPU suite + smoke + the driver's default bench line on the final tree
O=$GRAFT_REPO_ROOT/gpurun_out/r5fin3; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source'))
for k,v in d.items():
    if isinstance(v,dict) and 'value' in v and k!='cpu_baseline': print(k, round(v['value'],1), round(v['ms_per_step'],2))
"
