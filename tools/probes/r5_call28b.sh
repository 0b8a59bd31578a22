#!/bin/bash
# round 5, final evidence on the final tree: GPU suite, smoke, r5_final.sh (traces, launch groups, PMC), plus_train trace, default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r5fin6; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/probes/r5_final.sh r5fin6 > $O/final.log 2>&1; tail -3 $O/final.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_profpt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1; rm -rf $O/profpt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source'))
for k,v in d.items():
    if isinstance(v,dict) and 'value' in v and k!='cpu_baseline': print(k, round(v['value'],1), round(v['ms_per_step'],2))
"
