# round 5, call 17: the evidence run (tools/probes/r5_final.sh) + the driver's default line as the file of record
bash tools/probes/r5_final.sh r5fin > $GRAFT_REPO_ROOT/gpurun_out/r5fin_stdout.txt 2>&1
python bench.py > $GRAFT_REPO_ROOT/gpurun_out/r5fin/bench_default.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5fin/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r5fin/bench_default.json'))
print('headline', round(d['value'],1), d['ms_per_step'], d['roofline']['kernel'][:60], d['roofline']['frac'], d['roofline'].get('traffic'))
for k,v in d.items():
    if isinstance(v, dict) and 'value' in v: print(k, round(v['value'],1), v.get('unit'))
    elif isinstance(v, dict) and 'error' in v: print(k, 'ERROR', v['error'][:200])
"
