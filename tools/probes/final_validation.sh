python -m pytest tests -q -m gpu 2>&1 | tail -2
bash tools/probes/final_profiles.sh > gpurun_out/final_profiles.log 2>&1; rm -rf gpurun_out/r2g/prof; mkdir -p gpurun_out/fin
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench.err
python -c "import json; d=json.load(open('gpurun_out/fin/bench_default.json')); print('DEFAULT', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else None) for k,v in d.items() if isinstance(v,dict) and 'value' in v})"
PASSES=2 bash tools/probes/headline_ab_samebox.sh 2>&1 | grep HEAD-AB
python bench.py --no-cpu-baseline --mode plus_sampling --n-iter 8 --poses 4 2>/dev/null | tail -1 > gpurun_out/fin/bench_plus_sampling.json
python bench.py --no-cpu-baseline --mode plus_train 2>/dev/null | tail -1 > gpurun_out/fin/bench_plus_train.json
python -c "
import json
for n in ('plus_sampling','plus_train'):
    d=json.load(open('gpurun_out/fin/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'])"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --mode fwdbwd --n-prot 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pocket', d['value'], d['ms_per_step'])"; done
