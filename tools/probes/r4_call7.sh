O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c7}; mkdir -p $O
python tools/probes/edge_bwd_h128.py > $O/edge_h128.txt 2>&1; cat $O/edge_h128.txt | tail -22
for i in 1 2; do
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_PREFETCH=1 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket prefetch', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
(time python bench.py --cpu-baseline-full > $O/cpu_baseline_full.json 2> $O/cpu_baseline_full.log) 2>&1 | tail -3
cat $O/cpu_baseline_full.log | tail -20
