O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c8}; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
python tools/probes/edge_bwd_h128.py > $O/edge_h128.txt 2>&1; cat $O/edge_h128.txt | tail -24
python tools/probes/pocket_hostprof2.py > $O/hostprof2.txt 2>&1; head -6 $O/hostprof2.txt
for i in 1 2; do
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_PREFETCH=1 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket prefetch', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_DUMP_PROFILE=$O/pocket_launch_groups.txt python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 > /dev/null 2>&1
head -30 $O/pocket_launch_groups.txt | cut -c1-150
