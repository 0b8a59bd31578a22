O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c14}; mkdir -p $O
python -m pytest tests/test_gpu_model.py -x -q > $O/tests_a.log 2>&1; tail -12 $O/tests_a.log
for i in 1 2; do
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model plan', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_PLAN=0 python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model no plan', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
