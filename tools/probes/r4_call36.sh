O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c36}; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
for i in 1 2 3; do
python bench.py --mode model --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model saved-forward H>=128', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_EDGE_SAVE_MIN_H=256 python bench.py --mode model --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model saved-forward H>=256', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
