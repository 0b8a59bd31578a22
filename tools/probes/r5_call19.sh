# round 5, call 19: q|k|v site split for the v columns only: gaps, tests, speed
O=$GRAFT_REPO_ROOT/gpurun_out/r5c19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k "bf16_gap" 2>&1 | grep -E "headline shape|passed|failed" | tee $O/gap.txt
timeout 2400 python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py tests/test_gpu_model.py tests/test_gpu_training.py tests/test_gpu_param_pack.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
done
python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fwd', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
FABIND_SPLIT_SITES=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline level 0', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
