O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c15}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attn_mfma.py -x -q -k "fused" > $O/tests_a.log 2>&1; grep "fused attention fwd+bwd" $O/tests_a.log | cut -c1-260; tail -3 $O/tests_a.log
timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_stack.py tests/test_gpu_model.py -x -q > $O/tests_b.log 2>&1; tail -4 $O/tests_b.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline fused-train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_ATTN_FUSED_TRAIN=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline tensor path', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket fused-train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_ATTN_FUSED_TRAIN=0 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket tensor path', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
