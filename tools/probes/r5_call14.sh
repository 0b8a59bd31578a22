# round 5, call 14: 'bf16x3' with the bf16 edge pipeline and the bf16 fused attention (config.set_x3_edge / set_x3_attn): gates and speed
O=$GRAFT_REPO_ROOT/gpurun_out/r5c14; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_headline.py -q -s -k "bf16x3" 2>&1 | grep -E "headline shape|passed|failed|Error" | tee $O/gate.txt
timeout 2400 python -m pytest tests/test_gpu_production.py tests/test_gpu_stack.py tests/test_gpu_training.py -x -q -s > $O/tests.log 2>&1; grep -E "bf16x3|passed|failed" $O/tests.log | tail -12
python __graft_entry__.py --smoke 2>&1 | grep smoke
for cfg in "FABIND_X3_EDGE=bf16 FABIND_X3_ATTN=bf16" "FABIND_X3_EDGE=bf16 FABIND_X3_ATTN=fp32" "FABIND_X3_EDGE=split FABIND_X3_ATTN=fp32"; do
  for i in 1 2; do
  env $cfg python bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-50s' % sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2))" "[$cfg] gate mode n_iter 1" | tee -a $O/x3_ab.txt
  done
  env $cfg python bench.py --precision bf16x3 --n-iter 8 --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-50s' % sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2))" "[$cfg] gate mode n_iter 8" | tee -a $O/x3_ab.txt
done
