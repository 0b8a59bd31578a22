O=$GRAFT_REPO_ROOT/gpurun_out/r5c15; mkdir -p $O
for cfg in "FABIND_X3_EDGE=bf16 FABIND_X3_ATTN=fp32" "FABIND_X3_EDGE=split FABIND_X3_ATTN=bf16" "FABIND_X3_EDGE=split FABIND_X3_ATTN=fp32"; do
  echo "== $cfg" | tee -a $O/gates.txt
  env $cfg timeout 1200 python -m pytest tests/test_gpu_headline.py -q -s -k "bf16x3_meets" 2>&1 | grep -E "headline shape bf16x3" | tee -a $O/gates.txt
  env $cfg timeout 1800 python -m pytest tests/test_gpu_production.py -q -s -k "iabnet_production_size_matches" 2>&1 | grep -E "bf16x3:|IaBNet 512" | tee -a $O/gates.txt
  env $cfg python __graft_entry__.py --smoke 2>&1 | grep "bf16x3" | tee -a $O/gates.txt
done
