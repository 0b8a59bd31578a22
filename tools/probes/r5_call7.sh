# round 5, call 7: which of the inter-edge attention changes moved the bf16 headline gap (8.389e-5 -> 1.003e-4 A)?
O=$GRAFT_REPO_ROOT/gpurun_out/r5c7; mkdir -p $O
for cfg in "FABIND_INTER_ATTN_ROWS=0 FABIND_INTER_ATTN_INPLACE=0" "FABIND_INTER_ATTN_ROWS=1 FABIND_INTER_ATTN_INPLACE=0" "FABIND_INTER_ATTN_ROWS=1 FABIND_INTER_ATTN_INPLACE=1"; do
  echo "== $cfg" | tee -a $O/gap.txt
  env $cfg timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -s -k "bf16_gap or bf16x3_meets" 2>&1 | grep -E "headline shape" | tee -a $O/gap.txt
done
