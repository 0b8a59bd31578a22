O=$GRAFT_REPO_ROOT/gpurun_out/r5c13; mkdir -p $O
echo "== FABIND_SPLIT_SITES=2" | tee -a $O/gap.txt
FABIND_SPLIT_SITES=2 timeout 900 python -m pytest tests/test_gpu_headline.py -q -s -k "bf16_gap" 2>&1 | grep -E "headline shape|passed|failed" | tee -a $O/gap.txt
tools/ab.sh r5c13/split2 2 "FABIND_SPLIT_SITES=2" "FABIND_SPLIT_SITES=1" --steps 10 --warmup 3
tools/ab.sh r5c13/split2_fwd 1 "FABIND_SPLIT_SITES=2" "FABIND_SPLIT_SITES=1" --mode fwd --steps 10 --warmup 3
