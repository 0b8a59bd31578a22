# round 5, call 4: inter-edge attention with rows dealt by degree: parity, A/B, kernel times
O=$GRAFT_REPO_ROOT/gpurun_out/r5c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "inter_attn" > $O/tests_k.log 2>&1; tail -3 $O/tests_k.log
timeout 1800 python -m pytest tests/test_gpu_stack.py tests/test_gpu_plus.py -x -q > $O/tests_s.log 2>&1; tail -3 $O/tests_s.log
tools/ab.sh r5c4/rows 2 "FABIND_INTER_ATTN_ROWS=1" "FABIND_INTER_ATTN_ROWS=0" --steps 10 --warmup 3
tools/ab.sh r5c4/rows_pocket 2 "FABIND_INTER_ATTN_ROWS=1" "FABIND_INTER_ATTN_ROWS=0" --n-prot 100 --steps 30 --warmup 5
tools/ab.sh r5c4/rows_fwd 2 "FABIND_INTER_ATTN_ROWS=1" "FABIND_INTER_ATTN_ROWS=0" --mode fwd --steps 10 --warmup 3
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
FABIND_INTER_ATTN_ROWS=$v rocprofv3 --kernel-trace --stats -d $O/prof$v -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_prof$v.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/prof$v/fb_results.db > $O/kernel_stats_rows$v.txt 2>&1; rm -rf $O/prof$v
echo "== FABIND_INTER_ATTN_ROWS=$v"; grep -i "inter_attn" $O/kernel_stats_rows$v.txt | cut -c1-200; head -3 $O/kernel_stats_rows$v.txt | cut -c1-160
done
