O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c16}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attn_mfma.py -x -q > $O/tests_a.log 2>&1; grep "fused attention fwd+bwd" $O/tests_a.log | cut -c1-260; tail -3 $O/tests_a.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline fused-train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_ATTN_FUSED_TRAIN=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline tensor path', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket fused-train', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_ATTN_FUSED_TRAIN=0 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket tensor path', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fb_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
rm -rf $O/prof
grep -n "cross_attn\|pair_b\|transpose_pad\|combine\|cfb_" $O/fwdbwd_kernel_stats.txt | cut -c1-170
