O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c26}; mkdir -p $O
for i in 1 2; do
python bench.py --mode model --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model saved-forward H>=256', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_EDGE_SAVE_MIN_H=128 python bench.py --mode model --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model saved-forward H>=128', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_EDGE_SAVE_FWD=0 python bench.py --mode model --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model recompute          ', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
FABIND_EDGE_SAVE_MIN_H=128 rocprofv3 --kernel-trace --stats -d $O/profm -o md -- python3 $GRAFT_REPO_ROOT/bench.py --mode model --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_profm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profm/md_results.db > $O/model_kernel_stats_h128.txt 2>&1; rm -rf $O/profm
head -12 $O/model_kernel_stats_h128.txt | cut -c1-160
