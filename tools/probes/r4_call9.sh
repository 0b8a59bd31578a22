O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c9}; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
python tools/probes/pocket_hostprof2.py > $O/hostprof2.txt 2>&1; head -6 $O/hostprof2.txt
for i in 1 2 3; do
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('model', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
python tools/probes/edge_bwd_h128.py 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profp -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_profp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profp/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
rm -rf $O/profp
head -3 $O/pocket_kernel_stats.txt | cut -c1-150
