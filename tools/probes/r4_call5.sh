O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c5}; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline.py -x -q -k "queued_weight or tn_multi or pair_bias or backward_properties or las or segment" > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
python tools/probes/aten_sites.py > $O/aten_sites.txt 2>&1; head -5 $O/aten_sites.txt
python tools/probes/pocket_hostprof2.py > $O/hostprof2.txt 2>&1; head -6 $O/hostprof2.txt
for i in 1 2; do
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_BENCH_ST_BACKWARD=1 python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pocket st-backward', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
rm -rf $O/prof
head -24 $O/pocket_kernel_stats.txt | cut -c1-150
