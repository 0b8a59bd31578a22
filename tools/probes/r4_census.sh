# Round-4 first call: where the pocket-sized (100 / 40) fwd+bwd step and the full-model step spend launches and host time.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4census}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B --n-prot 100 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_pocket.json 2>$O/bench_pocket.err
rocprofv3 --kernel-trace --stats -d $O/prof -o pk -- python3 $B --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profm -o md -- python3 $B --mode model --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_profm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profm/md_results.db > $O/model_kernel_stats.txt 2>&1
python tools/probes/pocket_hostprof.py > $O/pocket_hostprof.txt 2>&1
python tools/probes/graph_capture_probe.py > $O/graph_capture_probe.txt 2>&1
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_model.json 2>/dev/null
rm -rf $O/prof $O/profm
ls $O; cat $O/bench_pocket.json; tail -5 $O/graph_capture_probe.txt
