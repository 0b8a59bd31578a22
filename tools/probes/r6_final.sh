# Round-6 evidence run (one gpurun call): rocprofv3 kernel traces of the default bench step (= BASELINE configs[2] read literally), the
# stack-only step, the pocket-sized step and the production full-model step; live launch groups; HBM bytes (two --pmc passes) and
# matrix-core / VALU utilisation (one pass) of the default step; peak memory with / without the deferred input-gradient GEMMs; the default
# bench line itself.  Everything lands under gpurun_out/$1; the summaries are copied into profiles/ by hand.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6fin}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python3 $B --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profs -o st -- python3 $B --mode fwdbwd --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_profs.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profp -o pk -- python3 $B --mode fwdbwd --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_profp.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profm -o md -- python3 $B --mode model --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_profm.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/fwdbwd_kernel_stats.txt > /dev/null 2>&1
python tools/prof_summary.py $(find $O/profs -name "*.db" | head -1) $O/stack_kernel_stats.txt > /dev/null 2>&1
python tools/prof_summary.py $(find $O/profp -name "*.db" | head -1) $O/pocket_kernel_stats.txt > /dev/null 2>&1
python tools/prof_summary.py $(find $O/profm -name "*.db" | head -1) $O/model_kernel_stats.txt > /dev/null 2>&1
FABIND_BENCH_DUMP_PROFILE=$O/launch_groups.txt python bench.py --no-cpu-baseline --no-extras > $O/bench_groups.json 2>/dev/null
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc.json > $O/pmc_summary.log 2>&1
{ echo "# default step = BASELINE configs[2] read literally (bf16, fwd+bwd, full IaBNet on the whole graph, six-term loss)"; python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1); } > $O/pmc_util.txt 2>&1
for d in 1 0; do FABIND_DEFER_DX=$d python - <<PY >> $O/defer_dx_memory.txt 2>/dev/null
import torch, subprocess, sys, json, os
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "3", "--warmup", "2"]
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
r = json.loads(buf.getvalue().strip().splitlines()[-1])
print("FABIND_DEFER_DX=%s: %.1f complexes/s, peak allocated %.2f GB, peak reserved %.2f GB" % (os.environ["FABIND_DEFER_DX"], r["value"], torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9))
PY
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rm -rf $O/prof $O/profs $O/profp $O/profm $O/pmc_f $O/pmc_w $O/pmc_u
ls $O; head -12 $O/pmc_util.txt; cat $O/defer_dx_memory.txt
