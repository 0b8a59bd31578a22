# Round-5 evidence run (one gpurun call): rocprofv3 kernel traces of the default bench step, the pocket-sized step and the full model,
# live launch groups, HBM bytes (two --pmc passes), matrix-core / VALU utilisation (one pass each for the headline and pocket steps).
# Everything lands under gpurun_out/$1; the summaries are copied into profiles/ by hand.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5fin}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats -d $O/prof -o fwdbwd -- python3 $B --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profp -o pk -- python3 $B --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_profp.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/profm -o md -- python3 $B --mode model --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_profm.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_up -o u -- python3 $B --n-prot 100 --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fwdbwd_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profp/pk_results.db > $O/pocket_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profm/md_results.db > $O/model_kernel_stats.txt 2>&1
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_plus_groups.json 2>/dev/null
FABIND_BENCH_DUMP_PROFILE=$O/launch_groups.txt python bench.py --no-cpu-baseline --no-extras > $O/bench_groups.json 2>/dev/null
FABIND_BENCH_DUMP_PROFILE=$O/pocket_launch_groups.txt python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 > $O/bench_pocket_groups.json 2>/dev/null
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc.json > $O/pmc_summary.log 2>&1
{ echo "# headline step (bf16, fwd+bwd)"; python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1); echo; echo "# pocket-sized step (bf16, fwd+bwd, 100 / 40)"; python tools/pmc_util.py $(find $O/pmc_up -name "*counter_collection.csv" | head -1); } > $O/pmc_util.txt 2>&1
rm -rf $O/prof $O/profp $O/profm $O/pmc_f $O/pmc_w $O/pmc_u $O/pmc_up
ls $O; head -30 $O/pmc_util.txt
