# Round-3 evidence run (one gpurun call): rocprofv3 kernel trace of the default bench step, live launch groups, HBM bytes (two --pmc passes),
# matrix-core / VALU utilisation (one pass, headline step: incl. the cross-attention forward / backward kernels), the same for the
# forward-only step (fused cross attention) and the split-bf16 step, the split-bf16 GEMM microbenchmark.  Everything lands under gpurun_out/$1.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3fin}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats -d $O/prof -o fwdbwd -- python3 $B --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_uf -o u -- python3 $B --mode fwd --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_ux -o u -- python3 $B --precision bf16x3 --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/profx3 -o x3 -- python3 $B --precision bf16x3 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/proffwd -o fwd -- python3 $B --mode fwd --no-cpu-baseline --no-extras --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $B --mode plus_train --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fwdbwd_results.db > $O/fwdbwd_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profx3/x3_results.db > $O/x3_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/proffwd/fwd_results.db > $O/fwd_kernel_stats.txt 2>&1
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
FABIND_BENCH_DUMP_PROFILE=$O/launch_groups.txt python bench.py --no-cpu-baseline --no-extras > $O/bench_groups.json 2>/dev/null
FABIND_BENCH_DUMP_PROFILE=$O/x3_launch_groups.txt python bench.py --precision bf16x3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc.json > $O/pmc_summary.log 2>&1
{ echo "# headline step (bf16, fwd+bwd)"; python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1); echo; echo "# forward-only step (fused cross attention)"; python tools/pmc_util.py $(find $O/pmc_uf -name "*counter_collection.csv" | head -1); echo; echo "# split-bf16 step (bf16x3, fwd+bwd)"; python tools/pmc_util.py $(find $O/pmc_ux -name "*counter_collection.csv" | head -1); } > $O/pmc_util.txt 2>&1
python tools/probes/gemm_x3_bench.py > $O/gemm_x3_bench.txt 2>&1
python tools/probes/attn_fused_phases.py > $O/attn_fused.txt 2>&1
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_plus_train.json 2>/dev/null
python bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_pocket.json 2>/dev/null
python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_fwd.json 2>/dev/null
FABIND_BENCH_PREFETCH=0 python bench.py --mode fwd --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_fwd_serial.json 2>/dev/null
rm -rf $O/prof $O/profx3 $O/proffwd $O/profpt $O/pmc_f $O/pmc_w $O/pmc_u $O/pmc_uf $O/pmc_ux
ls $O; cat $O/pmc_util.txt
