# Round-end evidence run (one gpurun call): rocprofv3 kernel trace of the default bench, live launch groups, HBM bytes (two --pmc passes),
# matrix-core / VALU utilisation (one pass), the counter table of the fused edge kernels (four passes), phase counters, skip-mask
# sensitivity and the two microbenchmarks.  Everything lands under gpurun_out/r2g/.
O=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats -d $O/prof -o fwdbwd -- python3 $B --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_u -o u -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
P=$GRAFT_REPO_ROOT/tools/probes/edge_bwd_one.py
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/e1 -o e -- python3 $P 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/e2 -o e -- python3 $P 5 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/e3 -o e -- python3 $P 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/e4 -o e -- python3 $P 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/prof/fwdbwd_results.db > $O/kernel_stats.txt 2>&1
FABIND_BENCH_DUMP_PROFILE=$O/launch_groups.txt python bench.py --no-cpu-baseline --no-extras > $O/bench_groups.json 2>/dev/null
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc.json > $O/pmc_summary.log 2>&1
python tools/pmc_util.py $(find $O/pmc_u -name "*counter_collection.csv" | head -1) > $O/pmc_util.txt 2>&1
E=$(for d in e1 e2 e3 e4; do find $O/$d -name "*counter_collection.csv" | head -1; done)
python tools/probes/pmc_table.py "fused edge backward, store-wave form (variant 5, 256 work-groups of 9 waves), per launch, E = 1,539,196 edges, H = 512" gcl_edge_fused_bwd3_kernel $E > $O/edge_pmc.txt 2>&1
python tools/probes/pmc_table.py "fused edge forward, second form (two work-groups per CU), per launch, same shape" gcl_edge_fused2_kernel $E >> $O/edge_pmc.txt 2>&1
python tools/edge_bwd_phases.py 64 5 > $O/phases_v5.txt 2>&1
python tools/edge_bwd_phases.py 64 0 > $O/phases_v0.txt 2>&1
MASK_ROUNDS=0 python tools/probes/edge_bwd2_sensitivity.py 5 2>&1 | grep round > $O/sensitivity_v5.txt
python tools/probes/edge_bwd_variants.py 2>&1 | grep -v amdgpu | grep "^round\|vs 0\|forward variant" > $O/variants.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/probes/store_rate.hip 2>/dev/null && /tmp/store_rate > $O/store_rate.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probes/mfma_peak.hip 2>/dev/null && /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
python tools/hbm_bench.py > $O/hbm_kernels.txt 2>&1
ls $O
