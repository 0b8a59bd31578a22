# inter-edge attention kernels inside the headline step: HBM bytes, L2 hits, wave counts, waits (two counter passes)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3y}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/e1 -o e -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/e2 -o e -- python3 $B --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
E=$(for d in e1 e2; do find $O/$d -name "*counter_collection.csv" | head -1; done)
for k in inter_attn_fwd_kernel inter_attn_bwd_a_kernel inter_attn_bwd_b_kernel segment_sum_kernel; do python tools/probes/pmc_table.py "$k, headline step, per launch" $k $E >> $O/inter_attn_pmc.txt 2>&1; done
rm -rf $O/e1 $O/e2
cat $O/inter_attn_pmc.txt
