# counter table of the split-bf16 GEMM next to the bf16 LDS-DMA kernel on the node-level shape M = 98,688, N = K = 512
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3x}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/probes/gemm_x3_one.py
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/e1 -o e -- python3 $P 98688 512 512 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/e2 -o e -- python3 $P 98688 512 512 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/e3 -o e -- python3 $P 98688 512 512 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/e4 -o e -- python3 $P 98688 512 512 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/e5 -o e -- python3 $P 98688 512 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
E=$(for d in e1 e2 e3 e4 e5; do find $O/$d -name "*counter_collection.csv" | head -1; done)
python tools/probes/pmc_table.py "split-bf16 GEMM (128-row tile), M = 98,688, N = K = 512, bias -> fp32, per launch" gemm_x3_kernel $E > $O/gemm_x3_pmc.txt 2>&1
python tools/probes/pmc_table.py "bf16 LDS-DMA GEMM on the same shape, per launch" gemm_bf16_pipe_kernel $E >> $O/gemm_x3_pmc.txt 2>&1
rm -rf $O/e1 $O/e2 $O/e3 $O/e4 $O/e5
cat $O/gemm_x3_pmc.txt
