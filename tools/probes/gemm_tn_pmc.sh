# Counter table of the TN (weight-gradient) contraction on the stand-alone bench (four --pmc passes); lands under gpurun_out/tn_pmc/.
O=$GRAFT_REPO_ROOT/gpurun_out/tn_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/probes/gemm_tn_bench.py
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/e1 -o e -- python3 $P 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/e2 -o e -- python3 $P 3 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/e3 -o e -- python3 $P 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/e4 -o e -- python3 $P 3 > /dev/null 2>&1
E=$(for d in e1 e2 e3 e4; do find $O/$d -name "*counter_collection.csv" | head -1; done)
python tools/probes/pmc_table.py "TN contraction, all shapes of tools/probes/gemm_tn_bench.py averaged per launch" gemm_tn_bf16_kernel $E > $O/tn_pmc.txt 2>&1
python - $E > $O/tn_pmc_big.txt 2>&1 <<'PY'
import csv, sys
from collections import defaultdict
v = defaultdict(list)
for p in sys.argv[1:]:
    for r in csv.DictReader(open(p)):
        if "gemm_tn_bf16" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 131072:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("TN contraction, E = 1,539,196, M = N = 512 (grid >= 131072 threads: the edge-level shape and the 8-tile node-level shapes), per launch:")
for k in sorted(v):
    print("  %-32s %.4g  (%d launches)" % (k, sum(v[k]) / len(v[k]), len(v[k])))
PY
cat $O/tn_pmc_big.txt
