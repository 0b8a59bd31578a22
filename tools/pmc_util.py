"""Matrix-core / VALU utilisation per kernel from one rocprofv3 --pmc pass (csv) with SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA,
SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.  util = busy SIMD-cycles / (1024 SIMDs x kernel cycles), the kernel's
cycles taken as GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs; checked against duration x clock).
SQ_ACTIVE_INST_VALU is in units of 4 cycles.  usage: pmc_util.py <counter_collection.csv> [out.txt]"""
import csv
import sys
from collections import defaultdict

KEEP = ("cross_attn_mfma", "cross_attn_fused", "gemm_x3", "pair_update_fused", "gcl_edge_fused", "gemm_bf16_pipe", "gemm_tn_bf16", "inter_attn", "cross_attn_fwd", "cross_attn_bwd")


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0]
        if any(k in name for k in KEEP):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["%-64s %7s %10s %10s %12s" % ("kernel", "calls", "MFMA util", "VALU util", "MFMA insts")]
    for name, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        if not n:
            continue
        cyc = sum(c["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0
        mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / cyc
        va = 4.0 * sum(c.get("SQ_ACTIVE_INST_VALU", [0])) / cyc
        lines.append("%-64s %7d %9.1f%% %9.1f%% %12.3g" % (name[:64], n, 100 * mf, 100 * va, sum(c.get("SQ_INSTS_MFMA", [0])) / n))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
