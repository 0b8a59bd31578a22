"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, section HBM) -> doubled here.
usage: pmc_summary.py <fetch_counter.csv> <write_counter.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

# kernel-name substring -> substring of the bench.py roofline label it belongs to
KEYS = {
    "gcl_edge_fused_bwd3_kernel": "gcl_edge_fused_bwd3_kernel<512>",
    "gcl_edge_fused_bwd4_kernel": "gcl_edge_fused_bwd4_kernel<512>",
    "gcl_edge_fused2_kernel": "gcl_edge_fused_kernel<512>",
    "gcl_edge_fused_bwd_kernel": "gcl_edge_fused_bwd_kernel<512>",
    "gcl_edge_fused_kernel": "gcl_edge_fused_kernel<512>",
    "gemm_tn_bf16_kernel": "fabind_gemm_tn M=512 N=512 E=15",
    "segment_sum_kernel": "segment_sum",
    "gcl_pre8_kernel": "gcl_pre",
}


def load(path):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return d


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    raw, per = {}, {}
    for (name, grid), fv in f.items():
        for sub, label in KEYS.items():
            if sub in name and (sub != "gcl_edge_fused_kernel" or "bwd" not in name):
                wv = w.get((name, grid), [0.0])
                # launches of different problem shapes can share a grid size (the TN contraction: 4 tiles x 64 splits and 8 tiles x 32
                # splits): keep the cluster of the largest fetch (values within 20 % of the maximum) and the matching share of the writes
                top = [v for v in fv if v >= 0.8 * max(fv)]
                if len(top) < len(fv):
                    wv = sorted(wv)[-len(top):] if len(wv) >= len(top) else wv
                    fv = top
                hbm = (2.0 * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024.0
                raw["%s grid=%d" % (name.split("(")[0][:60], grid)] = dict(
                    launches=len(fv), fetch_KiB_raw=sum(fv) / len(fv), write_KiB=sum(wv) / len(wv), hbm_bytes_per_launch=hbm)
                per[label] = max(per.get(label, 0.0), hbm)          # the largest launch shape of that kernel
    # the node / ligand / pair-level GEMM family (bench.py label "fabind_gemm (all shapes of the step)"): the MEAN over every launch of the
    # LDS-DMA GEMM kernel, like the family's `achieved` (ragged launches share the kernel symbol and are included: < 10 % of them)
    tot, n = 0.0, 0
    for (name, grid), fv in f.items():
        if "gemm_bf16_pipe_kernel" in name:
            wv = w.get((name, grid), [0.0] * len(fv))
            tot += (2.0 * sum(fv) + sum(wv)) * 1024.0
            n += len(fv)
    if n:
        per["fabind_gemm (all shapes"] = tot / n
        raw["gemm_bf16_pipe_kernel (all instances, all grids)"] = dict(launches=n, hbm_bytes_per_launch=tot / n)
    # the split-bf16 GEMM launches (bench.py family "fabind_gemm_x3": every contraction of 'bf16x3', the split-precision sites of 'bf16')
    tot, n = 0.0, 0
    for (name, grid), fv in f.items():
        if "gemm_x3_kernel" in name:
            wv = w.get((name, grid), [0.0] * len(fv))
            tot += (2.0 * sum(fv) + sum(wv)) * 1024.0
            n += len(fv)
    if n:
        per["fabind_gemm_x3 (all shapes"] = tot / n
        raw["gemm_x3_kernel (all instances, all grids)"] = dict(launches=n, hbm_bytes_per_launch=tot / n)
    res = {"note": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (KiB); FETCH_SIZE "
                   "doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B requests at 64 B). bench.py default "
                   "(fwd+bwd, B=64, 1500/40, H=512, bf16).",
           "per_kernel": per, "raw": raw}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1)[:4000])


if __name__ == "__main__":
    main()
