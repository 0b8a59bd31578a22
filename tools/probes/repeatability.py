"""Probe: run-to-run repeatability of one forward stack pass at the bench shape (B complexes, 1500/40, hidden 512, bf16) with
the MFMA and the fp32 cross-attention kernels; prints max |dX| (A) and max |dH| between run 0 and runs 1..n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine, kernels as K
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
engine.set_precision("bf16")
m = bench.build_model(512, 4, 1).to(dev).eval()
inp = bench.make_batch(B, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}


def run():
    with torch.no_grad():
        X, Hh = m(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"],
                  t["LAS_edge_index"], t["coord_LAS"])
    return X.clone(), Hh.clone()


for mfma in (True, False, True):
    K.CROSS_ATTN_MFMA = mfma
    X0, H0 = run()
    res = []
    for _ in range(4):
        X1, H1 = run()
        res.append((float((X1 - X0).abs().max()) * 5, float((H1 - H0).abs().max())))
    print("B=%d cross attention %s: (max|dX| A, max|dH|) of 4 reruns vs the first:" % (B, "MFMA" if mfma else "fp32 VALU"),
          ["(%.1e, %.1e)" % r for r in res], flush=True)
K.CROSS_ATTN_MFMA = None
