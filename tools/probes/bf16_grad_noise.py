"""Probe: bf16-mode gradient errors on the tiny reference fixture (stack_tiny_grad) with the MFMA and the fp32 cross-attention
kernels, and with / without the shared gradient buffers -- separates kernel noise from host-side changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from helpers import load_npz, stack_inputs
from test_gpu_stack import _build_stack
from fabind_amd import engine, kernels as K, ops
dev = torch.device("cuda:0")
g = load_npz("stack_tiny_grad")
nograd = set(str(s) for s in g["nograd"])


def run(mfma, sinks):
    K.CROSS_ATTN_MFMA = mfma
    orig = ops.shared_grad
    if not sinks:
        ops.shared_grad = lambda t: t
    engine.set_precision("bf16")
    try:
        m = _build_stack(g, dev)
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
        Hin = inp["H"].clone().requires_grad_(True)
        X, H = m(inp["X"].clone(), Hin, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                 inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"])
        ((X * torch.from_numpy(g["cot_X"]).to(dev)).sum() + (H * torch.from_numpy(g["cot_H"]).to(dev)).sum()).backward()
    finally:
        engine.set_precision("fp32")
        ops.shared_grad = orig
        K.CROSS_ATTN_MFMA = None
    rows, num, den = [], 0.0, 0.0
    for n, p in m.named_parameters():
        if n in nograd or p.grad is None:
            continue
        ref = g["grad_" + n]
        d = p.grad.float().cpu().numpy() - ref
        rows.append((float(np.abs(d).max() / np.abs(ref).max()), float(np.linalg.norm(d) / np.linalg.norm(ref)), n))
        num += float((d ** 2).sum()); den += float((ref ** 2).sum())
    rows.sort(reverse=True)
    print("MFMA attention %-5s shared buffers %-5s: whole-gradient l2-rel %.3e; worst max-rel:" % (mfma, sinks, (num / den) ** 0.5),
          ["%.3f %s" % (r[0], r[2].replace("gnn.", "").replace("cross_attn_module.", "cam.")) for r in rows[:4]], flush=True)


for mfma in (False, True):
    for sinks in (False, True):
        run(mfma, sinks)
