"""Where does the bf16 mode's ligand-RMSD gap at the headline shape come from?  The stack runs in 'bf16x3' mode (fp32 storage, split-bf16
contractions: 7e-7 A from the fp32 oracle) with ONE family of operations at a time degraded to what the 'bf16' mode does with it -- bf16
operand roundings of the node-level GEMMs, the fused edge pipeline in bf16 (bf16 AB rows, bf16 LDS tiles), the cross attention on bf16
MFMA tiles -- and the gap of each hybrid is printed next to the pure modes (n_iter = 1, the pass the 1e-4 A gate is stated on).
Run on the GPU box: python tools/probes/precision_sources.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import torch

from fabind_amd import engine, kernels as K, ops, synthetic
from helpers import rmsd
import test_gpu_headline as T

dev = torch.device("cuda:0")
inp = synthetic.make_stack_batch([(T.NP, T.NL)], T.H, seed=7)
lig = inp["mask"].numpy()
m = T._model(1)
Xr, Hr, _, _ = T._oracle(m, inp, 1)
m = m.to(dev)
bf = torch.bfloat16
orig = dict(gemm=K.gemm, fused_edge=ops.fused_edge, split3=K._split3, use_mfma=K.cross_attn_use_mfma)
DEG = {"gemm": False, "edge": False, "attn": False, "gemm_act_only": False, "gemm_w_only": False}


def gemm(A, W, *a, **k):
    """node-level GEMMs with bf16-ROUNDED operands (fp32 storage kept: the rounding is what the bf16 mode adds)"""
    if DEG["gemm"] and A.dtype == torch.float32 and W.dtype == torch.float32 and k.get("groups") is None:
        if not DEG["gemm_w_only"]:
            A = A.to(bf).float()
            if k.get("A2") is not None:
                k["A2"] = k["A2"].to(bf).float()
        if not DEG["gemm_act_only"]:
            W = W.to(bf).float()
    return orig["gemm"](A, W, *a, **k)


def fused_edge(AB, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop=0.0, frags=None):
    if not DEG["edge"]:
        return orig["fused_edge"](AB, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, frags)
    out = K.gcl_edge_fused(AB.to(bf), H, g.row_ctx, g.col_ctx, rhohat, w_r, K.pack_frag(W2), b2, K.pack_frag(Wc), bc, w3, AB.shape[0], 0.0, 0,
                           want16=False, rowptr=g.rp_ctx)
    return out[0], out[1]


K.gemm = gemm
ops.fused_edge = fused_edge


def run(label, mode, **deg):
    for k_ in DEG:
        DEG[k_] = bool(deg.get(k_, False))
    K.cross_attn_use_mfma = (lambda: True) if deg.get("attn") else orig["use_mfma"]
    engine.set_precision(mode)
    try:
        X, Hh, _ = T._hip(m, inp, dev)
    finally:
        engine.set_precision("fp32")
    gap = rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5)
    print("%-72s ligand RMSD gap %.3e A" % (label, gap), flush=True)


run("bf16 mode (the headline dtype)", "bf16")
run("bf16x3 mode (the gate-meeting mode)", "bf16x3")
run("bf16x3 + node-level GEMMs on bf16-rounded operands (activations AND weights)", "bf16x3", gemm=True)
run("bf16x3 + node-level GEMMs: activations rounded only", "bf16x3", gemm=True, gemm_act_only=True)
run("bf16x3 + node-level GEMMs: weights rounded only", "bf16x3", gemm=True, gemm_w_only=True)
run("bf16x3 + fused edge pipeline in bf16 (bf16 AB rows, bf16 tiles, bf16 weights)", "bf16x3", edge=True)
run("bf16x3 + cross attention on bf16 MFMA tiles", "bf16x3", attn=True)
run("bf16x3 + all three", "bf16x3", gemm=True, edge=True, attn=True)
