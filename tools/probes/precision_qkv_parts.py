"""Which columns of the inter-edge attention's q | k | v projection need the split-precision contraction?  Emulated bf16 mode (see
precision_mixed.py) with the level-2 sites kept, and the q | k | v site kept only for q, k, v column groups in turn (n_iter 1, 2, 8)."""
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
bf = torch.bfloat16
orig_gemm, orig_edge, orig_mfma = K.gemm, ops.fused_edge, K.cross_attn_use_mfma
KEEP = set()
QKV = {"cols": None}
names = {}
H = T.H


def gemm(A, W, *a, **k):
    nm = names.get(W.data_ptr())
    if nm == "att.Wqkv" and QKV["cols"] is not None and A.dtype == torch.float32:
        full = orig_gemm(A, W, *a, **k)
        kd = dict(k)
        kd.pop("out16", None)
        deg = orig_gemm(A.to(bf).float(), W.to(bf).float(), *a, **kd)
        lo, hi = QKV["cols"]
        out = deg[0].clone()
        out[:, lo:hi] = full[0][:, lo:hi]
        if k.get("out16") is not None:
            k["out16"].copy_(out.to(bf))
        return out, None
    if nm is not None and nm not in KEEP and A.dtype == torch.float32 and W.dtype == torch.float32:
        A = A.to(bf).float()
        if k.get("A2") is not None:
            k["A2"] = k["A2"].to(bf).float()
        W = W.to(bf).float()
    return orig_gemm(A, W, *a, **k)


def fused_edge(AB, rhohat, w_r, W2, b2, Wc, bc, w3, H_, g, p_drop=0.0, frags=None):
    out = K.gcl_edge_fused(AB.to(bf), H_, g.row_ctx, g.col_ctx, rhohat, w_r, K.pack_frag(W2), b2, K.pack_frag(Wc), bc, w3, AB.shape[0], 0.0, 0,
                           want16=False, rowptr=g.rp_ctx)
    return out[0], out[1]


L2 = ("att.Wc", "W_in", "att.Wo_p", "att.Wkv_c", "att.Wkv_p")
for n_iter in (1, 2, 8):
    m = T._model(n_iter)
    Xr, Hr, _, _ = T._oracle(m, inp, n_iter)
    m = m.to(dev)
    engine.set_precision("bf16x3")
    with torch.no_grad():
        P = engine.prepare_stack_params(m)
    names.clear()
    for key in ("W_in", "W_out", "W_ab0"):
        names[P[key].data_ptr()] = key
    for d_ in list(P["gcl"]) + [P["out_layer"]]:
        for key in ("W_ab", "Wn1", "Wn2"):
            names[d_[key].data_ptr()] = "gcl." + key
    for d_ in P["att"]:
        for key in ("Wqg_p", "Wkv_p", "Wo_p", "Wqg_c", "Wkv_c", "Wo_c", "Wt1_p", "Wt2_p", "Wt1_c", "Wt2_c", "W_ab32", "Wqkv", "Wc", "Wcomp1"):
            if key in d_ and torch.is_tensor(d_[key]):
                names[d_[key].data_ptr()] = "att." + key
    K.gemm, ops.fused_edge, K.cross_attn_use_mfma = gemm, fused_edge, (lambda: True)
    try:
        for label, keep, cols in (("level 2 (q | k | v all split)", L2 + ("att.Wqkv",), None), ("level 2 without the q | k | v site", L2, None),
                                  ("level 2, only q split", L2, (0, H)), ("level 2, only k split", L2, (H, 2 * H)), ("level 2, only v split", L2, (2 * H, 3 * H)),
                                  ("level 2, q and k split", L2, (0, 2 * H)), ("level 2, k and v split", L2, (H, 3 * H))):
            KEEP.clear()
            KEEP.update(keep)
            QKV["cols"] = cols
            X, _, _ = T._hip(m, inp, dev)
            print("n_iter=%d  emulated bf16, %-40s gap %.3e A" % (n_iter, label, rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5)), flush=True)
    finally:
        K.gemm, ops.fused_edge, K.cross_attn_use_mfma = orig_gemm, orig_edge, orig_mfma
        QKV["cols"] = None
        engine.set_precision("fp32")
