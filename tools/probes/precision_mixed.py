"""What would running a FEW node-level GEMM sites in split precision buy the bf16 mode?  (tools/probes/precision_sites.py: of the 1.0e-4 A
gap, the q|k|v projection of the inter-edge attention, its coordinate-MLP projection cv = Wc v and the input Linear carry 5.4e-5, 5.7e-5
and 3.2e-5 when they alone run on bf16-rounded operands.)  The bf16 mode is emulated as 'bf16x3' + every named GEMM site on bf16-rounded
operands + the bf16 edge pipeline + the bf16 attention tiles (that emulation reads 1.017e-4 A against the real mode's 1.003e-4); then the
listed sites keep their split-precision contraction.  n_iter = 1, 2, 8.
Run on the GPU box: python tools/probes/precision_mixed.py"""
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
names = {}


def gemm(A, W, *a, **k):
    nm = names.get(W.data_ptr())
    if nm is not None and nm not in KEEP and A.dtype == torch.float32 and W.dtype == torch.float32:
        A = A.to(bf).float()
        if k.get("A2") is not None:
            k["A2"] = k["A2"].to(bf).float()
        W = W.to(bf).float()
    return orig_gemm(A, W, *a, **k)


def fused_edge(AB, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop=0.0, frags=None):
    out = K.gcl_edge_fused(AB.to(bf), H, g.row_ctx, g.col_ctx, rhohat, w_r, K.pack_frag(W2), b2, K.pack_frag(Wc), bc, w3, AB.shape[0], 0.0, 0,
                           want16=False, rowptr=g.rp_ctx)
    return out[0], out[1]


for n_iter in (1, 2, 8):
    m = T._model(n_iter)
    Xr, Hr, _, _ = T._oracle(m, inp, n_iter)
    m = m.to(dev)
    engine.set_precision("bf16")
    X, _, _ = T._hip(m, inp, dev)
    print("n_iter=%d  %-66s gap %.3e A" % (n_iter, "bf16 mode (real)", rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5)), flush=True)
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
        sets = ((), ("att.Wc",), ("att.Wqkv",), ("att.Wc", "att.Wqkv"), ("att.Wc", "att.Wqkv", "W_in"),
                ("att.Wc", "att.Wqkv", "W_in", "att.Wt1_p", "att.Wt2_p", "gcl.Wn2"),
                ("att.Wc", "att.Wqkv", "W_in", "att.Wt1_p", "att.Wt2_p", "gcl.Wn2", "gcl.Wn1", "att.Wo_p", "att.Wkv_c", "att.Wkv_p"))
        if os.environ.get("PROBE_FP32_A_SITES"):       # only sites whose activation operand exists in fp32 in the bf16 mode (residual stream, q|k|v, agg, attention output)
            L1 = ("att.Wc", "att.Wqkv", "W_in")
            sets = (L1, L1 + ("att.Wt1_p",), L1 + ("att.Wt1_p", "gcl.Wn1"), L1 + ("att.Wt1_p", "gcl.Wn1", "att.Wo_p", "att.Wkv_c", "att.Wkv_p"),
                    L1 + ("att.Wt1_p", "gcl.Wn1", "att.Wo_p", "att.Wkv_c", "att.Wkv_p", "gcl.W_ab", "att.Wqg_p", "att.W_ab32", "W_out"))
        for keep in sets:
            KEEP.clear()
            KEEP.update(keep)
            X, _, _ = T._hip(m, inp, dev)
            print("n_iter=%d  emulated bf16, split precision kept at %-38s gap %.3e A" % (n_iter, "{" + ", ".join(k_.split(".")[-1] for k_ in keep) + "}",
                                                                                    rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5)), flush=True)
    finally:
        K.gemm, ops.fused_edge, K.cross_attn_use_mfma = orig_gemm, orig_edge, orig_mfma
        engine.set_precision("fp32")
