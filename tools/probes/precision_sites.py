"""Which node-level GEMM sites carry the bf16 mode's ligand-RMSD gap?  (tools/probes/precision_sources.py: bf16 operand roundings of the
node-level GEMMs alone give 6.9e-5 of the 1.0e-4 A; the bf16 edge pipeline 3.5e-6, the bf16 attention tiles 1.3e-6.)  The stack runs in
'bf16x3' mode and ONE site at a time (by parameter-pack name, all layers) gets bf16-rounded operands; then all sites BUT one.
Run on the GPU box: python tools/probes/precision_sites.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import torch

from fabind_amd import engine, kernels as K, synthetic
from helpers import rmsd
import test_gpu_headline as T

dev = torch.device("cuda:0")
inp = synthetic.make_stack_batch([(T.NP, T.NL)], T.H, seed=7)
lig = inp["mask"].numpy()
m = T._model(1)
Xr, Hr, _, _ = T._oracle(m, inp, 1)
m = m.to(dev)
bf = torch.bfloat16
orig_gemm = K.gemm
engine.set_precision("bf16x3")
with torch.no_grad():
    P = engine.prepare_stack_params(m)
names = {}
for key in ("W_in", "W_out", "W_ab0"):
    names[P[key].data_ptr()] = key
for d_ in list(P["gcl"]) + [P["out_layer"]]:
    for key in ("W_ab", "Wn1", "Wn2"):
        names[d_[key].data_ptr()] = "gcl." + key
for d_ in P["att"]:
    for key in ("Wqg_p", "Wkv_p", "Wo_p", "Wqg_c", "Wkv_c", "Wo_c", "Wt1_p", "Wt2_p", "Wt1_c", "Wt2_c", "W_ab32", "Wqkv", "Wc", "Wcomp1"):
        if key in d_ and torch.is_tensor(d_[key]):
            names[d_[key].data_ptr()] = "att." + key
SITES = sorted(set(names.values()))
MODE = {"only": None, "all_but": None}
seen = set()


def gemm(A, W, *a, **k):
    nm = names.get(W.data_ptr())
    seen.add(nm)
    deg = nm is not None and ((MODE["only"] is not None and nm == MODE["only"]) or (MODE["all_but"] is not None and nm != MODE["all_but"]))
    if deg and A.dtype == torch.float32 and W.dtype == torch.float32:
        A = A.to(bf).float()
        if k.get("A2") is not None:
            k["A2"] = k["A2"].to(bf).float()
        W = W.to(bf).float()
    return orig_gemm(A, W, *a, **k)


K.gemm = gemm


def run(label):
    engine.set_precision("bf16x3")
    try:
        X, Hh, _ = T._hip(m, inp, dev)
    finally:
        engine.set_precision("fp32")
    gap = rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5)
    print("%-60s ligand RMSD gap %.3e A" % (label, gap), flush=True)
    return gap


run("bf16x3, nothing degraded")
MODE["all_but"] = "__none__"
run("every named node-level GEMM site on bf16-rounded operands")
print("unnamed GEMM call sites seen:", None in seen)
MODE["all_but"] = None
for s_ in SITES:
    MODE["only"] = s_
    run("only " + s_)
MODE["only"] = None
for s_ in SITES:
    MODE["all_but"] = s_
    run("all but " + s_)
