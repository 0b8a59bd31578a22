import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from helpers import load_npz, stack_inputs, weights
from test_gpu_plus import _build
from fabind_amd import engine
dev = torch.device("cuda:0")
engine.set_precision("fp32")
g = load_npz("plus_stack_tiny_grad")
m = _build(g, dev)
inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
Hin = inp["H"].clone().requires_grad_(True)
X, H, Z = m(inp["X"].clone(), Hin, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"])
t = lambda k: torch.from_numpy(g[k]).to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
loss = 0
if which in ("all", "X"): loss = loss + (X * t("cot_X")).sum()
if which in ("all", "H"): loss = loss + (H * t("cot_H")).sum()
if which in ("all", "Z"): loss = loss + (Z * t("cot_Z")).sum()
loss.backward()
print("loss", float(loss), float(g["loss"]))
rows = []
for k, prm in m.named_parameters():
    if "grad_" + k not in g: continue
    ref = g["grad_" + k]
    got = prm.grad.cpu().numpy() if prm.grad is not None else np.zeros_like(ref)
    rows.append((np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12), k, float(np.abs(ref).max())))
rows.sort(reverse=True)
flt = sys.argv[2] if len(sys.argv) > 2 else None
for r in (rows if flt else rows[:25]):
    if not flt or flt in r[1]: print("%.3e  %-70s ref max %.3e" % r)
print("in_H rel err", np.abs(Hin.grad.cpu().numpy() - g["grad_in_H"]).max() / np.abs(g["grad_in_H"]).max())
print("ok (<1e-3):", sum(1 for r in rows if r[0] < 1e-3), "of", len(rows))
for r in rows[-8:]: print("%.3e  %-70s ref max %.3e" % r)
