"""PROBE (test-side script, not collected by pytest; run as `python tests/probe_retained_graph_spread.py` on a GPU box):
how far do two backward passes over the SAME retained graph differ?  (Float atomics in some adjoints reorder sums
between runs.)  Prints, per precision, the largest |first - second| over all gradients relative to (a) that tensor's own max and
(b) the largest gradient max of the model, over 20 repetitions -- the basis of the tolerance in
tests/test_gpu_stack.py::test_backward_twice_over_a_retained_graph_gives_the_same_gradients."""
import sys

import torch

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from test_gpu_stack import _build_stack, load_npz, stack_inputs  # noqa: E402
from fabind_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
g = load_npz("stack_tiny_grad")
for prec in ("fp32", "bf16"):
    engine.set_precision(prec)
    worst_own, worst_glob, worst_name = 0.0, 0.0, None
    for rep in range(20):
        m = _build_stack(g, dev)
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
        Hin = inp["H"].clone().requires_grad_(True)
        X, H = m(inp["X"].clone(), Hin, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                 inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"])
        loss = (X * torch.from_numpy(g["cot_X"]).to(dev)).sum() + (H * torch.from_numpy(g["cot_H"]).to(dev)).sum()
        named = [("input_H", Hin)] + [(n, p) for n, p in m.named_parameters() if p.requires_grad]
        ts = [t for _, t in named]
        first = torch.autograd.grad(loss, ts, retain_graph=True, allow_unused=True)
        second = torch.autograd.grad(loss, ts, retain_graph=False, allow_unused=True)
        gmax = max(float(a.abs().max()) for a in first if a is not None)
        for (n, _), a, b in zip(named, first, second):
            if a is None:
                continue
            d = float((a - b).abs().max())
            own = d / max(float(a.abs().max()), 1e-30)
            if own > worst_own:
                worst_own, worst_name = own, n
            worst_glob = max(worst_glob, d / gmax)
    print("RGS %s: worst |first-second| / own max = %.3e (%s);  / largest gradient max = %.3e" % (prec, worst_own, worst_name, worst_glob))
engine.set_precision("fp32")
