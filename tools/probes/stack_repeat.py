"""The headline step (B = 64, 1500 / 40, hidden 512, 4 layers + out layer, bf16 or bf16x3, fwd + bwd) N times from the same weights and
inputs: outputs and every parameter gradient compared with the first pass bit for bit -- run several copies at once for contention.
usage: stack_repeat.py [passes] [tag] [precision] [batch]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fabind_amd import engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tag = sys.argv[2] if len(sys.argv) > 2 else ""
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
engine.set_precision(prec)
m = bench.build_model(512, 4, 1).to(dev).eval()
inp = bench.make_batch(B, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
params = list(m.parameters())
names = [k for k, _ in m.named_parameters()]
ref, nbad = None, 0
for it in range(n):
    for p in params:
        p.grad = None
    X, Hh = m(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"],
              t["coord_LAS"])
    ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
    cur = (X.detach().clone(), Hh.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params])
    if ref is None:
        ref = cur
        continue
    bx, bh = not torch.equal(ref[0], cur[0]), not torch.equal(ref[1], cur[1])
    bg = [(k, "%.2e" % (float((a - b).abs().max()) / max(1e-30, float(a.abs().max())))) for k, a, b in zip(names, ref[2], cur[2])
          if a is not None and not torch.equal(a, b)]
    if bx or bh or bg:
        nbad += 1
        print("%s %s pass %d DIFFERS: X %s (%d elements) H %s (%d elements); %d gradients %s" % (
            tag, prec, it, bx, int((ref[0] != cur[0]).sum()), bh, int((ref[1] != cur[1]).sum()), len(bg), bg[:4]), flush=True)
print("%s %s B=%d: %d of %d passes differ from the first" % (tag, prec, B, nbad, n - 1), flush=True)
engine.set_precision("fp32")
