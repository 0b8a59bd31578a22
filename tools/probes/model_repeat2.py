"""Bit-reproducibility of the full-model training step at the shape of tests/test_gpu_dp.py (hidden 512 / 128, 16 ragged pocket-sized
complexes, eval mode, bf16): N forward + backward passes from the same weights, gradients compared with the first pass bit for bit.
usage: model_repeat2.py [passes] [tag]   (run several copies at once to add contention on the device)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d_ in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d_))
from fabind_amd import engine  # noqa: E402
from fabind_amd.models.model import compute_loss  # noqa: E402
import test_gpu_dp as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
engine.set_precision("bf16")
m = T._model(dev)
_, data = T._shard(0, dev)
names = [k for k, p in m.named_parameters() if p.requires_grad]
params = [p for p in m.parameters() if p.requires_grad]
ref = None
for it in range(n):
    for p in params:
        p.grad = None
    dc = data.clone()
    out = m(dc, stage=1, train=True)
    loss, _ = compute_loss(out, dc)
    loss.backward()
    g = [None if p.grad is None else p.grad.detach().clone() for p in params]
    if ref is None:
        ref, ref_out = g, [o.detach().clone() for o in out if torch.is_tensor(o)]
        continue
    bad_o = [i for i, (x, y) in enumerate(zip(ref_out, [o for o in out if torch.is_tensor(o)])) if not torch.equal(x, y.detach())]
    bad = [(k, float((a - b).abs().max()) / max(1e-30, float(a.abs().max()))) for k, a, b in zip(names, ref, g)
           if a is not None and not torch.equal(a, b)]
    print("%s pass %d: loss %.6f; outputs differing %s; %d gradients differ %s" % (tag, it, float(loss.detach()), bad_o, len(bad),
                                                                                    sorted(bad, key=lambda t: -t[1])[:4]), flush=True)
engine.set_precision("fp32")
