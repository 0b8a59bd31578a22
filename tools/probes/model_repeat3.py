"""Hunt for the rare, contention-only mismatch of the full-model step (tools/probes/model_repeat2.py under three concurrent copies):
every pass records the per-layer outputs of the LAST stack call (engine.DEBUG_CAPTURE: the complex model), the model outputs and the
gradients; a pass that differs from the first one lists the captured tensors / outputs that differ, in layer order.
usage: model_repeat3.py [passes] [tag]"""
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
params = [p for p in m.parameters() if p.requires_grad]
ref = None
nbad = 0
for it in range(n):
    for p in params:
        p.grad = None
    dc = data.clone()
    engine.DEBUG_CAPTURE = {}
    out = m(dc, stage=1, train=True)
    cap = dict(engine.DEBUG_CAPTURE)
    engine.DEBUG_CAPTURE = None
    loss, _ = compute_loss(out, dc)
    loss.backward()
    cur = (cap, [o.detach().clone() for o in out if torch.is_tensor(o)])
    if ref is None:
        ref = cur
        continue
    badc = [k for k in ref[0] if not torch.equal(ref[0][k], cur[0][k])]
    bado = [i for i, (x, y) in enumerate(zip(ref[1], cur[1])) if not torch.equal(x, y)]
    if badc or bado:
        nbad += 1
        det = []
        for k in badc[:6]:
            d = (ref[0][k].float() - cur[0][k].float()).abs()
            det.append((k, "%.2e" % float(d.max()), int((d > 0).sum()), tuple(cur[0][k].shape)))
        for i in bado[:4]:
            d = (ref[1][i].float() - cur[1][i].float()).abs()
            det.append(("out%d" % i, "%.2e" % float(d.max()), int((d > 0).sum()), tuple(cur[1][i].shape)))
        print("%s pass %d DIFFERS: captured %s; outputs %s; %s" % (tag, it, badc, bado, det), flush=True)
print("%s: %d of %d passes differ from the first" % (tag, nbad, n - 1), flush=True)
engine.set_precision("fp32")
