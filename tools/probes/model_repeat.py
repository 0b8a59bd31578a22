"""Is one full-model training step (IaBNet: pocket model -> crop -> complex model -> heads, six-term loss) bit-reproducible?  Two
forward + backward passes from identical weights and inputs (eval mode: no random draws): every output and every parameter gradient
compared bit for bit; the tensors that differ are listed with their largest difference."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from fabind_amd import engine, synthetic  # noqa: E402
from fabind_amd.models import get_model  # noqa: E402
from fabind_amd.models.model import compute_loss  # noqa: E402
from test_gpu_model import _Logger, _args  # noqa: E402

dev = torch.device("cuda:0")
for prec in ("bf16", "bf16x3", "fp32"):
    a = _args(128, 2, 1)
    a.pocket_pred_hidden_size = 64
    a.random_n_iter = False
    torch.manual_seed(0)
    engine.set_precision(prec)
    m = get_model(a, _Logger(), dev).to(dev).eval()
    base = synthetic.make_hetero_batch([(300, 19), (245, 34), (410, 26), (152, 11), (333, 40), (280, 8)], seed=3).to(dev)
    res = []
    for rep in range(2):
        for p in m.parameters():
            p.grad = None
        data = base.clone()
        out = m(data, stage=1, train=False)
        loss, _ = compute_loss(out, data)
        loss.backward()
        res.append(([o.detach().clone() for o in out if torch.is_tensor(o)], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    bad_o = [i for i, (x, y) in enumerate(zip(res[0][0], res[1][0])) if not torch.equal(x, y)]
    bad_g = [(k, float((res[0][1][k] - res[1][1][k]).abs().max()), float(res[0][1][k].abs().max())) for k in res[0][1]
             if not torch.equal(res[0][1][k], res[1][1][k])]
    print("%s: outputs differing %s; %d of %d parameter gradients differ" % (prec, bad_o, len(bad_g), len(res[0][1])))
    for row in sorted(bad_g, key=lambda r: -r[1] / max(r[2], 1e-30))[:12]:
        print("    %-70s max |diff| %.3e of max %.3e" % row)
engine.set_precision("fp32")
