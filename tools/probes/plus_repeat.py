"""FABind+ counterpart of model_repeat.py: two training steps of FABindPlus (eval mode: no random draws; 7-term loss with the
permutation-invariant term) from identical weights and inputs, every output and parameter gradient compared bit for bit."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d_ in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d_))
from fabind_amd import engine, synthetic  # noqa: E402
from fabind_amd.plus.models import compute_loss, get_model  # noqa: E402
from test_gpu_plus import _Logger, _args  # noqa: E402

dev = torch.device("cuda:0")
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tag = sys.argv[2] if len(sys.argv) > 2 else ""
for prec in (("bf16", "bf16x3") if n_pass == 2 else ("bf16",)):
    a = _args(128, 2, 1)
    for k, v in dict(pocket_pred_hidden_size=64, pocket_pred_layers=1, pocket_pred_n_iter=1, random_n_iter=False, use_for_radius_pred="ligand",
                     dis_map_thres=15.0, pocket_radius_buffer=5.0, min_pocket_radius=20.0, force_fix_radius=False, use_clustering=False,
                     gs_tau=1.0, gs_hard=False, pocket_radius=20.0, train_pred_pocket_noise=0.0, local_eval=False).items():
        setattr(a, k, v)
    torch.manual_seed(0)
    engine.set_precision(prec)
    m = get_model(a, _Logger()).to(dev).eval()
    sizes = [(300, 19), (245, 34), (410, 26), (152, 11), (333, 40), (280, 8)]
    base = synthetic.make_hetero_batch(sizes, seed=3).to(dev)
    radius = torch.tensor([6.0, 7.0, 5.0, 6.5, 6.0, 5.5], device=dev)
    num_atoms = [s[1] for s in sizes]
    isos = [[list(range(n)), list(reversed(range(n)))] for n in num_atoms]
    res = []
    for rep in range(n_pass):
        for p in m.parameters():
            p.grad = None
        data = base.clone()
        data.ligand_radius, data.num_atoms, data.isomorphisms = radius, num_atoms, isos
        out = m(data, train=False)
        loss, _ = compute_loss(out, data)
        loss.backward()
        res.append(([o.detach().clone() for o in out if torch.is_tensor(o)], float(loss),
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    nbad = 0
    for rep in range(1, n_pass):
        bad_o = [i for i, (x, y) in enumerate(zip(res[0][0], res[rep][0])) if not torch.equal(x, y)]
        bad_g = [(k, float((res[0][2][k] - res[rep][2][k]).abs().max()), float(res[0][2][k].abs().max())) for k in res[0][2]
                 if not torch.equal(res[0][2][k], res[rep][2][k])]
        if bad_o or bad_g or n_pass == 2:
            nbad += bool(bad_o or bad_g)
            print("%s %s pass %d: loss equal %s; outputs differing %s; %d of %d parameter gradients differ" % (tag, prec, rep, res[0][1] == res[rep][1], bad_o, len(bad_g), len(res[0][2])), flush=True)
            for row in sorted(bad_g, key=lambda r: -r[1] / max(r[2], 1e-30))[:6]:
                print("    %-70s max |diff| %.3e of max %.3e" % row)
    print("%s %s: %d of %d passes differ from the first" % (tag, prec, nbad, n_pass - 1), flush=True)
engine.set_precision("fp32")
