"""GPU: the one-launch parameter pack (fabind_amd/param_pack.py + fabind_multi_copy) against the same requests executed with torch
ops: every packed tensor bit-equal, every parameter gradient equal for random cotangents, and no parameter element requested twice."""
import os

import pytest
import torch

from test_gpu_stack import _args

pytestmark = pytest.mark.gpu


def _flat(P, prefix=""):
    out = {}
    for k, v in (P.items() if isinstance(P, dict) else enumerate(P)):
        name = "%s%s" % (prefix, k)
        if isinstance(v, (dict, list)):
            out.update(_flat(v, name + "."))
        elif torch.is_tensor(v):
            out[name] = v
    return out


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_param_pack_matches_torch_ops(precision):
    from fabind_amd import engine
    from fabind_amd.models.att_model import EfficientMCAttModel
    dev = torch.device("cuda:0")
    engine.set_precision(precision)
    try:
        torch.manual_seed(3)
        m = EfficientMCAttModel(_args(64, 2, 1), 64, 64, 1, n_layers=2, n_iter=1, normalize_coord=lambda x: x / 5.0,
                                unnormalize_coord=lambda x: x * 5.0).to(dev)
        params = [p for p in m.parameters() if p.requires_grad]
        res = {}
        for mode in ("check", "0"):                                   # one launch (with the overlap check) / torch ops
            os.environ["FABIND_PARAM_PACK"] = mode
            for p in params:
                p.grad = None
            flat = _flat(engine._build_stack_params(m))
            g = torch.Generator(device="cpu").manual_seed(11)
            loss = sum((v.float() * torch.randn(v.shape, generator=g).to(dev)).sum() for k, v in sorted(flat.items()) if v.requires_grad)
            loss.backward()
            res[mode] = ({k: v.detach().clone() for k, v in flat.items()},
                         [None if p.grad is None else p.grad.detach().clone() for p in params])
        assert sorted(res["check"][0]) == sorted(res["0"][0])
        for k, v in res["0"][0].items():
            w = res["check"][0][k]
            assert w.dtype == v.dtype and w.shape == v.shape and torch.equal(w, v), k
        worst = 0.0
        for ga, gb in zip(res["check"][1], res["0"][1]):
            assert (ga is None) == (gb is None)
            if ga is not None:
                worst = max(worst, float((ga - gb).abs().max() / gb.abs().max().clamp_min(1e-30)))
        print("parameter pack, %s: %d tensors bit-equal, parameter gradients max rel diff %.2e" % (precision, len(res["0"][0]), worst))
        assert worst <= 1e-6
    finally:
        os.environ.pop("FABIND_PARAM_PACK", None)
        engine.set_precision("bf16")


def test_kept_requests_follow_the_parameters_and_transposes_match():
    """The requests of a model's pack are planned once and re-run every call (engine._build_stack_params): a second call after an
    in-place parameter update returns the updated values, a parameter whose storage was replaced rebuilds the plan, gradients of the
    re-run pack equal those of a freshly planned one, and every W^T written next to a W (`_fab_T`: the LDS-tiled transpose path
    of fabind_multi_copy) is the transpose of that W."""
    from fabind_amd import engine
    from fabind_amd.models.att_model import EfficientMCAttModel
    dev = torch.device("cuda:0")
    engine.set_precision("bf16")
    torch.manual_seed(5)
    m = EfficientMCAttModel(_args(64, 2, 1), 64, 64, 1, n_layers=2, n_iter=1, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).to(dev)
    params = [p for p in m.parameters() if p.requires_grad]

    def build():
        for p in params:
            p.grad = None
        flat = _flat(engine._build_stack_params(m))
        g = torch.Generator(device="cpu").manual_seed(11)
        loss = sum((v.float() * torch.randn(v.shape, generator=g).to(dev)).sum() for k, v in sorted(flat.items()) if v.requires_grad)
        loss.backward()
        return flat, [None if p.grad is None else p.grad.detach().clone() for p in params]
    engine._PLAN_CACHE.clear()
    f1, g1 = build()                                                  # plans
    n_T = 0
    for k, v in f1.items():
        t = getattr(v, "_fab_T", None)
        if t is not None:
            n_T += 1
            assert torch.equal(t, v.detach().t().contiguous()), k
    assert n_T >= 10
    ent = engine._PLAN_CACHE[id(m)]
    f2, g2 = build()                                                  # re-runs the kept plan
    assert engine._PLAN_CACHE[id(m)][1] is ent[1]
    for k in f1:
        assert torch.equal(f1[k], f2[k]), k
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    w = m.gnn.gcl_0.edge_mlp[2].weight
    with torch.no_grad():
        w.add_(1.0)                                                   # in place: same plan, new values
    f3, _ = build()
    assert engine._PLAN_CACHE[id(m)][1] is ent[1]
    assert torch.equal(f3["gcl.0.W2"].float(), w.detach().to(torch.bfloat16).float())
    w.data = w.data.clone()                                           # replaced storage: the kept views are stale -> new plan
    with torch.no_grad():
        w.mul_(0.5)
    f4, _ = build()
    assert engine._PLAN_CACHE[id(m)][1] is not ent[1]
    assert torch.equal(f4["gcl.0.W2"].float(), w.detach().to(torch.bfloat16).float())
