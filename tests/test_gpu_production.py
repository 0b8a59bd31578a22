"""GPU: parity AT PRODUCTION SIZE for everything bench.py quotes besides the v1 stack headline (VERDICT r2 "missing" item 2;
the headline stack itself: test_gpu_headline.py).  The reference-run fixtures under tests/golden are hidden <= 128 / <= 260 nodes;
here the CPU oracle (oracle/fabind_oracle.py, oracle/fabind_plus_oracle.py -- pinned to the reference by test_oracle_golden.py) is
run on the GPU box's host at the bench sizes and the HIP path must agree with it:

(i)   full IaBNet, hidden 512 / pocket 128, 4 layers, n_iter = 8, two ragged complexes with 1500- and 1100-residue proteins ->
      pocket crop; stage 1 and stage 2: the 11-tuple, the six-term loss (1e-5 rel) and every parameter gradient against autograd
      through the oracle (FABind/fabind/models/model.py:82-369, main_fabind.py:398-417);
(ii)  FABind+, hidden 512, 5 layers: the stack at 1500 / 40 nodes (X, H and the threaded pair embedding) and the full model
      (13-tuple, seven-term loss with the permutation-invariant term, parameter gradients)
      (FABind_plus/fabind/models/model.py:63-401, att_model.py:165-223, utils/training.py:61-97).

Gates (fp32 mode): coordinates 1e-4 A RMSD, tuple members 1e-4 of their maximum, losses 1e-5 rel, gradients 3e-3 of each tensor's
maximum.  The same inputs in bf16: the gap is printed and bounded at 2x the first measurement (constants below)."""
import numpy as np
import pytest
import torch

import fabind_oracle as orc
import fabind_plus_oracle as porc
from helpers import rmsd
from test_gpu_stack import _args
from test_gpu_plus import _args as _plus_args

pytestmark = pytest.mark.gpu
SIZES = [(1500, 40), (1100, 27)]


class _Logger:
    def log_message(self, s):
        pass


def _threads():
    torch.set_num_threads(min(torch.get_num_threads(), 32))            # the measured best on the GPU box's host


def _grad_rows(named_params, ref_grads):
    """[(error / tolerance scale, name, own max / largest max, l2-rel)] over EVERY parameter the oracle's autograd reaches.  The error
    of a tensor is its largest entry error over (its own largest entry + 1e-6 x the model's largest gradient entry): the second term is
    the fp32 round-off (8 eps) a tensor a million times smaller than the largest one inherits from shared upstream sums (the
    model's gradient entries span 1e-7 .. 45 here).  A parameter the oracle gives no gradient (the 33 never-used tensors) must not get one."""
    gmax = max(float(v.abs().max()) for v in ref_grads.values() if v is not None)
    rows = []
    for n, p in named_params:
        r = ref_grads.get(n)
        if r is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        rmax = float(r.abs().max())
        if p.grad is None:                                   # exact zeros by symmetry (softmax-shift-invariant biases): round-off in the oracle
            assert rmax <= 1e-6 * gmax, (n, rmax, gmax)
            continue
        d = p.grad.detach().cpu() - r
        rows.append((float(d.abs().max()) / (rmax + 1e-6 * gmax), n, rmax / gmax, float(d.norm() / max(float(r.norm()), 1e-30))))
    rows.sort(reverse=True)
    return rows


def _grads_ok(rows, tol=3e-3):
    """Every tensor within `tol` of its largest entry -- or, for the few tensors behind a ReLU, within 1e-2 with an l2 error below 1e-3:
    a ReLU unit whose pre-activation is within fp32 round-off of zero takes the other branch in one of the two implementations, and one
    such unit moves single entries of the weight gradient in front of it by a few 1e-3 of the tensor's maximum (seen in both directions
    between builds whose only difference was the summation order inside LayerNorm)."""
    return all(r[0] <= tol or (r[0] <= 1e-2 and r[3] <= 1e-3) for r in rows)


def _print_rows(title, rows):
    print("%s: %d parameter tensors compared; worst (max-rel, name, |ref|max / largest, l2-rel):" % (title, len(rows)))
    for r_ in rows[:5]:
        print("    %.3e  %s  %.2e  %.3e" % r_)
    l2 = sorted(r[3] for r in rows if r[2] >= 1e-6)        # (tensors at the round-off floor have no meaningful relative error)
    print("    per-tensor l2-rel over the %d tensors above 1e-6 of the largest gradient: median %.3e, worst %.3e" % (len(l2), l2[len(l2) // 2], l2[-1]))


# ------------------------------------------------------------------------------------------------------------------------------
# (i) full IaBNet
# ------------------------------------------------------------------------------------------------------------------------------
def _iabnet(n_iter=8):
    from fabind_amd import synthetic
    from fabind_amd.models import get_model
    a = _args(512, 4, n_iter)
    a.pocket_pred_hidden_size = 128
    a.random_n_iter = False
    torch.manual_seed(0)
    m = get_model(a, _Logger(), None).eval()
    synthetic.condition_for_large_graphs(m)
    return synthetic.condition_model_inputs(m)


_IAB = {}


def _iabnet_oracle(stage):
    """Oracle forward + six-term loss + autograd at production size, once per stage (15 s + 3 s on the host)."""
    if stage not in _IAB:
        from fabind_amd import synthetic
        _threads()
        m = _iabnet()
        data = synthetic.make_hetero_batch(SIZES, seed=11)
        cfg = dict(orc.DEFAULT_CFG)
        cfg.update(mean_layers=4, n_iter=8)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
        out = orc.model_forward(sd, cfg, data.clone(), stage=stage)
        loss, terms = orc.compute_loss(out, data)
        loss.backward()
        _IAB[stage] = dict(out=[o.detach() if torch.is_tensor(o) else o for o in out], loss=float(loss.detach()),
                           terms={k: float(v.detach()) for k, v in terms.items()},
                           grads={k: v.grad for k, v in sd.items() if v.is_floating_point()}, data=data)
    return _IAB[stage]


TUPLE = ((2, "y_pred"), (3, "y_pred_by_coords"), (4, "pocket_cls_pred"), (8, "pred_pocket_center"), (9, "dis_map"))
# bf16 (split-precision sites level 3, round 6) at production size and n_iter 8: stage 1 is asserted AT THE GATE (1e-4 A; measured 9.1e-5);
# stage 2 -- the ligand travels 17 A from the predicted pocket centre -- reads 1.3e-4 A: the gate is missed by 1.3x there, bounded at 2x the
# measurement; 'bf16x3' (asserted at the gate below) is the mode bench.py quotes for that loop.  (Round 5, level 2: 5.9e-4 / 6.4e-4 A.)
IAB_BF16_BOUND = {1: (1e-4, 1e-5), 2: (2.7e-4, 1e-5)}


@pytest.mark.parametrize("stage", [1, 2])
def test_iabnet_production_size_matches_oracle(stage):
    from fabind_amd import engine
    from fabind_amd.models.model import compute_loss
    dev = torch.device("cuda:0")
    ref = _iabnet_oracle(stage)
    engine.set_precision("fp32")
    m = _iabnet().to(dev)
    data = ref["data"].clone().to(dev)
    out = m(data, stage=stage, train=False)
    init = ref["data"]["compound"].node_coords.numpy()
    moved = rmsd(ref["out"][0].numpy(), init)
    gap = rmsd(out[0].detach().cpu().numpy(), ref["out"][0].numpy())
    print("IaBNet 512/128, n_iter 8, stage %d, proteins of %s residues: ligand moved %.3f A; RMSD vs oracle %.3e A"
          % (stage, [s[0] for s in SIZES], moved, gap))
    assert moved > 1e-2 and gap < 1e-4
    for i, n in TUPLE:
        r, g = ref["out"][i].numpy(), out[i].detach().cpu().numpy()
        assert r.shape == g.shape, n
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), n
    assert np.array_equal(out[5].cpu().numpy(), ref["out"][5].numpy())
    assert np.array_equal(out[6].cpu().numpy(), ref["out"][6].numpy())
    assert int(out[10]) == int(ref["out"][10])
    loss, terms = compute_loss(out, data)
    lerr = abs(float(loss.detach()) - ref["loss"]) / abs(ref["loss"])
    print("    six-term loss %.6f vs oracle %.6f (rel %.2e); terms:" % (float(loss.detach()), ref["loss"], lerr),
          {k: "%.2e" % (abs(float(v.detach()) - ref["terms"][k]) / max(abs(ref["terms"][k]), 1e-2)) for k, v in terms.items()})
    assert lerr <= 1e-5
    for k, v in terms.items():
        assert abs(float(v.detach()) - ref["terms"][k]) <= 1e-5 * max(abs(ref["terms"][k]), 1e-2), k
    loss.backward()
    rows = _grad_rows(m.named_parameters(), ref["grads"])
    _print_rows("IaBNet production size, stage %d, fp32 gradients vs oracle autograd" % stage, rows)
    assert len(rows) >= 350 and _grads_ok(rows), (len(rows), rows[0])

    # the same input in the bench dtype
    engine.set_precision("bf16")
    try:
        data = ref["data"].clone().to(dev)
        with torch.no_grad():
            o16 = m(data, stage=stage, train=False)
            l16, _ = compute_loss(o16, data)
    finally:
        engine.set_precision("fp32")
    g16 = rmsd(o16[0].cpu().numpy(), ref["out"][0].numpy())
    le16 = abs(float(l16) - ref["loss"]) / abs(ref["loss"])
    print("    bf16: ligand RMSD gap %.3e A (gate 1e-4: %s), loss rel gap %.3e" % (g16, "met" if g16 < 1e-4 else "missed", le16))
    assert g16 < IAB_BF16_BOUND[stage][0] and le16 < IAB_BF16_BOUND[stage][1]

    # ... and in the gate-meeting fast mode (split bf16): the north_star gates themselves, 1e-4 A and 1e-5 relative loss
    engine.set_precision("bf16x3")
    try:
        data = ref["data"].clone().to(dev)
        with torch.no_grad():
            o3 = m(data, stage=stage, train=False)
            l3, _ = compute_loss(o3, data)
    finally:
        engine.set_precision("fp32")
    g3 = rmsd(o3[0].cpu().numpy(), ref["out"][0].numpy())
    le3 = abs(float(l3) - ref["loss"]) / abs(ref["loss"])
    print("    bf16x3: ligand RMSD gap %.3e A, loss rel gap %.3e (gates 1e-4 A / 1e-5)" % (g3, le3))
    assert g3 < 1e-4 and le3 <= 1e-5

    # ... and with the mode's bf16-edge option (config.set_x3_edge("bf16"), round 5): the same gates, less margin (measured 5.1e-5 / 2.0e-5 A)
    from fabind_amd import config
    engine.set_precision("bf16x3")
    config.set_x3_edge("bf16")
    try:
        data = ref["data"].clone().to(dev)
        with torch.no_grad():
            o3e = m(data, stage=stage, train=False)
            l3e, _ = compute_loss(o3e, data)
    finally:
        config.set_x3_edge("split")
        engine.set_precision("fp32")
    g3e = rmsd(o3e[0].cpu().numpy(), ref["out"][0].numpy())
    le3e = abs(float(l3e) - ref["loss"]) / abs(ref["loss"])
    print("    bf16x3 + bf16 edge pipeline: ligand RMSD gap %.3e A, loss rel gap %.3e (gates 1e-4 A / 1e-5)" % (g3e, le3e))
    assert g3e < 1e-4 and le3e <= 1e-5


# bf16 gap of the whole-graph step = bench.py's HEADLINE (ligand RMSD in A, relative loss gap): the north-star gates themselves
CONFIG3_BF16_BOUND = (1e-4, 1e-5)                # measured 1.6e-5 A and 2.0e-7 (round 5, split sites level 2: 9.4e-5 / 5.6e-5 A); whole-gradient l2 error 6.7e-4


def test_config3_whole_graph_matches_oracle():
    """BASELINE configs[2] READ LITERALLY: one 1500 / 40 complex whose pocket is the WHOLE protein
    (`pocket_radius = 1e9`), full IaBNet stage 1 -- the hidden-128 pocket model AND the 4-layer hidden-512 complex model AND the
    distance-map head over all 1500 x 40 pairs together, which no other test compares with the oracle (the production tests above use
    the 20 A crop; the headline tests the stack alone) -- with the pocket-cls + pocket-centre + coord + both distance-map + distill
    losses, forward and backward, n_iter = 1 as in bench.py's `config3_whole_graph` sub-object.  fp32 vs the CPU oracle: 11-tuple,
    six loss terms to 1e-5 relative, every parameter gradient; bf16 (the sub-object's dtype): the gap printed and bounded.
    Reference: FABind/fabind/models/model.py:82-369 (forward), main_fabind.py:398-417 (loss)."""
    from fabind_amd import engine, synthetic
    from fabind_amd.models.model import compute_loss
    _threads()
    dev = torch.device("cuda:0")
    data0 = synthetic.make_hetero_batch([(1500, 40)], seed=12, pocket_radius=1e9)
    m = _iabnet(n_iter=1)
    cfg = dict(orc.DEFAULT_CFG)
    cfg.update(mean_layers=4, n_iter=1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    ref_out = orc.model_forward(sd, cfg, data0.clone(), stage=1)
    ref_loss, ref_terms = orc.compute_loss(ref_out, data0)
    ref_loss.backward()
    ref_grads = {k: v.grad for k, v in sd.items() if v.is_floating_point()}
    ref_out = [o.detach() if torch.is_tensor(o) else o for o in ref_out]
    assert ref_out[9].numel() == 1500 * 40                              # the distance-map head saw every (residue, atom) pair: the whole graph
    engine.set_precision("fp32")
    m = m.to(dev)
    data = data0.clone().to(dev)
    out = m(data, stage=1, train=False)
    init = data0["compound"].node_coords.numpy()
    moved = rmsd(ref_out[0].numpy(), init)
    gap = rmsd(out[0].detach().cpu().numpy(), ref_out[0].numpy())
    print("config 3 read literally (1500 / 40, whole-protein pocket, IaBNet stage 1, n_iter 1): ligand moved %.3f A; RMSD vs oracle %.3e A" % (moved, gap))
    assert moved > 1e-2 and gap < 1e-4
    for i, n in TUPLE:
        r, g = ref_out[i].numpy(), out[i].detach().cpu().numpy()
        assert r.shape == g.shape, n
        assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), n
    assert np.array_equal(out[5].cpu().numpy(), ref_out[5].numpy()) and np.array_equal(out[6].cpu().numpy(), ref_out[6].numpy())
    loss, terms = compute_loss(out, data)
    lerr = abs(float(loss.detach()) - float(ref_loss)) / abs(float(ref_loss))
    print("    six-term loss %.6f vs oracle %.6f (rel %.2e)" % (float(loss.detach()), float(ref_loss), lerr))
    assert lerr <= 1e-5
    for k, v in terms.items():
        assert abs(float(v.detach()) - float(ref_terms[k])) <= 1e-5 * max(abs(float(ref_terms[k])), 1e-2), k
    loss.backward()
    rows = _grad_rows(m.named_parameters(), ref_grads)
    _print_rows("config 3 whole graph, fp32 gradients vs oracle autograd", rows)
    assert len(rows) >= 350 and _grads_ok(rows), (len(rows), rows[0])
    # the sub-object's dtype: forward gap and the loss; and its gradients as a whole against the oracle's (bf16-grade)
    engine.set_precision("bf16")
    try:
        for p_ in m.parameters():
            p_.grad = None
        data = data0.clone().to(dev)
        o16 = m(data, stage=1, train=False)
        l16, _ = compute_loss(o16, data)
        l16.backward()
    finally:
        engine.set_precision("fp32")
    g16 = rmsd(o16[0].detach().cpu().numpy(), ref_out[0].numpy())
    le16 = abs(float(l16.detach()) - float(ref_loss)) / abs(float(ref_loss))
    num = sum(float(((p_.grad.detach().cpu() - ref_grads[n_]) ** 2).sum()) for n_, p_ in m.named_parameters() if p_.grad is not None and ref_grads.get(n_) is not None)
    den = sum(float((ref_grads[n_] ** 2).sum()) for n_, p_ in m.named_parameters() if p_.grad is not None and ref_grads.get(n_) is not None)
    print("    bf16: ligand RMSD gap %.3e A (gate 1e-4: %s), loss rel gap %.3e, whole-gradient l2 error %.3e"
          % (g16, "met" if g16 < 1e-4 else "missed", le16, (num / den) ** 0.5))
    assert g16 < CONFIG3_BF16_BOUND[0] and le16 < CONFIG3_BF16_BOUND[1]
    assert (num / den) ** 0.5 < 5e-2


def test_iabnet_production_size_inference():
    from fabind_amd import engine
    _threads()
    dev = torch.device("cuda:0")
    engine.set_precision("fp32")
    m = _iabnet()
    ref = _iabnet_oracle(1)
    cfg = dict(orc.DEFAULT_CFG)
    cfg.update(mean_layers=4, n_iter=8)
    with torch.no_grad():
        cr, _ = orc.model_inference({k: v.detach() for k, v in m.state_dict().items()}, cfg, ref["data"].clone())
        c, _ = m.to(dev).inference(ref["data"].clone().to(dev))
    gap = rmsd(c.cpu().numpy(), cr.numpy())
    print("IaBNet production size, inference: RMSD vs oracle %.3e A" % gap)
    assert gap < 1e-4


# ------------------------------------------------------------------------------------------------------------------------------
# (ii) FABind+
# ------------------------------------------------------------------------------------------------------------------------------
PH, PL = 512, 5


def _plus_stack(n_iter):
    from fabind_amd import synthetic
    from fabind_amd.plus.models.att_model import EfficientMCAttModel
    torch.manual_seed(0)
    m = EfficientMCAttModel(_plus_args(PH, PL, n_iter), PH, PH, 1, n_layers=PL, n_iter=n_iter, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).eval()
    return synthetic.condition_plus_for_parity(m)


# measured at the first run of this file: bf16 ligand RMSD gap (A), asserted at 2x
PLUS_STACK_BF16_BOUND = {1: 1.8e-2, 2: 3.2e-2}                      # measured 8.7e-3 / 1.6e-2 A


@pytest.mark.parametrize("n_iter", [1, 2])
def test_plus_stack_production_size_matches_oracle(n_iter):
    """FABind+ stack at 1500 protein / 40 ligand nodes, hidden 512, 5 layers: X, H and the returned pair embedding."""
    from fabind_amd import engine, synthetic
    _threads()
    dev = torch.device("cuda:0")
    inp = synthetic.make_stack_batch([(1500, 40)], PH, seed=7)
    m = _plus_stack(n_iter)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        Xr, Hr, Zr = porc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"],
                                        inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"],
                                        PL, n_iter)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    m = m.to(dev)
    lig = inp["mask"].numpy()
    moved = rmsd(Xr.numpy()[lig] * 5, inp["X"].numpy()[lig] * 5)
    res = {}
    for prec in ("fp32", "bf16", "bf16x3"):
        engine.set_precision(prec)
        try:
            with torch.no_grad():
                X, Hh, Z = m(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"],
                             t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
        finally:
            engine.set_precision("fp32")
        res[prec] = (rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5),
                     float((Hh.cpu() - Hr).abs().max()) / max(1.0, float(Hr.abs().max())),
                     float((Z.cpu() - Zr).abs().max()) / max(1.0, float(Zr.abs().max())))
        assert Z.shape == Zr.shape
    print("FABind+ stack 1500/40, hidden 512, 5 layers, n_iter=%d: ligand moved %.3f A; fp32 RMSD vs oracle %.3e A, H %.3e, Z %.3e; "
          "bf16 gap %.3e A, H %.3e, Z %.3e; bf16x3 gap %.3e A, H %.3e, Z %.3e" % ((n_iter, moved) + res["fp32"] + res["bf16"] + res["bf16x3"]))
    assert moved > 1e-2
    assert res["bf16x3"][0] < 1e-4 and res["bf16x3"][1] <= 1e-4 and res["bf16x3"][2] <= 1e-4     # the gate-meeting fast mode: the gate itself
    assert res["fp32"][0] < 1e-4 and res["fp32"][1] <= 1e-4 and res["fp32"][2] <= 1e-4
    assert res["bf16"][0] < PLUS_STACK_BF16_BOUND[n_iter] and res["bf16"][1] < 5e-2 and res["bf16"][2] < 5e-2


def _plus_model(n_iter=8):
    from fabind_amd import synthetic
    from fabind_amd.plus.models import get_model
    a = _plus_args(PH, PL, n_iter)
    for k, v in dict(pocket_pred_hidden_size=128, pocket_pred_layers=1, pocket_pred_n_iter=1, random_n_iter=False,
                     use_for_radius_pred="ligand", dis_map_thres=15.0, pocket_radius_buffer=5.0, min_pocket_radius=20.0,
                     force_fix_radius=False, use_clustering=False, gs_tau=1.0, gs_hard=False, pocket_radius=20.0,
                     train_pred_pocket_noise=0.0, local_eval=False).items():
        setattr(a, k, v)
    torch.manual_seed(0)
    m = get_model(a, _Logger()).eval()
    return synthetic.condition_plus_for_parity(m)


_PLUS = {}
_RADIUS = [6.0, 7.5]


def _isos():
    return [[list(range(n)), list(reversed(range(n)))] for _, n in SIZES]


def _plus_oracle(stage):
    if stage not in _PLUS:
        from fabind_amd import synthetic
        _threads()
        m = _plus_model()
        data = synthetic.make_hetero_batch(SIZES, seed=11)
        cfg = dict(porc.PLUS_CFG)
        cfg.update(mean_layers=PL, n_iter=8)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
        out, coords_shifted = porc.model_forward(sd, cfg, data.clone(), stage=stage)
        loss, terms = porc.compute_loss(out, coords_shifted, data, [n for _, n in SIZES], _isos(), torch.tensor(_RADIUS))
        loss.backward()
        _PLUS[stage] = dict(out=[o.detach() if torch.is_tensor(o) else o for o in out], coords=coords_shifted.detach(),
                            loss=float(loss.detach()), terms={k: float(v.detach()) for k, v in terms.items()},
                            grads={k: v.grad for k, v in sd.items() if v.is_floating_point()}, data=data)
    return _PLUS[stage]


PLUS_NAMES = ["coords", "compound_batch", "y_pred", "y_pred_by_coords", "pocket_cls_pred", "pocket_cls", "protein_out_mask_whole",
              "protein_coords_batched_whole", "pred_pocket_center", "dis_map", "keepNode_less_5", "pocket_radius_pred",
              "pocket_center_bias"]
PLUS_BF16_BOUND = {1: (3.7e-2, 1e-3), 2: (4e-2, 1e-3)}      # measured 1.8e-2 / 2.0e-2 A, loss 4.6e-4 / 1.4e-4


@pytest.mark.parametrize("stage", [1, 2])
def test_plus_model_production_size_matches_oracle(stage):
    """FABindPlus.forward at hidden 512 / pocket 128, 5 layers, n_iter 8 on 1500- / 1100-residue proteins: 13-tuple, shifted
    data.coords, the seven-term loss and every parameter gradient."""
    from fabind_amd import engine
    from fabind_amd.plus.models import compute_loss
    dev = torch.device("cuda:0")
    ref = _plus_oracle(stage)
    engine.set_precision("fp32")
    m = _plus_model().to(dev)

    def batch():
        d = ref["data"].clone().to(dev)
        d.ligand_radius, d.num_atoms, d.isomorphisms = torch.tensor(_RADIUS, device=dev), [n for _, n in SIZES], _isos()
        return d
    data = batch()
    out = m(data, stage=stage, train=False)
    assert len(out) == 13
    gap = rmsd(out[0].detach().cpu().numpy(), ref["out"][0].numpy())
    init = ref["data"]["compound"].node_coords.numpy()
    print("FABind+ 512/128, 5 layers, n_iter 8, stage %d: RMSD vs oracle %.3e A (ligand rms distance from its input pose %.2f A)"
          % (stage, gap, rmsd(ref["out"][0].numpy() + ref["out"][12].numpy()[ref["out"][1].numpy()], init)))
    assert gap < 1e-4
    for i, n in enumerate(PLUS_NAMES):
        if n in ("coords", "keepNode_less_5"):
            continue
        r, g = ref["out"][i].numpy(), out[i].detach().cpu().numpy()
        assert g.shape == r.shape, n
        if r.dtype.kind in "biu":
            assert np.array_equal(g, r), n
        else:
            assert np.abs(g - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), n
    assert int(out[10]) == int(ref["out"][10])
    assert float((data.coords.cpu() - ref["coords"]).abs().max()) < 1e-4          # shifted in place like the reference
    loss, terms = compute_loss(out, data)
    lerr = abs(float(loss.detach()) - ref["loss"]) / abs(ref["loss"])
    print("    seven-term loss %.6f vs oracle %.6f (rel %.2e)" % (float(loss.detach()), ref["loss"], lerr))
    assert lerr <= 1e-5
    for k, v in terms.items():
        assert abs(float(v.detach()) - ref["terms"][k]) <= 1e-5 * max(abs(ref["terms"][k]), 1e-2), k
    loss.backward()
    rows = _grad_rows(m.named_parameters(), ref["grads"])
    _print_rows("FABind+ production size, stage %d, fp32 gradients vs oracle autograd" % stage, rows)
    assert len(rows) >= 480 and _grads_ok(rows), (len(rows), rows[0])

    engine.set_precision("bf16")
    try:
        data = batch()
        with torch.no_grad():
            o16 = m(data, stage=stage, train=False)
            l16, _ = compute_loss(o16, data)
    finally:
        engine.set_precision("fp32")
    g16 = rmsd(o16[0].cpu().numpy(), ref["out"][0].numpy())
    le16 = abs(float(l16) - ref["loss"]) / abs(ref["loss"])
    print("    bf16: ligand RMSD gap %.3e A, loss rel gap %.3e" % (g16, le16))
    assert g16 < PLUS_BF16_BOUND[stage][0] and le16 < PLUS_BF16_BOUND[stage][1]

    engine.set_precision("bf16x3")                  # the gate-meeting fast mode: 1e-4 A / 1e-5 relative loss
    try:
        data = batch()
        with torch.no_grad():
            o3 = m(data, stage=stage, train=False)
            l3, _ = compute_loss(o3, data)
    finally:
        engine.set_precision("fp32")
    g3 = rmsd(o3[0].cpu().numpy(), ref["out"][0].numpy())
    le3 = abs(float(l3) - ref["loss"]) / abs(ref["loss"])
    print("    bf16x3: ligand RMSD gap %.3e A, loss rel gap %.3e (gates 1e-4 A / 1e-5)" % (g3, le3))
    assert g3 < 1e-4 and le3 <= 1e-5
