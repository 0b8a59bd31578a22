"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by EXECUTING THE REFERENCE in this container.

Run:  python oracle/make_golden.py           (needs /root/reference; it is absent on the GPU box)

The reference's own python (FABind/fabind/models/*.py) is imported from where it lies under
/root/reference with the stand-ins of oracle/refshim.py; seeded random weights (reference init, then
the last coord-MLP layers scaled up so that atoms really move -- SURVEY.md section 7 "discriminating
goldens") and seeded synthetic inputs (fabind_amd/synthetic.py) go in, and inputs + weights + outputs
come out as small .npz fixtures.  Only data is written; no reference source is copied.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from fabind_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
COORD_GAIN = 300.0  # multiplies every *.coord_mlp.2.weight (reference init gain is 1e-3)


class _Logger:
    def log_message(self, s):
        pass


def _np(t):
    return t.detach().cpu().numpy()


def _boost(module):
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("coord_mlp.2.weight"):
                p.mul_(COORD_GAIN)
            # reference input/shrink/enlarge/distmap linears are xavier(gain=1e-3): lift them so the
            # heads carry signal through to the outputs being compared
            if n in ("protein_linear_whole_protein.weight", "compound_linear_whole_protein.weight",
                     "embedding_shrink.weight", "embedding_enlarge.weight", "distmap_mlp.0.weight",
                     "distmap_mlp.2.weight", "protein_to_pocket.linear1.weight",
                     "protein_to_pocket.linear2.weight"):
                p.mul_(300.0)


def _hetero_to_ref(d):
    """Copy a fabind_amd.synthetic.HeteroBatch into the shim's FakeHeteroData (same fields)."""
    out = refshim.FakeHeteroData()
    for key, st in d._stores.items():
        for k, v in st.items():
            out[key][k] = v.clone() if torch.is_tensor(v) else v
    for k, v in d._glob.items():
        setattr(out, k, v.clone() if torch.is_tensor(v) else v)
    return out


def golden_stack(mods, name, sizes, hidden, layers, n_iter, seed):
    torch.manual_seed(seed)
    args = refshim.production_args(hidden_size=hidden, mean_layers=layers, n_iter=n_iter)
    norm = lambda x: x / 5.0
    unnorm = lambda x: x * 5.0
    model = mods["models.att_model"].EfficientMCAttModel(
        args, hidden, hidden, 1, n_edge_feats=0, n_layers=layers, n_iter=n_iter, inter_cutoff=10.0,
        intra_cutoff=8.0, normalize_coord=norm, unnormalize_coord=unnorm).eval()
    _boost(model)
    batch = synthetic.make_stack_batch(sizes, hidden, seed=seed)
    cap = {}

    def hook(label):
        def f(mod, inp, out):
            cap[label + ".h"] = _np(out[0])
            cap[label + ".x"] = _np(out[1].reshape(-1, 3))
            if len(out) > 2:
                cap[label + ".alpha"] = _np(out[2])
        return f

    def edge_hook(mod, inp, out):
        if "ctx_edges_noBond" not in cap:
            cap["ctx_edges_noBond"], cap["inter_edges"] = _np(out[0]), _np(out[1])

    hs = [model.extract_edges.register_forward_hook(edge_hook)]
    res = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        m = model.to(dt)
        if tag == "f32":
            for i in range(layers):
                hs.append(getattr(m.gnn, "gcl_%d" % i).register_forward_hook(hook("gcl_%d" % i)))
                hs.append(getattr(m.gnn, "att_%d" % i).register_forward_hook(hook("att_%d" % i)))
        with torch.no_grad():
            X, H = m(batch["X"].clone().to(dt), batch["H"].clone().to(dt), batch["batch_id"], batch["segment_id"],
                     batch["mask"], batch["is_global"], batch["compound_edge_index"], batch["LAS_edge_index"],
                     batch["coord_LAS"].clone().to(dt))
        res["out_X_" + tag], res["out_H_" + tag] = _np(X), _np(H)
        if tag == "f32":
            for h in hs:
                h.remove()
    model.to(torch.float32)
    moved = np.sqrt(((res["out_X_f32"] - _np(batch["X"])) ** 2).sum(-1)).max() * 5
    noise = np.sqrt(((res["out_X_f32"] - res["out_X_f64"]) ** 2).sum(-1)).max() * 5
    print("%s: max displacement %.3f A, f32-vs-f64 %.2e A, E_ctx %d E_inter %d" % (
        name, moved, noise, cap["ctx_edges_noBond"].shape[1], cap["inter_edges"].shape[1]))
    save = {("in_" + k): _np(v) for k, v in batch.items() if torch.is_tensor(v)}
    save["sizes"] = np.array(sizes)
    save["cfg"] = np.array([hidden, layers, n_iter, seed])
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    save.update(res)
    save.update({"cap_" + k: v for k, v in cap.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def _loss_and_gradients(model, data, save, name):
    """Train-step loss (6 terms) + per-parameter gradient norm and 16 samples for both stages, into `save`."""
    # train-step loss + gradients (eval-mode modules: dropout off, no Gumbel noise; stage fixed)
    for stage in (1, 2):
        model.zero_grad()
        d = _hetero_to_ref(data)
        out = model(d, stage=stage, train=False)
        F = torch.nn.functional
        coords, c_batch, y_pred, y_by, logits, pocket_cls, p_mask, _, center, dis_map, _ = out
        terms = dict(
            pocket_cls=F.binary_cross_entropy_with_logits(logits, pocket_cls.float()) * (p_mask.numel() / p_mask.sum()),
            pocket_center=0.05 * torch.nn.HuberLoss(delta=3.0)(center, d.coords_center),
            contact=F.mse_loss(y_pred, dis_map), contact_by_pred=F.mse_loss(y_by, dis_map),
            distill=F.mse_loss(y_by, y_pred), coord=F.smooth_l1_loss(coords, d.coords))
        loss = sum(terms.values())
        loss.backward()
        save["s%d_loss" % stage] = _np(loss)
        for k, v in terms.items():
            save["s%d_loss_%s" % (stage, k)] = _np(v)
        for n, p in model.named_parameters():
            if p.grad is not None:  # norm + 16 evenly spaced entries per tensor keeps the fixture small
                gflat = p.grad.flatten()
                idx = torch.linspace(0, gflat.numel() - 1, 16).long()
                save["s%d_gradnorm_%s" % (stage, n)] = _np(gflat.norm())
                save["s%d_gradsmp_%s" % (stage, n)] = _np(gflat[idx])
        print("%s stage %d: loss %.6f" % (name, stage, float(loss)), {k: round(float(v), 5) for k, v in terms.items()})


def golden_model(mods, name, sizes, hidden, pocket_hidden, layers, n_iter, seed):
    torch.manual_seed(seed)
    args = refshim.production_args(hidden_size=hidden, pocket_pred_hidden_size=pocket_hidden, mean_layers=layers,
                                   n_iter=n_iter, random_n_iter=False)
    model = mods["models.model"].get_model(args, _Logger(), None).eval()
    _boost(model)
    data = synthetic.make_hetero_batch(sizes, seed=seed)
    save = {"sizes": np.array(sizes), "cfg": np.array([hidden, pocket_hidden, layers, n_iter, seed])}
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    names = ["coords", "compound_batch", "y_pred", "y_pred_by_coords", "pocket_cls_pred", "pocket_cls",
             "protein_out_mask_whole", "protein_coords_batched_whole", "pred_pocket_center", "dis_map"]
    for stage in (1, 2):
        with torch.no_grad():
            out = model(_hetero_to_ref(data), stage=stage, train=False)
        for n, o in zip(names, out[:10]):
            save["s%d_%s" % (stage, n)] = _np(o)
        save["s%d_keepNode_less_5" % stage] = np.array(out[10])
        moved = (out[0] - data["compound"].node_coords).norm(dim=-1).max()
        print("%s stage %d: ligand moved up to %.3f A, center %s" % (name, stage, moved, _np(out[8][0])))
    with torch.no_grad():
        out = model.inference(_hetero_to_ref(data))
    save["inf_coords"] = _np(out[0])
    _loss_and_gradients(model, data, save, name)
    hd = {}
    for key, st in data._stores.items():
        kname = key if isinstance(key, str) else "|".join(key)
        for k, v in st.items():
            if torch.is_tensor(v):
                hd["d_%s::%s" % (kname, k)] = _np(v)
    for k, v in data._glob.items():
        if torch.is_tensor(v):
            hd["d_::%s" % k] = _np(v)
    save.update(hd)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_stack_grad(mods, name, sizes, hidden, layers, seed):
    """Gradients of a scalar of the stack outputs w.r.t. every parameter and the input H (n_iter=1)."""
    torch.manual_seed(seed)
    args = refshim.production_args(hidden_size=hidden, mean_layers=layers, n_iter=1)
    model = mods["models.att_model"].EfficientMCAttModel(
        args, hidden, hidden, 1, n_edge_feats=0, n_layers=layers, n_iter=1, inter_cutoff=10.0, intra_cutoff=8.0,
        normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0).eval()
    _boost(model)
    batch = synthetic.make_stack_batch(sizes, hidden, seed=seed)
    Hin = batch["H"].clone().requires_grad_(True)
    g = torch.Generator().manual_seed(seed + 77)
    N = Hin.shape[0]
    cx, ch = torch.randn(N, 1, 3, generator=g), torch.randn(N, hidden, generator=g)
    X, H = model(batch["X"].clone(), Hin, batch["batch_id"], batch["segment_id"], batch["mask"], batch["is_global"],
                 batch["compound_edge_index"], batch["LAS_edge_index"], batch["coord_LAS"].clone())
    loss = (X * cx).sum() + (H * ch).sum()
    loss.backward()
    save = {("in_" + k): _np(v) for k, v in batch.items() if torch.is_tensor(v)}
    save["sizes"], save["cfg"] = np.array(sizes), np.array([hidden, layers, 1, seed])
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    save["cot_X"], save["cot_H"], save["loss"] = _np(cx), _np(ch), _np(loss)
    save["out_X_f32"], save["out_H_f32"] = _np(X), _np(H)
    save["grad_in_H"] = _np(Hin.grad)
    nograd = []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
        else:
            save["grad_" + n] = _np(p.grad)
    save["nograd"] = np.array(nograd)
    print("%s: loss %.5f, %d params without grad" % (name, float(loss), len(nograd)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_stack_plus(mods, name, sizes, hidden, layers, n_iter, seed):
    """FABind+ stack (FABind_plus/fabind/models/att_model.py:131-223): (X, H, pair_embed) + per-layer captures."""
    torch.manual_seed(seed)
    args = refshim.production_args_plus(hidden_size=hidden, mean_layers=layers, n_iter=n_iter)
    model = mods["models.att_model"].EfficientMCAttModel(
        args, hidden, hidden, 1, n_edge_feats=0, n_layers=layers, n_iter=n_iter, inter_cutoff=10.0, intra_cutoff=8.0,
        normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("coord_mlp.linear2.weight"):          # xavier gain 1e-3 in the reference: lift it so atoms move
                p.mul_(COORD_GAIN)
    batch = synthetic.make_stack_batch(sizes, hidden, seed=seed)
    cap, hs = {}, []

    def hook(label):
        def f(mod, inp, out):
            cap[label + ".h"] = _np(out[0])
            cap[label + ".x"] = _np(out[1].reshape(-1, 3))
            if len(out) > 2:
                cap[label + ".alpha"] = _np(out[2])
        return f

    res = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        m = model.to(dt)
        if tag == "f32":
            for i in range(layers):
                hs.append(getattr(m.gnn, "gcl_%d" % i).register_forward_hook(hook("gcl_%d" % i)))
                hs.append(getattr(m.gnn, "att_%d" % i).register_forward_hook(hook("att_%d" % i)))
        with torch.no_grad():
            X, H, Z = m(batch["X"].clone().to(dt), batch["H"].clone().to(dt), batch["batch_id"], batch["segment_id"],
                        batch["mask"], batch["is_global"], batch["compound_edge_index"], batch["LAS_edge_index"],
                        batch["coord_LAS"].clone().to(dt))
        res["out_X_" + tag], res["out_H_" + tag], res["out_Z_" + tag] = _np(X), _np(H), _np(Z)
        for h in hs:
            h.remove()
        hs = []
    model.to(torch.float32)
    moved = np.sqrt(((res["out_X_f32"] - _np(batch["X"])) ** 2).sum(-1)).max() * 5
    noise = np.sqrt(((res["out_X_f32"] - res["out_X_f64"]) ** 2).sum(-1)).max() * 5
    print("%s: max displacement %.3f A, f32-vs-f64 %.2e A, |Z| max %.3f" % (name, moved, noise, np.abs(res["out_Z_f32"]).max()))
    save = {("in_" + k): _np(v) for k, v in batch.items() if torch.is_tensor(v)}
    save["sizes"] = np.array(sizes)
    save["cfg"] = np.array([hidden, layers, n_iter, seed])
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    save.update(res)
    save.update({"cap_" + k: v for k, v in cap.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_model_plus(mods, name, sizes, hidden, pocket_hidden, layers, n_iter, seed):
    """FABindPlus.forward (stage 1 and 2, eval) and .inference; 13-tuples + the in-place shifted data.coords."""
    torch.manual_seed(seed)
    # min_pocket_radius 10 A (production: 20) so that the radius crop of these small synthetic proteins is partial and the
    # pocket-centred frame / pocket_center_bias are non-trivial
    args = refshim.production_args_plus(hidden_size=hidden, pocket_pred_hidden_size=pocket_hidden, mean_layers=layers,
                                        n_iter=n_iter, random_n_iter=False, min_pocket_radius=10.0)
    model = mods["models.model"].get_model(args, _Logger()).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("coord_mlp.linear2.weight"):
                p.mul_(COORD_GAIN)
            if n in ("protein_linear_whole_protein.weight", "compound_linear_whole_protein.weight", "embedding_shrink.weight",
                     "embedding_enlarge.weight"):
                p.mul_(300.0)                   # xavier gain 1e-3 in the reference: lift so the heads carry signal
    data = synthetic.make_hetero_batch(sizes, seed=seed)
    save = {"sizes": np.array(sizes), "cfg": np.array([hidden, pocket_hidden, layers, n_iter, seed]),
            "min_pocket_radius": np.array(10.0)}
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    names = ["coords", "compound_batch", "y_pred", "y_pred_by_coords", "pocket_cls_pred", "pocket_cls",
             "protein_out_mask_whole", "protein_coords_batched_whole", "pred_pocket_center", "dis_map", "keepNode_less_5",
             "pocket_radius_pred", "pocket_center_bias"]
    for stage in (1, 2):
        d = _hetero_to_ref(data)
        with torch.no_grad():
            out = model(d, stage=stage, train=False)
        for n, o in zip(names, out):
            save["s%d_%s" % (stage, n)] = _np(o) if torch.is_tensor(o) else np.array(o)
        save["s%d_data_coords" % stage] = _np(d.coords)
        print("%s stage %d: radius %s, bias %s, |y_pred| %.3f" % (name, stage, _np(out[11]).ravel(), _np(out[12][0]),
                                                               float(out[2].abs().max())))
    d = _hetero_to_ref(data)
    with torch.no_grad():
        out = model.inference(d)
    save["inf_coords"] = _np(out[0])
    hd = {}
    for key, st in data._stores.items():
        kname = key if isinstance(key, str) else "|".join(key)
        for k, v in st.items():
            if torch.is_tensor(v):
                hd["d_%s::%s" % (kname, k)] = _np(v)
    for k, v in data._glob.items():
        if torch.is_tensor(v):
            hd["d_::%s" % k] = _np(v)
    save.update(hd)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_model_plus_sampling(mods, name, sizes, hidden, pocket_hidden, layers, n_iter, seed):
    """FABind+ sampling-mode pieces in eval (deterministic) form: confidence / ranking head (stack_mlp) and the DBSCAN
    centre choice (python `random` seeded) -> inference 3-tuple."""
    import random
    torch.manual_seed(seed)
    args = refshim.production_args_plus(hidden_size=hidden, pocket_pred_hidden_size=pocket_hidden, mean_layers=layers,
                                        n_iter=n_iter, random_n_iter=False, min_pocket_radius=10.0, confidence_training=True,
                                        stack_mlp=True, confidence_use_ln_mlp=True, confidence_dropout=0.2,
                                        confidence_mlp_hidden_scale=1, use_clustering=True, dbscan_eps=9.0,
                                        dbscan_min_samples=2, choose_cluster_prob=0.5)
    model = mods["models.model"].get_model(args, _Logger()).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("coord_mlp.linear2.weight"):
                p.mul_(COORD_GAIN)
            if n in ("protein_linear_whole_protein.weight", "compound_linear_whole_protein.weight", "embedding_shrink.weight",
                     "embedding_enlarge.weight"):
                p.mul_(300.0)
    data = synthetic.make_hetero_batch(sizes, seed=seed)
    save = {"sizes": np.array(sizes), "cfg": np.array([hidden, pocket_hidden, layers, n_iter, seed]),
            "min_pocket_radius": np.array(10.0), "py_seed": np.array(4321)}
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    random.seed(4321)
    with torch.no_grad():
        out = model.inference(_hetero_to_ref(data))
    save["inf_coords"], save["inf_batch"], save["inf_conf"] = _np(out[0]), _np(out[1]), _np(out[2])
    print("%s: confidence %s" % (name, _np(out[2])))
    # confidence-mode forward (train_confidence.py / utils/training_confidence.py:39,189): the 7-tuple, for the predicted
    # pocket (stage 2) and teacher-forced (train=True), and the gradient of a weighted score sum w.r.t. the ranking head
    names7 = ["coords", "compound_batch", "pocket_cls_pred", "protein_out_mask_whole", "keepNode_less_5", "confidence",
              "pocket_center_bias"]
    for k, (tag, kw) in enumerate((("cf2", dict(stage=2, train=False)), ("cf1", dict(stage=1, train=True))), 1):
        d = _hetero_to_ref(data)
        model.zero_grad()
        random.seed(4321 + k)                       # forward draws the DBSCAN cluster choice too (model.py:147-167)
        out = model(d, **kw)
        for n, o in zip(names7, out):
            save["%s_%s" % (tag, n)] = _np(o) if torch.is_tensor(o) else np.array(o)
        wts = torch.arange(1, out[5].numel() + 1, dtype=out[5].dtype)
        (out[5].reshape(-1) * wts).sum().backward()
        for n, p in model.named_parameters():
            if n.startswith("ranking") and p.grad is not None:
                save["%s_grad_%s" % (tag, n)] = _np(p.grad)
        print("%s %s: confidence %s" % (name, tag, _np(out[5]).ravel()))
    for key, st in data._stores.items():
        kname = key if isinstance(key, str) else "|".join(key)
        for k, v in st.items():
            if torch.is_tensor(v):
                save["d_%s::%s" % (kname, k)] = _np(v)
    for k, v in data._glob.items():
        if torch.is_tensor(v):
            save["d_::%s" % k] = _np(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_plus_loss(mods, name, seed):
    """The reference's own compute_permutation_loss (utils/permutation_loss.py) + the other train-step terms
    (utils/training.py:61-97) on the stage-2 outputs stored in plus_model_tiny, with synthetic automorphism lists."""
    import importlib
    sys.path.insert(0, os.path.join(refshim.REFERENCE_ROOT, "FABind_plus", "fabind"))
    try:
        perm = importlib.import_module("utils.permutation_loss")
    finally:
        sys.path.pop(0)
    g = dict(np.load(os.path.join(OUT, "plus_model_tiny.npz")))
    gen = torch.Generator().manual_seed(seed)
    t = lambda k: torch.from_numpy(g[k])
    coords, cb = t("s2_coords"), t("s2_compound_batch")
    B = int(cb.max()) + 1
    num_atoms = torch.bincount(cb, minlength=B).tolist()
    isos = []
    for n in num_atoms:                                   # identity + 3 random permutations (2 for odd complexes)
        cur = [list(range(n))]
        for _ in range(3):
            cur.append(torch.randperm(n, generator=gen).tolist())
        isos.append(cur)
    isos[-1] = [list(range(num_atoms[-1]))]               # a ligand without symmetry: single automorphism
    target = t("s2_data_coords") + 0.3 * torch.randn(coords.shape, generator=gen)
    # make a non-identity automorphism the best one for complex 0
    n0 = num_atoms[0]
    target[:n0] = coords[:n0][torch.tensor(isos[0][2])] + 0.05 * torch.randn(n0, 3, generator=gen)

    class _D:
        pass
    d = _D()
    d.num_atoms, d.isomorphisms = num_atoms, isos
    d.__len__ = lambda self=None: B
    _D.__len__ = lambda self: B
    crit = torch.nn.SmoothL1Loss(reduction="none")
    new_idx = perm.update_best_isomorphism_idx(coords, target, d, crit)
    coord_loss = 1.5 * perm.compute_permutation_loss(coords, target, d, crit).mean()
    F = torch.nn.functional
    logits, pocket_cls, p_mask = t("s2_pocket_cls_pred"), t("s2_pocket_cls"), t("s2_protein_out_mask_whole")
    ligand_radius = torch.tensor([6.0, 4.5])
    center_true = torch.from_numpy(g["d_::coords_center"])
    terms = dict(
        coord=coord_loss,
        pocket_cls=F.binary_cross_entropy_with_logits(logits, pocket_cls.float()) * (p_mask.numel() / p_mask.sum()),
        pocket_center=0.05 * torch.nn.HuberLoss(delta=3.0)(t("s2_pred_pocket_center"), center_true),
        contact=F.mse_loss(t("s2_y_pred"), t("s2_dis_map")), contact_by_pred=F.mse_loss(t("s2_y_pred_by_coords"), t("s2_dis_map")),
        distill=F.mse_loss(t("s2_y_pred_by_coords"), t("s2_y_pred")),
        pocket_radius=0.05 * torch.nn.HuberLoss(delta=3.0)(t("s2_pocket_radius_pred").squeeze(1), ligand_radius))
    save = {"target": _np(target), "new_idx": _np(new_idx), "ligand_radius": _np(ligand_radius), "loss": _np(sum(terms.values())),
            "n_iso": np.array([len(i) for i in isos])}
    for b, cur in enumerate(isos):
        save["iso_%d" % b] = np.array(cur)
    for k, v in terms.items():
        save["loss_" + k] = _np(v)
    print(name, {k: round(float(v), 5) for k, v in terms.items()}, "best iso of complex 0 non-identity:",
          bool((new_idx[:n0] != torch.arange(n0)).any()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_stack_plus_grad(mods, name, sizes, hidden, layers, seed):
    """Gradients of a scalar of the FABind+ stack outputs (X, H, pair_embed) w.r.t. every parameter and the input H."""
    torch.manual_seed(seed)
    args = refshim.production_args_plus(hidden_size=hidden, mean_layers=layers, n_iter=1)
    model = mods["models.att_model"].EfficientMCAttModel(
        args, hidden, hidden, 1, n_edge_feats=0, n_layers=layers, n_iter=1, inter_cutoff=10.0, intra_cutoff=8.0,
        normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("coord_mlp.linear2.weight"):
                p.mul_(COORD_GAIN)
    batch = synthetic.make_stack_batch(sizes, hidden, seed=seed)
    Hin = batch["H"].clone().requires_grad_(True)
    g = torch.Generator().manual_seed(seed + 77)
    N = Hin.shape[0]
    cx, ch = torch.randn(N, 1, 3, generator=g), torch.randn(N, hidden, generator=g)
    X, H, Z = model(batch["X"].clone(), Hin, batch["batch_id"], batch["segment_id"], batch["mask"], batch["is_global"],
                    batch["compound_edge_index"], batch["LAS_edge_index"], batch["coord_LAS"].clone())
    cz = torch.randn(Z.shape, generator=g)
    loss = (X * cx).sum() + (H * ch).sum() + (Z * cz).sum()
    loss.backward()
    save = {("in_" + k): _np(v) for k, v in batch.items() if torch.is_tensor(v)}
    save["sizes"], save["cfg"] = np.array(sizes), np.array([hidden, layers, 1, seed])
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    save["cot_X"], save["cot_H"], save["cot_Z"], save["loss"] = _np(cx), _np(ch), _np(cz), _np(loss)
    save["out_X_f32"], save["out_H_f32"] = _np(X), _np(H)
    save["grad_in_H"] = _np(Hin.grad)
    nograd = []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
        else:
            save["grad_" + n] = _np(p.grad)
    save["nograd"] = np.array(nograd)
    print("%s: loss %.5f, %d params without grad" % (name, float(loss), len(nograd)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_post_optim(name, seed):
    """Ligand post-optimisation (FABind/fabind/utils/post_optim_utils.py:36-64) run by the reference itself: random-walk
    ligands (1.5 A bonds, LAS = index distance <= 2 as in fabind_amd.synthetic), prediction = conformer + 0.6 A noise;
    short horizon (5 epochs), the production horizon (1000 epochs) and the LAS-free variant."""
    ref_mod = refshim.load_post_optim()
    torch.set_num_threads(1)
    rng = np.random.default_rng(seed)
    sizes = [12, 20, 25, 40]
    save = {"sizes": np.array(sizes)}
    for li, n in enumerate(sizes):
        st = rng.normal(size=(n, 3))
        st /= np.linalg.norm(st, axis=1, keepdims=True)
        ref = np.cumsum(1.5 * st, 0).astype(np.float32)
        idx = np.arange(n)
        ii, jj = np.nonzero((np.abs(idx[:, None] - idx[None]) <= 2) & (idx[:, None] != idx[None]))
        las = np.stack([ii, jj]).astype(np.int64)
        pred = (ref + rng.normal(scale=0.6, size=ref.shape)).astype(np.float32)
        save["ref_%d" % li], save["pred_%d" % li], save["las_%d" % li] = ref, pred, las
        for ep in (5, 1000):
            x, loss, rmsd = ref_mod.post_optimize_compound_coords(torch.from_numpy(ref), torch.from_numpy(pred), total_epoch=ep,
                                                                  LAS_edge_index=torch.from_numpy(las))
            save["x_%d_e%d" % (li, ep)], save["loss_%d_e%d" % (li, ep)], save["rmsd_%d_e%d" % (li, ep)] = _np(x), np.array(loss), np.array(rmsd)
            print("%s ligand %d (%d atoms) epochs %d: loss %.4f rmsd %.4f" % (name, li, n, ep, loss, rmsd))
        x, loss, rmsd = ref_mod.post_optimize_compound_coords(torch.from_numpy(ref), torch.from_numpy(pred), total_epoch=5,
                                                              LAS_edge_index=None)
        save["x_%d_nolas" % li], save["loss_%d_nolas" % li], save["rmsd_%d_nolas" % li] = _np(x), np.array(loss), np.array(rmsd)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_eval_metrics(mods, name, seed):
    """The reference's own evaluation loop (utils/utils.py:445-604) over three small batches.  The "model" handed to it
    returns prepared 11-tuples (ligand poses at 0.3-6 A from the truth, random pocket logits incl. complexes with no
    positive residue = the reference's skip branch), so every branch and threshold of the metric code is exercised; the
    fixture holds those tuples, the data fields the loop reads and the metrics dict it returns."""
    from argparse import Namespace
    g = torch.Generator().manual_seed(seed)
    args = refshim.production_args()
    for k, v in dict(disable_tqdm=True, tqdm_interval=1.0, pair_distance_loss_weight=1.0, pair_distance_distill_loss_weight=1.0,
                     pocket_cls_loss_weight=1.0, pocket_distance_loss_weight=0.05, coord_loss_weight=1.0).items():
        setattr(args, k, v)
    nn = torch.nn
    crit = dict(com=nn.SmoothL1Loss(), pair=nn.MSELoss(), cls=nn.BCEWithLogitsLoss(reduction="mean"), cen=nn.HuberLoss(delta=3.0))
    ev = mods["utils.utils"].evaluate_mean_pocket_cls_coord_multi_task
    sizes = [[(60, 8), (75, 6), (50, 10), (40, 5)], [(66, 7), (58, 9), (45, 12)], [(52, 6), (70, 11), (64, 8), (48, 7)]]
    save = {"n_batches": np.array(len(sizes)), "gs_tau": np.array(args.gs_tau)}
    names = ["coords", "compound_batch", "y_pred", "y_pred_by_coords", "pocket_cls_pred", "pocket_cls",
             "protein_out_mask_whole", "protein_coords_batched_whole", "pred_pocket_center", "dis_map"]
    loader, outs = [], []
    for bi, sz in enumerate(sizes):
        b = synthetic.make_hetero_batch(sz, seed=seed + bi)
        d = _hetero_to_ref(b)
        B = len(sz)
        d.pdb = ["c%d_%d" % (bi, i) for i in range(B)]
        d.to = lambda device, _d=d: _d
        cb, pb = b["compound"].batch, b["protein_whole"].batch
        noise = torch.tensor([0.3, 1.5, 3.0, 6.0])[:B]
        coords = b.coords + torch.randn(b.coords.shape, generator=g) * noise[cb][:, None] + torch.randn(B, 3, generator=g)[cb] * noise[cb][:, None]
        L = torch.bincount(pb, minlength=B)
        Lmax = int(L.max())
        mask = torch.arange(Lmax)[None, :] < L[:, None]
        logits = torch.randn(B, Lmax, generator=g) * 2.0 - 1.0
        logits[B - 1] = -3.0 - torch.rand(Lmax, generator=g)                   # no positive residue: the skip branch
        logits = logits * mask
        pocket_cls = torch.zeros(B, Lmax, dtype=torch.long)
        pocket_cls[pb, torch.arange(pb.shape[0]) - torch.cumsum(L, 0)[pb] + L[pb]] = b.pocket_idx.long()
        xyz = torch.zeros(B, Lmax, 3)
        xyz[pb, torch.arange(pb.shape[0]) - torch.cumsum(L, 0)[pb] + L[pb]] = b.node_xyz_whole.float()
        dis_map = b.dis_map.float()
        y_pred = (dis_map + torch.randn(dis_map.shape, generator=g)).clamp(0, 10)
        y_by = (dis_map + 2.0 * torch.randn(dis_map.shape, generator=g)).clamp(0, 10)
        center = b.coords_center + torch.randn(B, 3, generator=g) * 2.0
        out = (coords, cb, y_pred, y_by, logits, pocket_cls, mask, xyz, center, dis_map, bi)
        for n, o in zip(names, out[:10]):
            save["b%d_%s" % (bi, n)] = _np(o)
        save["b%d_keepNode_less_5" % bi] = np.array(bi)
        save["b%d_data_coords" % bi], save["b%d_coords_center" % bi] = _np(b.coords), _np(b.coords_center)
        loader.append(d)
        outs.append(out)
    it = iter(outs)
    met = ev(Namespace(is_main_process=True), args, loader, lambda data, stage: next(it), crit["com"], crit["pair"], crit["cls"],
             crit["cen"], 0.01, "cpu", pred_dis=True, stage=1)
    for k, v in met.items():
        save["metric::%s" % k] = np.array(float(v))
    print(name, {k: round(float(v), 4) for k, v in met.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_data_builder(mods, name, seed):
    """The reference's per-sample HeteroData builder (utils/utils.py:202-442, production options) on random raw complexes:
    the raw inputs and every field it emits, per sample (the test collates them).  Sample 2 has its ligand far from the
    protein (< 5 residues in the pocket radius -> the first-100-residues fallback)."""
    from argparse import Namespace
    g = torch.Generator().manual_seed(seed)
    args = Namespace(train_pred_pocket_noise=0.0, local_eval=False, train_ligand_torsion_noise=False, data_path=None)
    build = mods["utils.utils"].construct_data_from_graph_gvp_mean
    save = {"n": np.array(4)}
    for i, (L, nc) in enumerate([(130, 9), (96, 14), (150, 6), (88, 11)]):
        R = (3 * 135.0 * L / (4 * math.pi)) ** (1 / 3)
        v = torch.randn(L, 3, generator=g)
        prot = v / v.norm(dim=1, keepdim=True) * R * torch.rand(L, 1, generator=g) ** (1 / 3) + torch.tensor([12.0, -7.0, 3.0])
        st = torch.randn(nc, 3, generator=g)
        walk = torch.cumsum(1.5 * st / st.norm(dim=1, keepdim=True), 0)
        centre = torch.tensor([12.0 + 0.4 * R, -7.0, 3.0]) if i != 2 else torch.tensor([12.0 + R + 45.0, -7.0, 3.0])
        coords = (walk - walk.mean(0) + centre).numpy().astype(np.float64)
        rdkit = (walk - walk.mean(0)) @ torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        idx = torch.arange(nc)
        bonds = torch.stack([torch.cat([idx[:-1], idx[1:]]), torch.cat([idx[1:], idx[:-1]])], 1)
        bonds = torch.cat([bonds, torch.zeros(bonds.shape[0], 1, dtype=torch.long)], 1)          # [Eb, 3]: (i, j, bond type)
        m = ((idx[:, None] - idx[None]).abs() <= 2) & (idx[:, None] != idx[None])
        las = torch.nonzero(m).t().contiguous()
        esm, cf = torch.randn(L, 16, generator=g), torch.randn(nc, 8, generator=g)
        raw = dict(protein_node_xyz=prot, protein_esm2_feat=esm, coords=coords, compound_node_features=cf,
                   input_atom_edge_list=bonds, LAS_edge_index=las, rdkit_coords=rdkit.numpy())
        for k, v_ in raw.items():
            save["raw%d_%s" % (i, k)] = _np(v_) if torch.is_tensor(v_) else np.asarray(v_)
        data, _, _ = build(args, prot.clone(), "A" * L, coords.copy(), cf, bonds, None, las, rdkit.numpy().copy(), pdb_id="x%d" % i,
                           group="test", protein_esm2_feat=esm)
        for key, stt in data._stores.items():
            kname = key if isinstance(key, str) else "|".join(key)
            for k, v_ in stt.items():
                if torch.is_tensor(v_):
                    save["out%d_%s::%s" % (i, kname, k)] = _np(v_)
        for k, v_ in data._glob.items():
            if torch.is_tensor(v_):
                save["out%d_::%s" % (i, k)] = _np(v_)
        print(name, i, "pocket residues", int(data["pocket"].keepNode.sum()), "of", L)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def _parse_6g3c():
    """C-alpha coordinates of inference_examples/pdb_files/6g3c.pdb and heavy atoms + heavy-atom bonds of
    gt_mol_files/6g3c/6g3c_ligand.sdf (V2000), with plain text parsing (no RDKit / Biopython here)."""
    root = os.path.join(refshim.REFERENCE_ROOT, "FABind", "inference_examples")
    ca = []
    for line in open(os.path.join(root, "pdb_files", "6g3c.pdb")):
        if line.startswith("ATOM") and line[12:16].strip() == "CA" and line[16] in (" ", "A"):
            ca.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    lines = open(os.path.join(root, "gt_mol_files", "6g3c", "6g3c_ligand.sdf")).read().split("\n")
    na, nb = int(lines[3][:3]), int(lines[3][3:6])
    xyz, elem = [], []
    for l in lines[4:4 + na]:
        xyz.append([float(l[0:10]), float(l[10:20]), float(l[20:30])])
        elem.append(l[31:34].strip())
    heavy = [i for i, e in enumerate(elem) if e != "H"]
    new = {a: k for k, a in enumerate(heavy)}
    bonds = []
    for l in lines[4 + na:4 + na + nb]:
        a, b = int(l[0:3]) - 1, int(l[3:6]) - 1
        if a in new and b in new:
            bonds += [[new[a], new[b]], [new[b], new[a]]]
    return np.array(ca, np.float32), np.array(xyz, np.float32)[heavy], np.array(bonds, np.int64)


def golden_model_6g3c(mods, name, seed):
    """BASELINE config 0 / SURVEY 8(c): the reference's own example complex 6g3c (real geometry: 272 C-alpha, the ligand's
    heavy atoms and bonds) through the reference's per-sample builder (utils/utils.py:202-442) and model (1 layer, n_iter 1,
    fp32; hidden 64 / 32 to keep the fixture small).  Stand-ins for what this image lacks: node features are random
    (ESM2 / torchdrug absent); the "RDKit conformer" is the crystal pose under a random rotation, re-centred, and the LAS
    edges are the atom pairs at bond-graph distance <= 2 (RDKit absent) -- geometry, bonds and every model operation are
    the reference's."""
    from argparse import Namespace
    g = torch.Generator().manual_seed(seed)
    ca, lig, bonds = _parse_6g3c()
    L, nc = ca.shape[0], lig.shape[0]
    adj = np.zeros((nc, nc), bool)
    adj[bonds[:, 0], bonds[:, 1]] = True
    two = (adj.astype(np.int64) @ adj.astype(np.int64)) > 0
    las = np.stack(np.nonzero((adj | two) & ~np.eye(nc, dtype=bool)))
    rot = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].numpy().astype(np.float32)
    rdk = (lig - lig.mean(0)) @ rot.T
    esm, cf = 0.1 * torch.randn(L, 1280, generator=g), 0.1 * torch.randn(nc, 56, generator=g)
    edge3 = torch.from_numpy(np.concatenate([bonds, np.zeros((bonds.shape[0], 1), np.int64)], 1))
    bargs = Namespace(train_pred_pocket_noise=0.0, local_eval=False, train_ligand_torsion_noise=False, data_path=None)
    ref_d, _, _ = mods["utils.utils"].construct_data_from_graph_gvp_mean(
        bargs, torch.from_numpy(ca), "A" * L, lig.astype(np.float64), cf, edge3, None, torch.from_numpy(las), rdk.copy(),
        pdb_id="6g3c", group="test", protein_esm2_feat=esm)
    # PyG collation of a batch of one: batch vectors of zeros per node / edge-list store
    data = synthetic.HeteroBatch()
    for key, st in ref_d._stores.items():
        for k, v in st.items():
            data[key][k] = v
    for k, v in ref_d._glob.items():
        if torch.is_tensor(v):
            setattr(data, k, v)
    for st, f in (("compound", "node_feats"), ("protein_whole", "node_feats"), ("pocket", "node_feats"),
                  ("complex", "node_coords"), ("complex_whole_protein", "node_coords"), ("compound_atom_edge_list", "x"),
                  ("LAS_edge_list", "x")):
        data[st].batch = torch.zeros(data[st][f].shape[0], dtype=torch.long)
    data["compound"].node_coords = data["compound"].node_coords.float()
    data["compound"].rdkit_coords = torch.as_tensor(data["compound"].rdkit_coords).float()
    data.pocket_residue_center = data.node_xyz.mean(0, keepdim=True)
    print("%s: %d residues (%d in the 20 A pocket), %d heavy atoms, %d directed bonds, %d LAS edges" % (
        name, L, int(data["pocket"].keepNode.sum()), nc, bonds.shape[0], las.shape[1]))
    torch.manual_seed(seed)
    args = refshim.production_args(hidden_size=64, pocket_pred_hidden_size=32, mean_layers=1, n_iter=1, random_n_iter=False)
    model = mods["models.model"].get_model(args, _Logger(), None).eval()
    _boost(model)
    save = {"sizes": np.array([[L, nc]]), "cfg": np.array([64, 32, 1, 1, seed])}
    save.update({"w_" + k: _np(v) for k, v in model.state_dict().items()})
    names = ["coords", "compound_batch", "y_pred", "y_pred_by_coords", "pocket_cls_pred", "pocket_cls",
             "protein_out_mask_whole", "protein_coords_batched_whole", "pred_pocket_center", "dis_map"]
    for stage in (1, 2):
        with torch.no_grad():
            out = model(_hetero_to_ref(data), stage=stage, train=False)
        for n, o in zip(names, out[:10]):
            save["s%d_%s" % (stage, n)] = _np(o)
        save["s%d_keepNode_less_5" % stage] = np.array(out[10])
        print("%s stage %d: ligand moved up to %.3f A from its start, RMSD to the crystal pose %.2f A" % (
            name, stage, (out[0] - data["compound"].node_coords).norm(dim=-1).max(),
            float(((out[0] - data.coords) ** 2).sum(-1).mean().sqrt())))
    with torch.no_grad():
        out = model.inference(_hetero_to_ref(data))
    save["inf_coords"] = _np(out[0])
    _loss_and_gradients(model, data, save, name)
    for key, st in data._stores.items():
        kname = key if isinstance(key, str) else "|".join(key)
        for k, v in st.items():
            if torch.is_tensor(v):
                save["d_%s::%s" % (kname, k)] = _np(v)
    for k, v in data._glob.items():
        if torch.is_tensor(v):
            save["d_::%s" % k] = _np(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def main_plus():
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    mods = refshim.load_reference("FABind_plus")
    golden_stack_plus(mods, "plus_stack_tiny_it1", [(24, 7), (31, 5)], 32, 2, 1, seed=10)
    golden_stack_plus(mods, "plus_stack_tiny_it2", [(40, 9), (26, 6), (33, 12)], 64, 3, 2, seed=11)
    golden_model_plus(mods, "plus_model_tiny", [(70, 8), (85, 6)], 64, 32, 2, 2, seed=12)
    golden_model_plus_sampling(mods, "plus_model_sampling_tiny", [(120, 8), (140, 11)], 64, 32, 2, 1, seed=13)
    golden_plus_loss(mods, "plus_loss_tiny", seed=14)
    golden_stack_plus_grad(mods, "plus_stack_tiny_grad", [(24, 7), (31, 5)], 32, 2, seed=15)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "plus":
        return main_plus()
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(1)
        return golden_eval_metrics(refshim.load_reference("FABind"), "eval_metrics", seed=31)
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(1)
        return golden_data_builder(refshim.load_reference("FABind"), "data_builder", seed=41)
    if len(sys.argv) > 1 and sys.argv[1] == "6g3c":
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(1)
        return golden_model_6g3c(refshim.load_reference("FABind"), "model_6g3c", seed=51)
    if len(sys.argv) > 1 and sys.argv[1] == "post":
        os.makedirs(OUT, exist_ok=True)
        return golden_post_optim("post_optim", seed=21)
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    mods = refshim.load_reference("FABind")
    golden_stack(mods, "stack_tiny_it1", [(24, 7), (31, 5)], 32, 2, 1, seed=0)
    golden_stack(mods, "stack_tiny_it3", [(24, 7), (31, 5)], 32, 2, 3, seed=1)
    golden_stack(mods, "stack_mid_it2", [(100, 40), (90, 22)], 64, 2, 2, seed=2)
    golden_stack_grad(mods, "stack_tiny_grad", [(24, 7), (31, 5)], 32, 2, seed=3)
    golden_model(mods, "model_tiny", [(70, 8), (85, 6)], 64, 32, 2, 2, seed=4)


if __name__ == "__main__":
    main()
