"""GPU: the HIP docking stack (through libfabind_hip.so) against the golden vectors captured from the
reference and against the CPU oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

import fabind_oracle as orc
from helpers import load_npz, rmsd, stack_inputs, weights

pytestmark = pytest.mark.gpu
STACKS = ["stack_tiny_it1", "stack_tiny_it3", "stack_mid_it2"]


def _args(hidden, layers, n_iter):
    from argparse import Namespace
    return Namespace(
        mode=5, n_iter=n_iter, mean_layers=layers, hidden_size=hidden, pocket_pred_hidden_size=32,
        pocket_pred_layers=1, pocket_pred_n_iter=1, refine="refine_coord", coordinate_scale=5.0,
        geometry_reg_step_size=0.001, rm_layernorm=True, add_attn_pair_bias=True, explicit_pair_embed=True,
        add_cross_attn_layer=True, norm_type="per_sample", random_n_iter=True, center_dist_threshold=4.0,
        stage_prob=0.25, distmap_pred="mlp", use_esm2_feat=True, esm2_concat_raw=False, inter_cutoff=10.0,
        intra_cutoff=8.0, pocket_radius=20.0, gs_tau=1.0, gs_hard=False, local_eval=False,
        train_pred_pocket_noise=0.0, compound_coords_init_mode="pocket_center_rdkit", ablation_no_attention=False,
        ablation_no_attention_with_cross_attn=False, keep_trig_attn=False, opm=False, rm_F_norm=False,
        fix_pocket=False, rm_LAS_constrained_optim=False)


def _build_stack(g, dev):
    from fabind_amd.models.att_model import EfficientMCAttModel
    hidden, layers, n_iter, _ = [int(v) for v in g["cfg"]]
    m = EfficientMCAttModel(_args(hidden, layers, n_iter), hidden, hidden, 1, n_layers=layers, n_iter=n_iter,
                            normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0)
    m.load_state_dict(weights(g), strict=True)
    return m.to(dev).eval()


def _run(m, inp, dev):
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    with torch.no_grad():
        return m(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"],
                 t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])


def test_edges_match_reference_sets():
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    g = load_npz("stack_mid_it2")
    inp = stack_inputs(g)
    lay = engine.Layout(inp["batch_id"].to(dev), inp["segment_id"].to(dev))
    br = inp["compound_edge_index"][0].to(torch.int32).to(dev)
    bc = inp["compound_edge_index"][1].to(torch.int32).to(dev)
    gr = engine.Graph(lay, inp["X"][:, 0].contiguous().to(dev), br, bc, lay.ranges(br), 8.0 / 5.0, 10.0 / 5.0)
    ref_ctx = np.concatenate([g["in_compound_edge_index"], g["cap_ctx_edges_noBond"]], 1)
    mine_ctx = np.stack([gr.row_ctx.cpu().numpy(), gr.col_ctx.cpu().numpy()])
    key = lambda e: sorted(map(tuple, e.T.tolist()))
    assert key(mine_ctx) == key(ref_ctx)
    mine_int = np.stack([gr.row_int.cpu().numpy(), gr.col_int.cpu().numpy()])
    assert np.array_equal(mine_int, g["cap_inter_edges"])           # inter edges: identical order too
    assert np.all(np.diff(mine_ctx[0]) >= 0)                         # row-sorted CSR


def test_edges_at_the_cutoff_boundary_are_consistent():
    """Regression: thousands of pair distances sit exactly on (or one ulp around) the inter / intra cut-offs.  The count
    pass, the fill pass and both directions of an edge must take the same decision -- otherwise a CSR slot stays
    uninitialised or an edge has no mirror (seen in FABind+ sampling as a device fault before the distance predicate was
    compiled with floating-point contraction off)."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    B, P, C = 6, 400, 24
    X, batch, seg = [], [], []
    for b in range(B):
        # lattice with spacing 0.4 (normalised units): (5,0,0), (3,4,0), ... x 0.4 = 2.0 = inter cut-off exactly,
        # (4,0,0) x 0.4 = 1.6 = intra cut-off; plus offsets of a few ulps
        lig = rng.integers(-3, 4, size=(C, 3)) * 0.4
        prot = rng.integers(-6, 7, size=(P, 3)) * 0.4
        x = np.concatenate([np.zeros((1, 3)), lig, np.zeros((1, 3)), prot]).astype(np.float32)
        x = x * (1.0 + rng.integers(-2, 3, size=x.shape) * 6e-8).astype(np.float32) + np.float32(b * 0.1)
        X.append(x); batch += [b] * len(x); seg += [0] * (C + 1) + [1] * (P + 1)
    x = torch.from_numpy(np.concatenate(X)).to(dev)
    lay = engine.Layout(torch.tensor(batch, device=dev), torch.tensor(seg, dtype=torch.float32, device=dev))
    br = torch.zeros(0, dtype=torch.int32, device=dev)
    gr = engine.Graph(lay, x, br, br, lay.ranges(br), 8.0 / 5.0, 10.0 / 5.0)
    N = lay.N
    for row, col, rp in ((gr.row_int, gr.col_int, gr.rp_int), (gr.row_ctx, gr.col_ctx, gr.rp_ctx)):
        r, c = row.long(), col.long()
        assert int(c.min()) >= 0 and int(c.max()) < N                       # every slot the count pass reserved is filled
        owner = torch.repeat_interleave(torch.arange(N, device=dev), (rp[1:] - rp[:-1]).long())
        assert torch.equal(owner, r)
    r, c = gr.row_int.long(), gr.col_int.long()
    assert torch.equal(torch.sort(r * N + c)[0], torch.sort(c * N + r)[0])   # every inter edge has its mirror
    assert gr.E_int > 1000
    m = gr.mirror.long()
    assert torch.equal(r[m], c) and torch.equal(c[m], r)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", STACKS)
def test_stack_forward_fp32_matches_reference(name, mode):
    """north_star gate: ligand coordinates within 1e-4 A RMSD of the reference CPU path -- in fp32 mode (exact-fp32 MFMA) and in
    bf16x3 mode (fp32 storage, split-bf16 contractions: the fast gate-meeting mode)."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    engine.set_precision(mode)
    g = load_npz(name)
    m = _build_stack(g, dev)
    cap = {}
    engine.DEBUG_CAPTURE = cap
    try:
        X, H = _run(m, stack_inputs(g), dev)
    finally:
        engine.DEBUG_CAPTURE = None
        engine.set_precision("fp32")
    if name == "stack_tiny_it1":
        for k in ("gcl_0.h", "gcl_0.x", "att_0.h", "att_0.x", "att_0.alpha", "gcl_1.h", "att_1.x"):
            ref = g["cap_" + k]
            got = cap[k].cpu().numpy().reshape(ref.shape)
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
    lig = g["in_mask"]
    assert rmsd(X.cpu().numpy()[lig] * 5, g["out_X_f32"][lig] * 5) < 1e-4
    assert np.abs(H.cpu().numpy() - g["out_H_f32"]).max() <= 1e-4 * max(1.0, np.abs(g["out_H_f32"]).max())


@pytest.mark.parametrize("name", STACKS)
def test_stack_forward_bf16_close(name):
    """bf16 MFMA operands: the gap to the reference is reported and bounded at 2x the largest value measured on each fixture
    (1.3e-4 / 3.4e-4 / 1.8e-4 A in round 1, 1.2e-4 / 2.9e-4 / 1.4e-4 A with the MFMA cross attention of round 2); bf16 does not
    meet the 1e-4 A gate on these fixtures -- the gate is asserted in fp32 mode."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    engine.set_precision("bf16")
    try:
        g = load_npz(name)
        X, H = _run(_build_stack(g, dev), stack_inputs(g), dev)
    finally:
        engine.set_precision("fp32")
    lig = g["in_mask"]
    gap = rmsd(X.cpu().numpy()[lig] * 5, g["out_X_f32"][lig] * 5)
    print("bf16 ligand RMSD gap vs reference [A]:", name, gap)
    assert gap < {"stack_tiny_it1": 2.6e-4, "stack_tiny_it3": 6.8e-4, "stack_mid_it2": 3.6e-4}[name]


def test_stack_forward_vs_oracle_larger():
    """Seeded synthetic batch beyond the fixtures (4 complexes, 150/40, H=128, 2 layers, 2 iterations)."""
    from fabind_amd import engine, synthetic
    from fabind_amd.models.att_model import EfficientMCAttModel
    dev = torch.device("cuda:0")
    engine.set_precision("fp32")
    torch.manual_seed(11)
    H, L, it = 128, 2, 2
    m = EfficientMCAttModel(_args(H, L, it), H, H, 1, n_layers=L, n_iter=it, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("coord_mlp.2.weight"):
                p.mul_(300.0)
    inp = synthetic.make_stack_batch([(150, 40), (120, 33), (170, 12), (90, 25)], H, seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    Xr, Hr = orc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"],
                               inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"], L, it)
    X, Hh = _run(m.to(dev), inp, dev)
    lig = inp["mask"].numpy()
    assert rmsd(X.cpu().numpy()[lig] * 5, Xr.numpy()[lig] * 5) < 1e-4
    assert (Hh.cpu() - Hr).abs().max() <= 1e-4 * max(1.0, float(Hr.abs().max()))


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_stack_gradients_match_reference(mode):
    """Backward through the HIP kernels: d(loss)/d(parameters, input H) vs the reference's autograd (fp32 mode and the
    split-bf16 mode, same gates)."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    engine.set_precision(mode)
    try:
        g = load_npz("stack_tiny_grad")
        m = _build_stack(g, dev)
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
        Hin = inp["H"].clone().requires_grad_(True)
        X, H = m(inp["X"].clone(), Hin, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                 inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"])
        loss = (X * torch.from_numpy(g["cot_X"]).to(dev)).sum() + (H * torch.from_numpy(g["cot_H"]).to(dev)).sum()
        assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
        loss.backward()
    finally:
        engine.set_precision("fp32")
    # bf16x3: the forward quantities above are held to the same gates in both modes; its GRADIENTS carry single-bf16 weight-gradient
    # contractions and ReLU-derivative flips (tests/test_gpu_headline.py measures both): 2e-2 per tensor instead of 3e-3
    gtol = 3e-3 if mode == "fp32" else 2e-2
    ref = g["grad_in_H"]
    assert np.abs(Hin.grad.cpu().numpy() - ref).max() <= (2e-3 if mode == "fp32" else 2e-2) * np.abs(ref).max()
    nograd = set(str(s) for s in g["nograd"])
    bad = []
    for n, p in m.named_parameters():
        if n in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        ref = g["grad_" + n]
        if p.grad is None:
            # a per-layer constant added to every logit of a softmax row has exactly zero gradient; the HIP path
            # drops those constants (pair_transition.linear_2.bias, attn_bias_proj.bias), the reference keeps ~1e-9 noise
            assert np.abs(ref).max() < 1e-6, n
            continue
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        if not err <= gtol * np.abs(ref).max() + 1e-7:
            bad.append((n, float(err), float(np.abs(ref).max())))
    assert not bad, bad


def _double_backward(prec, dev, g):
    """Two backward passes over one retained graph -> [(name, first, second)] for the input and every parameter."""
    from fabind_amd import engine
    engine.set_precision(prec)
    try:
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
        return [(n, a, b) for (n, _), a, b in zip(named, first, second)]
    finally:
        engine.set_precision("fp32")


def _passes_differ(triples):
    """Names of the gradients that differ between the two passes beyond summation-order noise.  Float atomics in some adjoints
    reorder sums between runs: measured over 20 repetitions (tests/probe_retained_graph_spread.py) the passes differ by at most
    4.7e-8 (fp32) / 3.6e-11 (bf16) of the model's LARGEST gradient magnitude; relative to a tensor's own maximum the figure is
    meaningless for the biases whose true gradient is zero (softmax shift invariance: 1.8e-4 of round-off).  Hence two terms:
    1e-3 of the tensor's own maximum + 1e-6 of the largest one.  A stale shared buffer changes the affected gradients by 100 %."""
    gmax = max(float(a.abs().max()) for _, a, _ in triples if a is not None)
    bad = []
    for n, a, b in triples:
        if (a is None) != (b is None):
            bad.append(n)
        elif a is not None and float((a - b).abs().max()) > 1e-3 * float(a.abs().max()) + 1e-6 * gmax:
            bad.append(n)
    return bad


def test_backward_twice_over_a_retained_graph_gives_the_same_gradients(monkeypatch):
    """The shared gradient buffer of the per-node pair projections (ops.GradSink: several accumulating adjoints add into ONE
    buffer instead of autograd summing padded copies) must be released after every backward pass: a second backward over the
    retained graph has to reproduce the first one's gradients, not add to (or be swallowed by) a stale buffer.  fp32 and bf16
    (separate adjoint paths).  The test checks itself: with the release disabled the same comparison must fail."""
    from fabind_amd import ops
    dev = torch.device("cuda:0")
    g = load_npz("stack_tiny_grad")
    for prec in ("fp32", "bf16"):
        triples = _double_backward(prec, dev, g)
        assert any(a is not None for _, a, _ in triples)
        assert _passes_differ(triples) == [], (prec, _passes_differ(triples))
    # mutation: keep the buffer reference across passes -> the second pass must come out different
    monkeypatch.setattr(ops._SinkOwner, "backward", staticmethod(lambda ctx, grad: (grad, None)))
    assert _passes_differ(_double_backward("fp32", dev, g)) != [], "the comparison cannot see a stale shared gradient buffer"


def _random_stack(H, L, it, seed):
    from fabind_amd.models.att_model import EfficientMCAttModel
    torch.manual_seed(seed)
    m = EfficientMCAttModel(_args(H, L, it), H, H, 1, n_layers=L, n_iter=it, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("coord_mlp.2.weight"):
                p.mul_(300.0)
    return m


def _vs_oracle(m, inp, L, it, tol=1e-4):
    dev = torch.device("cuda:0")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    Xr, Hr = orc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"],
                               inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"], L, it)
    X, Hh = _run(m.to(dev), inp, dev)
    assert rmsd(X.cpu().numpy() * 5, Xr.numpy() * 5) < tol
    assert (Hh.cpu() - Hr).abs().max() <= tol * max(1.0, float(Hr.abs().max()))
    return m


def test_edge_case_no_inter_edges_uses_reference_fallback():
    """Ligand far from the protein: the batch has no inter edge -> one fake symmetric pair (att_model.py:85-86)."""
    from fabind_amd import engine, synthetic
    engine.set_precision("fp32")
    inp = synthetic.make_stack_batch([(40, 6), (35, 5)], 32, seed=9)
    lig = (inp["segment_id"] == 0) & ~inp["is_global"]
    inp["X"][lig] += 50.0                                   # 250 A away
    m = _vs_oracle(_random_stack(32, 2, 2, 21), inp, 2, 2)
    assert m.last_graph.E_int == 2


def test_edge_case_single_complex_pocket_model_shape():
    """B = 1 and the pocket-prediction configuration (hidden 128, 1 layer, 1 iteration)."""
    from fabind_amd import engine, synthetic
    engine.set_precision("fp32")
    _vs_oracle(_random_stack(128, 1, 1, 22), synthetic.make_stack_batch([(260, 31)], 128, seed=10), 1, 1)


def test_edge_case_tiny_and_ragged_complexes():
    """Very ragged batch incl. a 2-atom ligand and a 6-residue protein."""
    from fabind_amd import engine, synthetic
    engine.set_precision("fp32")
    _vs_oracle(_random_stack(64, 2, 3, 23), synthetic.make_stack_batch([(6, 2), (200, 45), (17, 3), (64, 64)], 64, seed=11), 2, 3)


def test_construct_edges_api_matches_reference_sets():
    """ComplexGraph.construct_edges keeps the reference's signature and edge sets."""
    from fabind_amd import engine
    from fabind_amd.models.att_model import ComplexGraph
    dev = torch.device("cuda:0")
    g = load_npz("stack_tiny_it1")
    inp = stack_inputs(g)
    cg = ComplexGraph(_args(32, 2, 1), normalize_coord=lambda x: x / 5.0)
    ctx, inter, (rb, ro) = cg.construct_edges(inp["X"].to(dev), inp["batch_id"].to(dev), inp["segment_id"].to(dev),
                                              inp["is_global"].to(dev))
    key = lambda e: sorted(map(tuple, np.asarray(e).T.tolist()))
    assert key(ctx.cpu().numpy()) == key(g["cap_ctx_edges_noBond"])
    assert np.array_equal(inter.cpu().numpy(), g["cap_inter_edges"])
    assert rb.shape[0] == inter.shape[1] // 2


def test_stack_gradients_bf16_close():
    """bf16 training path (stored activation derivatives, TN weight-gradient GEMM, ragged pair-bias adjoint):
    gradients stay close to the reference's fp32 autograd (loose bound; bf16 operands)."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    engine.set_precision("bf16")
    try:
        g = load_npz("stack_tiny_grad")
        m = _build_stack(g, dev)
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
        Hin = inp["H"].clone().requires_grad_(True)
        X, H = m(inp["X"].clone(), Hin, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                 inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"])
        loss = (X * torch.from_numpy(g["cot_X"]).to(dev)).sum() + (H * torch.from_numpy(g["cot_H"]).to(dev)).sum()
        loss.backward()
    finally:
        engine.set_precision("fp32")
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-2 * abs(float(g["loss"]))
    ref = g["grad_in_H"]
    assert np.abs(Hin.grad.cpu().numpy() - ref).max() <= 6e-2 * np.abs(ref).max()
    nograd = set(str(s) for s in g["nograd"])
    num = den = 0.0
    rows = []
    for n, p in m.named_parameters():
        if n in nograd:
            continue
        ref = g["grad_" + n]
        if p.grad is None:
            # same rule as the fp32 test above: only the softmax-shift-invariant constants may come without a gradient (the
            # bf16 mode runs different backward kernels, so this is checked here too instead of being skipped)
            assert np.abs(ref).max() < 1e-6, n
            continue
        d = p.grad.float().cpu().numpy() - ref
        num += float((d ** 2).sum()); den += float((ref ** 2).sum())
        rows.append((float(np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30)), n))
    rows.sort(reverse=True)
    glob = (num / den) ** 0.5
    print("bf16 gradients on stack_tiny_grad: whole-gradient l2-rel %.3e, worst tensors %s" % (glob, rows[:3]))
    # On this 55-node, hidden-32 fixture single entries of the small tensors flip with ReLU masks under bf16 rounding (max-entry
    # errors of 10-17 % with either cross-attention kernel, tools/probes/bf16_grad_noise.py); what is bounded is the error of the
    # whole gradient vector (measured 4.5e-3 fp32 attention / 4.7e-3 MFMA attention) and, loosely, of each tensor.
    assert glob <= 2e-2
    assert rows[0][0] <= 0.5, rows[:5]


def test_train_mode_dropout_statistics():
    """Train mode: dropout (p = 0.1, reference module defaults) is active, seeded, and vanishes for p = 0."""
    from fabind_amd import engine, synthetic
    from fabind_amd.models.att_model import EfficientMCAttModel
    dev = torch.device("cuda:0")
    engine.set_precision("fp32")
    H, L = 64, 2
    inp = synthetic.make_stack_batch([(50, 9), (64, 12)], H, seed=4)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    a = _args(H, L, 1)
    a.random_n_iter = False

    def run(m, seed):
        torch.manual_seed(seed)
        X, Hh = m(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"],
                  t["LAS_edge_index"], t["coord_LAS"])
        return X.detach(), Hh
    torch.manual_seed(0)
    m = EfficientMCAttModel(a, H, H, 1, n_layers=L, n_iter=1, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).to(dev)
    m.eval()
    Xe, He = run(m, 1)
    m.train()
    X1, H1 = run(m, 1)
    X1b, H1b = run(m, 1)
    X2, H2 = run(m, 2)
    assert torch.equal(H1, H1b)                                   # same seed -> same masks
    assert (H1 - H2).abs().max() > 1e-4 and (H1 - He).abs().max() > 1e-4
    H1.sum().backward()                                           # gradients flow through the dropout path
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    # dropout probability 0 in train mode == eval mode
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    X0, H0 = run(m, 3)
    assert torch.allclose(H0, He, rtol=1e-5, atol=1e-5) and torch.allclose(X0, Xe, rtol=1e-5, atol=1e-6)


def test_fabind_layer_forward_composes_to_the_stack():
    """north_star API: `FABindLayer.forward(h, x, ctx)` (one loop body of MCAttEGNN.forward, egnn.py:402-449).
    Layer 0 reproduces the reference captures; all layers + out_layer (through MC_E_GCL.forward with the reference's
    signature) + linear_out reproduce the stack output, and so does MCAttEGNN.forward(h, x, ctx=ctx)."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    engine.set_precision("fp32")
    g = load_npz("stack_tiny_it1")
    m = _build_stack(g, dev)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in stack_inputs(g).items()}
    args9 = (inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"], inp["compound_edge_index"],
             inp["LAS_edge_index"], inp["coord_LAS"])
    with torch.no_grad():
        ctx = m.context(inp["X"].clone(), *args9)
        h = torch.nn.functional.linear(inp["H"], m.gnn.linear_in.weight, m.gnn.linear_in.bias)
        x = inp["X"].clone()                                                  # [N,1,3]
        layers = m.gnn.layers()
        assert len(layers) == m.gnn.n_layers
        h, x, alpha = layers[0](h, x, ctx, return_attention=True)
        for k, got in (("att_0.h", h), ("att_0.alpha", alpha)):
            ref = g["cap_" + k]
            assert np.abs(got.cpu().numpy().reshape(ref.shape) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
        for layer in layers[1:]:
            h, x = layer(h, x, ctx)
        ei = torch.stack([ctx.g.row_ctx.long(), ctx.g.col_ctx.long()])
        ei = ei[:, torch.randperm(ei.shape[1], device=dev)]                   # the reference signature takes any edge order
        h, x = m.gnn.out_layer(h, ei, x, batch_id=inp["batch_id"])
        Hout = torch.nn.functional.linear(h, m.gnn.linear_out.weight, m.gnn.linear_out.bias)
        lig = g["in_mask"]
        assert rmsd(x.cpu().numpy()[lig] * 5, g["out_X_f32"][lig] * 5) < 1e-4
        assert np.abs(Hout.cpu().numpy() - g["out_H_f32"]).max() <= 1e-4 * max(1.0, np.abs(g["out_H_f32"]).max())
        H2, x2 = m.gnn(inp["H"], inp["X"].clone(), ctx=ctx)
        assert rmsd(x2.cpu().numpy()[lig] * 5, g["out_X_f32"][lig] * 5) < 1e-4
        assert np.abs(H2.cpu().numpy() - g["out_H_f32"]).max() <= 1e-4 * max(1.0, np.abs(g["out_H_f32"]).max())
        # (the reference's positional signature is served by the dense adapter: tests/test_gpu_dense_api.py)


def test_mc_e_gcl_forward_reference_signature_vs_oracle():
    """MC_E_GCL.forward(h, edge_index, coord, batch_id=...) on an arbitrary (unsorted, asymmetric) edge list against the
    CPU oracle's restatement of egnn.py:130-144, fp32 and bf16."""
    from argparse import Namespace
    from fabind_amd import engine
    from fabind_amd.models.egnn import MC_E_GCL
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    H, sizes = 64, [37, 5, 90]
    N = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    off = np.concatenate([[0], np.cumsum(sizes)])
    rows, cols = [], []
    for b, n in enumerate(sizes):
        e = n * 6
        rows.append(torch.randint(0, n, (e,), generator=gen) + int(off[b]))
        cols.append(torch.randint(0, n, (e,), generator=gen) + int(off[b]))
    row, col = torch.cat(rows), torch.cat(cols)
    keep = row != col
    row, col = row[keep], col[keep]
    h = torch.randn(N, H, generator=gen)
    x = torch.randn(N, 1, 3, generator=gen)
    torch.manual_seed(5)
    mod = MC_E_GCL(Namespace(rm_F_norm=False, norm_type="per_sample"), H, H, H, 1, coord_change_maximum=2.0, dropout=0.1)
    with torch.no_grad():
        mod.coord_mlp[2].weight.mul_(300.0)                                    # xavier gain 0.001 would hide coordinate errors
    sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    with torch.no_grad():
        h_ref, x_ref = orc.gcl_forward(sd, "", h, row, col, x[:, 0], batch, len(sizes), 2.0)
    mod = mod.to(dev).eval()
    for prec, tol in (("fp32", 1e-4), ("bf16", 3e-2)):
        engine.set_precision(prec)
        try:
            with torch.no_grad():
                h2, x2 = mod(h.to(dev), torch.stack([row, col]).to(dev), x.to(dev), batch_id=batch.to(dev))
        finally:
            engine.set_precision("fp32")
        assert x2.shape == x.shape
        assert (h2.cpu() - h_ref).abs().max() <= tol * max(1.0, float(h_ref.abs().max())), prec
        assert (x2[:, 0].cpu() - x_ref).abs().max() <= tol * max(1.0, float(x_ref.abs().max())), prec


def test_prefetched_layout_and_graph_are_picked_up_and_change_nothing():
    """engine.prefetch on a feeder stream (the batch layout and the graph of the input coordinates, att_model.py:209-214, built ahead
    of the stack call): the call on the same tensor objects takes both -- no second construction -- and returns bit-identical
    coordinates and features; a call on other tensor objects, or after X changed, builds its own."""
    from fabind_amd import engine
    dev = torch.device("cuda:0")
    g = load_npz("stack_mid_it2")
    m = _build_stack(g, dev)
    inp = stack_inputs(g)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    args = lambda X, b: (X, t["H"], b, t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"],
                         t["coord_LAS"])
    with torch.no_grad():
        X_ref, H_ref = m(*args(t["X"].clone(), t["batch_id"].clone()))
    side = torch.cuda.Stream(dev)
    built = []
    orig = engine.Graph.__init__

    def counting(self, *a, **k):
        built.append(1)
        return orig(self, *a, **k)
    engine.Graph.__init__ = counting
    try:
        with torch.cuda.stream(side):
            X0, b0 = t["X"].clone(), t["batch_id"].clone()
            engine.prefetch(m, X0, b0, t["segment_id"], t["compound_edge_index"])
            ev = torch.cuda.Event()
            ev.record(side)
        torch.cuda.current_stream(dev).wait_event(ev)
        n_pre = len(built)
        assert n_pre == 1 and len(engine._PREFETCHED) >= 1
        with torch.no_grad():
            X1, H1 = m(*args(X0, b0))
        n_iter = int(g["cfg"][2])
        assert len(built) - n_pre == n_iter - 1                          # the first pass took the prefetched graph
        assert not any(e[0]() is X0 for e in engine._PREFETCHED)         # consumed
        assert torch.equal(X1, X_ref) and torch.equal(H1, H_ref)
        # a prefetch for other tensor objects is not taken; a changed X invalidates its entry
        with torch.cuda.stream(side):
            X2, b2 = t["X"].clone(), t["batch_id"].clone()
            engine.prefetch(m, X2, b2, t["segment_id"], t["compound_edge_index"])
        side.synchronize()
        X2.add_(0.0)                                                      # version bump
        n0 = len(built)
        with torch.no_grad():
            X3, H3 = m(*args(X2, b2))
        assert len(built) - n0 == n_iter
        assert torch.equal(X3, X_ref) and torch.equal(H3, H_ref)
    finally:
        engine.Graph.__init__ = orig
        del engine._PREFETCHED[:]


def test_edge_case_rows_with_more_than_64_inter_edges():
    """A compact protein around the ligand: every ligand atom has > 64 inter edges (the row walks of the inter-edge attention load
    their per-edge scalars in batches of 64 edges, one per lane: this exercises a second and third batch) -- forward vs the oracle,
    and input gradients vs oracle autograd."""
    from fabind_amd import engine, synthetic
    engine.set_precision("fp32")
    inp = synthetic.make_stack_batch([(170, 9), (140, 12)], 32, seed=12, snap=False)
    pr = (inp["segment_id"] > 0.5) & ~inp["is_global"]
    lig = (inp["segment_id"] < 0.5) & ~inp["is_global"]
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for b in range(2):
            sel_l = lig & (inp["batch_id"] == b)
            sel_p = pr & (inp["batch_id"] == b)
            centre = inp["X"][sel_l].mean(0, keepdim=True)
            # residues in a 6.5 A ball around the ligand centre (coordinates are in units of 5 A): all within the 10 A inter cut-off
            # of every ligand atom after the ligand itself is shrunk to a 1.5 A ball
            inp["X"][sel_l] = centre + 0.3 * (torch.rand(int(sel_l.sum()), 1, 3, generator=g) - 0.5)
            inp["X"][sel_p] = centre + 1.3 * (torch.rand(int(sel_p.sum()), 1, 3, generator=g) - 0.5) * 2 * 0.75
    m = _vs_oracle(_random_stack(32, 2, 2, 24), inp, 2, 2, tol=2e-4)
    gr = m.last_graph
    deg = (gr.rp_int[1:] - gr.rp_int[:-1])
    assert int(deg.max()) > 128, int(deg.max())
    # gradients with respect to the node features through both layers
    dev = torch.device("cuda:0")
    sd = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
    Hr = inp["H"].clone().requires_grad_(True)
    Xr, Ho = orc.stack_forward(sd, "", inp["X"].clone(), Hr, inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                               inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"], 2, 1)
    ((Xr * Xr).sum() + (Ho * Ho).sum() * 1e-3).backward()
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    Hd = t["H"].clone().requires_grad_(True)
    m.n_iter = 1
    X, Hh = m(t["X"].clone(), Hd, t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"],
              t["LAS_edge_index"], t["coord_LAS"])
    ((X * X).sum() + (Hh * Hh).sum() * 1e-3).backward()
    err = float((Hd.grad.cpu() - Hr.grad).abs().max() / Hr.grad.abs().max())
    assert err <= 3e-3, err
