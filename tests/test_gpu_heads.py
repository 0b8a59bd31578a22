"""GPU: the two fused head kernels of round 6 (csrc/heads.hip) against the plain-torch statement of the reference's ops.
K17 pocket centre  = FABind/fabind/models/model.py:146-158 (+ utils/utils.py:687-699); K20 six-term loss = main_fabind.py:398-417.
Values AND gradients (the kernels' hand-written adjoints against torch autograd of the reference formulas); fp32: 1e-6-grade."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_center(logits, mask, xyz, tau, hard, noise):
    pt = logits.sigmoid().unsqueeze(-1)
    logp = torch.log(torch.clamp(torch.cat([1. - pt, pt], dim=-1), min=1e-6, max=1 - 1e-6))
    g = logp if noise is None else logp + noise
    y_soft = (g / tau).softmax(-1)
    if hard:
        idx = y_soft.max(-1, keepdim=True)[1]
        y = torch.zeros_like(logp).scatter_(-1, idx, 1.0) - y_soft.detach() + y_soft
    else:
        y = y_soft
    w = (y[:, :, 1] * mask).unsqueeze(-1)
    return (w * xyz).sum(1) / w.sum(1)


def ref_loss(coords, y_pred, y_by, logits, center, coords_true, dis_map, cls, mask, center_true, w):
    terms = dict(
        pocket_cls=w['cls'] * F.binary_cross_entropy_with_logits(logits, cls.float()) * (mask.numel() / mask.sum()),
        pocket_center=w['center'] * F.huber_loss(center, center_true, delta=w['delta']),
        contact=w['pair'] * F.mse_loss(y_pred, dis_map), contact_by_pred=w['pair'] * F.mse_loss(y_by, dis_map),
        distill=w['distill'] * F.mse_loss(y_by, y_pred), coord=w['coord'] * F.smooth_l1_loss(coords, coords_true))
    return sum(terms.values()), terms


def _head_inputs(B, L, seed, spread=4.0):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(3, L // 3), L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None])
    logits = torch.randn(B, L, generator=g) * spread          # (spread 4 -> some sigmoids beyond the 1e-6 clamp at +-13.8? no: tails to ~1e-5)
    logits[0, 0], logits[0, 1] = 20.0, -20.0                    # both clamp branches are exercised
    xyz = torch.randn(B, L, 3, generator=g) * 15.0
    return logits.to(DEV), mask.to(DEV), xyz.to(DEV)


@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("noisy", [False, True])
def test_pocket_center_matches_reference_ops(hard, noisy):
    from fabind_amd import ops
    for B, L, tau in ((3, 37, 1.0), (5, 1500, 0.7)):
        logits, mask, xyz = _head_inputs(B, L, seed=B)
        noise = None
        if noisy:
            torch.manual_seed(11)
            noise = -torch.empty(B, L, 2, device=DEV).exponential_().log()
        l1 = logits.clone().requires_grad_(True)
        l2 = logits.clone().requires_grad_(True)
        c_ref = ref_center(l1, mask, xyz, tau, hard, noise)
        c = ops.pocket_center(l2 * mask, mask, xyz, tau=tau, hard=hard, noise=noise)          # (the model hands over masked logits)
        assert c.shape == (B, 3)
        assert float((c - c_ref).abs().max()) <= 2e-5 * max(1.0, float(c_ref.abs().max())), (B, L, hard, noisy)
        gc = torch.randn(B, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
        (c_ref * gc).sum().backward()
        (c * gc).sum().backward()
        ref = l1.grad * mask
        err = float((l2.grad - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        assert err < 2e-4, (B, L, hard, noisy, err)
        assert float(l2.grad[~mask].abs().max()) == 0.0 if (~mask).any() else True


def test_pocket_center_is_bit_repeatable():
    from fabind_amd import ops
    logits, mask, xyz = _head_inputs(8, 1500, seed=2)
    a = ops.pocket_center(logits, mask, xyz)
    for _ in range(5):
        assert torch.equal(a, ops.pocket_center(logits, mask, xyz))


def _loss_inputs(n_atoms, n_pair, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    coords, coords_true = r(n_atoms, 3) * 3, r(n_atoms, 3) * 3            # |d| on both sides of the SmoothL1 knee
    y_pred, y_by, dis_map = torch.rand(n_pair, generator=g) * 10, torch.rand(n_pair, generator=g) * 10, torch.rand(n_pair, generator=g) * 10
    lens = torch.randint(max(2, L // 2), L + 1, (B,), generator=g)
    mask = torch.arange(L)[None, :] < lens[:, None]
    logits = r(B, L) * 3 * mask
    cls = ((torch.rand(B, L, generator=g) < 0.2) & mask).long()
    center, center_true = r(B, 3) * 4, r(B, 3) * 4                         # |d| on both sides of delta = 3
    return [t.to(DEV) for t in (coords, y_pred, y_by, logits, center, coords_true, dis_map, cls, mask, center_true)]


W = dict(coord=1.0, pair=1.0, distill=1.0, cls=1.0, center=0.05, delta=3.0)


@pytest.mark.parametrize("shape", [(40, 1500 * 40, 1, 1500), (77, 9113, 3, 211), (2560, 64 * 60000, 64, 1500), (5, 0, 2, 9)])
@pytest.mark.parametrize("cls_dtype", [torch.int64, torch.float32])
def test_six_term_loss_and_gradient_seeds_match_reference_ops(shape, cls_dtype):
    from fabind_amd import ops
    n_atoms, n_pair, B, L = shape
    if n_pair > 10 ** 6 and cls_dtype != torch.int64:
        pytest.skip("the large shape once")
    t = _loss_inputs(n_atoms, n_pair, B, L, seed=n_atoms)
    t[7] = t[7].to(cls_dtype)
    w = dict(W, coord=1.5, pair=0.7, distill=1.3, cls=0.9)
    leaves_ref = [x.clone().requires_grad_(True) for x in t[:5]]
    leaves = [x.clone().requires_grad_(True) for x in t[:5]]
    if n_pair == 0:                          # F.mse_loss of empty tensors is NaN in torch; the kernel defines the empty terms as 0
        loss, terms = ops.six_term_loss(*leaves, *t[5:], w)
        assert float(terms["contact"]) == 0.0 and float(terms["distill"]) == 0.0 and np.isfinite(float(loss))
        loss.backward()
        assert all(x.grad is not None and bool(torch.isfinite(x.grad).all()) for x in (leaves[0], leaves[3], leaves[4]))
        return
    loss_ref, terms_ref = ref_loss(*leaves_ref, *t[5:], w)
    loss, terms = ops.six_term_loss(*leaves, *t[5:], w)
    assert list(terms) == ["pocket_cls", "pocket_center", "contact", "contact_by_pred", "distill", "coord"]
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))                     # north_star: losses to 1e-5 relative
    for k in terms:
        assert abs(float(terms[k]) - float(terms_ref[k])) <= 1e-5 * max(abs(float(terms_ref[k])), 1e-3), k
    # gradients: of the total, and of a weighting of the single terms (g_terms path)
    loss_ref.backward()
    loss.backward()
    for a, b, n in zip(leaves, leaves_ref, ("coords", "y_pred", "y_by", "logits", "center")):
        err = float((a.grad - b.grad).abs().max()) / max(float(b.grad.abs().max()), 1e-20)
        assert err < 1e-5, (n, err)
    for x in leaves + leaves_ref:
        x.grad = None
    cw = dict(pocket_cls=0.3, pocket_center=-1.2, contact=2.0, contact_by_pred=0.0, distill=0.5, coord=1.1)
    _, terms_ref = ref_loss(*leaves_ref, *t[5:], w)
    l2, terms = ops.six_term_loss(*leaves, *t[5:], w)
    (sum(cw[k] * terms_ref[k] for k in cw)).backward()
    (sum(cw[k] * terms[k] for k in cw) + 0.0 * l2).backward()
    for a, b, n in zip(leaves, leaves_ref, ("coords", "y_pred", "y_by", "logits", "center")):
        err = float((a.grad - b.grad).abs().max()) / max(float(b.grad.abs().max()), 1e-20)
        assert err < 1e-5, (n, err)


def test_six_term_loss_is_bit_repeatable_and_rearms_its_ticket():
    from fabind_amd import ops
    t = _loss_inputs(2560, 640000, 64, 300, seed=4)
    a, ta = ops.six_term_loss(*t, W)
    for _ in range(6):
        b, tb = ops.six_term_loss(*t, W)
        assert torch.equal(a, b) and all(torch.equal(ta[k], tb[k]) for k in ta)
    assert all(int(v.item()) == 0 for v in ops._LOSS_TICKET.values())


# ---- the distance head's pair operations on per-complex blocks (csrc/heads.hip) -------------------------------------------------------
def _blocks(kcnt, ncnt):
    from fabind_amd import ops
    bl = ops.PairBlocks(kcnt, ncnt, torch.device(DEV))
    pi, ci = [], []
    p0 = c0 = 0
    for P, C in zip(kcnt, ncnt):
        for i in range(P):
            pi += [p0 + i] * C
            ci += list(range(c0, c0 + C))
        p0, c0 = p0 + P, c0 + C
    return bl, torch.tensor(pi, device=DEV), torch.tensor(ci, device=DEV)


@pytest.mark.parametrize("kcnt,ncnt", [([5], [3]), ([130, 7, 260], [9, 41, 2]), ([300, 129], [300, 5]), ([1500] * 3, [40] * 3)])
def test_pair_dist_matches_cdist_clamp_and_its_autograd(kcnt, ncnt):
    """y_pred_by_coords of reference model.py:349,363-365: clamp(5 * |xp_i / 5 - x_j|, 0, 10) over the valid pairs, and its gradient to the
    ligand coordinates (clamp active on both sides, a zero distance)."""
    from fabind_amd import ops
    bl, pi, ci = _blocks(kcnt, ncnt)
    g = torch.Generator().manual_seed(sum(kcnt))
    xp = (torch.randn(sum(kcnt), 3, generator=g) * 1.2).to(DEV)                    # normalised coordinates: distances 0 ... ~4 (x 5 A: clamp at 10 bites)
    xc = (torch.randn(sum(ncnt), 3, generator=g) * 0.6).to(DEV)
    xc[0] = xp[0]                                                                  # a zero distance: gradient 0, not NaN
    a, b = xc.clone().requires_grad_(True), xc.clone().requires_grad_(True)
    ref = (5.0 * (xp[pi] - a[ci]).norm(dim=-1)).clamp(0, 10)
    got = ops.pair_dist(xp, b, bl, scale=5.0, lo=0.0, hi=10.0)
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-5
    assert float((ref >= 10).float().mean()) > 0.01                                # the clamp is exercised
    dy = torch.randn(ref.shape, generator=g).to(DEV)
    (ref * dy).sum().backward()
    (got * dy).sum().backward()
    assert bool(torch.isfinite(b.grad).all())
    err = float((a.grad.nan_to_num() - b.grad).abs().max()) / float(a.grad.nan_to_num().abs().max())
    assert err < 2e-5, err
    got2 = ops.pair_dist(xp, b.detach().requires_grad_(True), bl, scale=5.0)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("W", [64, 128, 512])
@pytest.mark.parametrize("kcnt,ncnt", [([5], [3]), ([130, 7, 260], [9, 41, 2]), ([200, 129], [45, 5])])
def test_rows_hadamard_block_adjoint_matches_autograd(W, kcnt, ncnt):
    """einsum('bik,bjk->bijk')[z_mask] (reference model.py:355) under autograd: the block-descriptor adjoint against torch's, and against
    the CSR-walk adjoint of rounds 4-5 (same dout, bf16)."""
    from fabind_amd import engine, ops
    engine.set_precision("bf16")
    try:
        bl, pi, ci = _blocks(kcnt, ncnt)
        npk = sum(kcnt)
        g = torch.Generator().manual_seed(W + npk)
        t0 = torch.randn(npk + sum(ncnt), W, generator=g).to(DEV)
        dout = torch.randn(bl.n_pairs, W, generator=g).to(DEV).to(torch.bfloat16)
        t1 = t0.clone().requires_grad_(True)
        hd = ops.rows_hadamard(t1, pi, npk + ci, a_sorted=True, blocks=bl, n_a=npk)
        assert hd.dtype == torch.bfloat16
        hd.backward(dout)
        t2 = t0.clone().requires_grad_(True)
        (t2[pi] * t2[npk + ci]).backward(dout.float())
        err = float((t1.grad - t2.grad).abs().max()) / float(t2.grad.abs().max())
        assert err < 1e-5, err
        t3 = t0.clone().requires_grad_(True)
        ops.rows_hadamard(t3, pi, npk + ci, a_sorted=True).backward(dout)            # the CSR walk
        assert float((t1.grad - t3.grad).abs().max()) / float(t3.grad.abs().max()) < 1e-5
        t4 = t0.clone().requires_grad_(True)
        ops.rows_hadamard(t4, pi, npk + ci, a_sorted=True, blocks=bl, n_a=npk).backward(dout)
        assert torch.equal(t4.grad, t1.grad)                                         # bit-repeatable
    finally:
        engine.set_precision("fp32")
