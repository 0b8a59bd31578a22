"""GPU: batched ligand post-optimisation kernel (csrc/post_optim.hip, through the C ABI) against fixtures produced by the
reference's own utils/post_optim_utils.py and against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import load_npz

pytestmark = pytest.mark.gpu


def _batch(g, dev):
    refs, preds, las, batch, off = [], [], [], [], 0
    for li, n in enumerate(g["sizes"]):
        refs.append(g["ref_%d" % li]); preds.append(g["pred_%d" % li]); las.append(g["las_%d" % li] + off)
        batch += [li] * int(n)
        off += int(n)
    t = lambda a, dt: torch.from_numpy(np.concatenate(a, -1 if a[0].ndim == 2 and a[0].shape[0] == 2 else 0)).to(dt).to(dev)
    return (torch.from_numpy(np.concatenate(refs)).to(dev), torch.from_numpy(np.concatenate(preds)).to(dev),
            torch.tensor(batch, device=dev), torch.from_numpy(np.concatenate(las, 1)).to(dev))


def test_post_optim_short_horizon_matches_reference_and_oracle():
    from fabind_amd.utils.post_optim_utils import post_optimize_compound_coords_batched
    from oracle.post_optim_oracle import post_optimize_compound_coords as orc
    dev = torch.device("cuda:0")
    g = load_npz("post_optim")
    ref, pred, batch, las = _batch(g, dev)
    x, loss, rmsd = post_optimize_compound_coords_batched(ref, pred, batch, total_epoch=5, LAS_edge_index=las)
    xn, lossn, _ = post_optimize_compound_coords_batched(ref, pred, batch, total_epoch=5, LAS_edge_index=None)
    off = 0
    for li, n in enumerate(g["sizes"]):
        n = int(n)
        xs = x[off:off + n].cpu().numpy()
        xo, lo, ro = orc(g["ref_%d" % li], g["pred_%d" % li], 5, g["las_%d" % li])
        assert np.abs(xs - xo).max() < 2e-5                                   # same arithmetic as the oracle, any size
        assert np.abs(xs - g["x_%d_e5" % li]).max() < (1e-5 if n <= 25 else 1e-3)   # the reference's own run
        assert abs(float(loss[li]) - float(g["loss_%d_e5" % li])) <= 1e-3 * float(g["loss_%d_e5" % li])
        assert abs(float(rmsd[li]) - float(g["rmsd_%d_e5" % li])) < 1e-4
        assert np.abs(xn[off:off + n].cpu().numpy() - g["x_%d_nolas" % li]).max() < (1e-5 if n <= 25 else 1e-3)
        assert abs(float(lossn[li]) - float(g["loss_%d_nolas" % li])) <= 1e-3 * float(g["loss_%d_nolas" % li])
        off += n


def test_post_optim_full_horizon_reaches_the_reference_level_and_single_ligand_api():
    from fabind_amd.utils.post_optim_utils import post_optimize_compound_coords, post_optimize_compound_coords_batched
    dev = torch.device("cuda:0")
    g = load_npz("post_optim")
    ref, pred, batch, las = _batch(g, dev)
    x, loss, rmsd = post_optimize_compound_coords_batched(ref, pred, batch, total_epoch=1000, LAS_edge_index=las)
    assert torch.isfinite(x).all()
    off = 0
    for li, n in enumerate(g["sizes"]):
        n = int(n)
        lr, rr = float(g["loss_%d_e1000" % li]), float(g["rmsd_%d_e1000" % li])
        assert abs(float(loss[li]) - lr) <= 0.03 * lr, (li, float(loss[li]), lr)   # chaotic iteration: same level, not same bits
        assert abs(float(rmsd[li]) - rr) < 0.08, (li, float(rmsd[li]), rr)
        assert float(loss[li]) < float(g["loss_%d_e5" % li])
        assert np.abs(x[off:off + n].cpu().numpy() - g["x_%d_e1000" % li]).max() < 0.6
        off += n
    # the reference's per-ligand signature; a batch of one gives the same bits as the same ligand inside a larger batch
    n0 = int(g["sizes"][0])
    x0, l0, r0 = post_optimize_compound_coords(ref[:n0], pred[:n0], total_epoch=1000, LAS_edge_index=las[:, las[0] < n0])
    assert torch.equal(x0, x[:n0]) and abs(l0 - float(loss[0])) < 1e-6 and isinstance(r0, float)
    with pytest.raises(RuntimeError):
        post_optimize_compound_coords(ref[:n0].cpu(), pred[:n0].cpu(), total_epoch=5)


def test_post_optim_restores_bond_geometry_at_batch_scale():
    """Size-independent property at production scale: 512 ligands of 20-150 atoms in one launch; the LAS distance error
    falls by an order of magnitude and no two atoms end closer than ~1.2 A."""
    from fabind_amd.utils.post_optim_utils import post_optimize_compound_coords_batched
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    refs, preds, las, batch, off = [], [], [], [], 0
    for li in range(512):
        n = int(rng.integers(20, 151))
        st = rng.normal(size=(n, 3)); st /= np.linalg.norm(st, axis=1, keepdims=True)
        r = np.cumsum(1.5 * st, 0).astype(np.float32)
        idx = np.arange(n)
        ii, jj = np.nonzero((np.abs(idx[:, None] - idx[None]) <= 2) & (idx[:, None] != idx[None]))
        refs.append(r); preds.append((r + rng.normal(scale=0.5, size=r.shape)).astype(np.float32))
        las.append(np.stack([ii, jj]) + off); batch += [li] * n; off += n
    ref, pred = torch.from_numpy(np.concatenate(refs)).to(dev), torch.from_numpy(np.concatenate(preds)).to(dev)
    e = torch.from_numpy(np.concatenate(las, 1)).to(dev)
    x, loss, rmsd = post_optimize_compound_coords_batched(ref, pred, torch.tensor(batch, device=dev), 1000, e)
    dev_err = lambda c: ((c[e[0]] - c[e[1]]).norm(dim=-1) - (ref[e[0]] - ref[e[1]]).norm(dim=-1)).abs().mean().item()
    before, after = dev_err(pred), dev_err(x)
    print("mean LAS distance error %.3f -> %.3f A" % (before, after))
    assert after < 0.15 * before and torch.isfinite(loss).all() and torch.isfinite(rmsd).all()


def test_post_optim_edge_inputs():
    """Single-atom ligand (zero gradient everywhere: Adam must not produce NaN), a ligand without any LAS edge in a batch that
    has some, and a ligand beyond the kernel's 512-atom limit (must raise, not corrupt memory)."""
    from fabind_amd.utils.post_optim_utils import post_optimize_compound_coords_batched
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    sizes = [1, 6, 9]                                   # ligand 0: one atom; ligand 1: no LAS edges; ligand 2: chain
    ref = torch.randn(sum(sizes), 3, generator=g) * 3.0
    pred = ref + 0.4 * torch.randn(sum(sizes), 3, generator=g)
    batch = torch.tensor(sum([[i] * n for i, n in enumerate(sizes)], []))
    o = 7
    idx = torch.arange(9)
    las = torch.stack([torch.cat([idx[:-1], idx[1:]]), torch.cat([idx[1:], idx[:-1]])]) + o
    x, loss, rmsd = post_optimize_compound_coords_batched(ref.to(dev), pred.to(dev), batch.to(dev), 200, las.to(dev))
    assert torch.isfinite(x).all() and torch.isfinite(loss).all() and torch.isfinite(rmsd).all()
    assert torch.equal(x[0].cpu(), pred[0])                                    # nothing acts on a lone atom
    assert abs(float(loss[0]) - 2.44) < 1e-5                                   # the reference's 2 * relu(1.22 - 0) self term
    d = lambda c: (c[las[0]] - c[las[1]]).norm(dim=-1)
    assert (d(x.cpu()) - d(ref)).abs().mean() < 0.5 * (d(pred) - d(ref)).abs().mean()   # the chain's bond lengths improve
    big = torch.randn(513, 3, generator=g)
    with pytest.raises(RuntimeError, match="512 atoms"):
        post_optimize_compound_coords_batched(big.to(dev), big.to(dev), torch.zeros(513, dtype=torch.long, device=dev), 5, None)
