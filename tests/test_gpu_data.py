"""GPU: the device batch builder (fabind_amd/data.py) against the reference's per-sample HeteroData builder
(utils/utils.py:202-442, run by oracle/make_golden.py `data`), collated the way PyG's DataLoader collates."""
import numpy as np
import pytest
import torch

from helpers import load_npz

pytestmark = pytest.mark.gpu

NODE_COUNT_FIELD = {"complex": "node_coords", "complex_whole_protein": "node_coords", "compound": "node_feats",
                    "pocket": "node_feats", "protein_whole": "node_feats"}


def _collate(g):
    """PyG `Batch.from_data_list` semantics on the fixture's per-sample fields: node tensors concatenated, `edge_index`
    shifted by the running node count of its node type, graph-level tensors concatenated along dim 0."""
    n = int(g["n"])
    per = [{k.split("_", 1)[1]: v for k, v in g.items() if k.startswith("out%d_" % i)} for i in range(n)]
    out, running = {}, {}
    for i in range(n):
        counts = {st: per[i]["%s::%s" % (st, f)].shape[0] for st, f in NODE_COUNT_FIELD.items()}
        for key, v in per[i].items():
            store, field = key.split("::")
            if field == "edge_index":
                v = v + running.get(store.split("|")[0], 0)
                out.setdefault(key, []).append(v)
            else:
                out.setdefault(key, []).append(v)
        for st, c in counts.items():
            running[st] = running.get(st, 0) + c
            out.setdefault(st + "::batch", []).append(np.full(c, i, dtype=np.int64))
    return {k: np.concatenate(v, 1 if k.endswith("edge_index") else 0) for k, v in out.items()}


def _raw(g):
    return [dict(protein_node_xyz=g["raw%d_protein_node_xyz" % i], protein_esm2_feat=g["raw%d_protein_esm2_feat" % i],
                 coords=g["raw%d_coords" % i], compound_node_features=g["raw%d_compound_node_features" % i],
                 input_atom_edge_list=g["raw%d_input_atom_edge_list" % i], LAS_edge_index=g["raw%d_LAS_edge_index" % i],
                 rdkit_coords=g["raw%d_rdkit_coords" % i], pdb="x%d" % i) for i in range(int(g["n"]))]


def test_device_batch_builder_matches_reference_builder():
    from fabind_amd.data import build_batch
    dev = torch.device("cuda:0")
    g = load_npz("data_builder")
    ref = _collate(g)
    batch = build_batch(_raw(g), dev)
    checked = 0
    for key, want in ref.items():
        store, field = key.split("::")
        if store == "":
            got = getattr(batch, field)
        else:
            got = batch[tuple(store.split("|")) if "|" in store else store][field]
        got = got.cpu().numpy()
        assert got.shape == want.shape, (key, got.shape, want.shape)
        if want.dtype.kind == "f":
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), key
        else:
            assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), key
        checked += 1
    assert checked >= 35
    assert int(batch["pocket"].keepNode.sum()) == sum(int(g["out%d_pocket::keepNode" % i].sum()) for i in range(4))
    assert batch.pdb == ["x0", "x1", "x2", "x3"]
    with pytest.raises(RuntimeError):
        build_batch(_raw(g), "cpu")


def test_feeder_preserves_order_and_content():
    """The double-buffered feeder (host pack thread + side stream) yields the same batches, in order, as direct calls."""
    from fabind_amd.data import DeviceFeeder, build_batch
    dev = torch.device("cuda:0")
    g = load_npz("data_builder")
    raw = _raw(g)
    groups = [raw[:2], raw[2:], raw[1:3]]
    got = list(DeviceFeeder(groups, dev))
    assert len(got) == 3
    for b, grp in zip(got, groups):
        direct = build_batch(grp, dev)
        assert b.pdb == [s["pdb"] for s in grp]
        # segment means use float atomics (index_add_): equal to rounding, not bit for bit
        assert torch.allclose(b["complex"].node_coords, direct["complex"].node_coords, atol=1e-5)
        assert torch.allclose(b.dis_map, direct.dis_map, atol=1e-5)
        assert torch.equal(b["complex", "LAS", "complex"].edge_index, direct["complex", "LAS", "complex"].edge_index)


def test_built_batch_feeds_the_model():
    """A batch assembled on the device from raw arrays (production feature widths) runs through IaBNet forward + loss."""
    from fabind_amd import engine
    from fabind_amd.data import build_batch
    from fabind_amd.models import get_model
    from fabind_amd.models.model import compute_loss
    from test_gpu_stack import _args
    dev = torch.device("cuda:0")
    engine.set_precision("fp32")
    g = torch.Generator().manual_seed(7)
    raw = []
    for L, nc in ((90, 9), (120, 14), (70, 6)):
        prot = torch.randn(L, 3, generator=g) * 9.0
        walk = torch.cumsum(1.5 * torch.nn.functional.normalize(torch.randn(nc, 3, generator=g), dim=1), 0)
        idx = torch.arange(nc)
        bonds = torch.stack([torch.cat([idx[:-1], idx[1:]]), torch.cat([idx[1:], idx[:-1]])], 1)
        m = ((idx[:, None] - idx[None]).abs() <= 2) & (idx[:, None] != idx[None])
        raw.append(dict(protein_node_xyz=prot, protein_esm2_feat=0.1 * torch.randn(L, 1280, generator=g),
                        coords=walk - walk.mean(0) + torch.tensor([4.0, 0.0, 0.0]),
                        compound_node_features=0.1 * torch.randn(nc, 56, generator=g), input_atom_edge_list=bonds,
                        LAS_edge_index=torch.nonzero(m).t(), rdkit_coords=walk - walk.mean(0)))
    batch = build_batch(raw, dev)

    class _L:
        def log_message(self, m):
            pass
    a = _args(32, 1, 1)
    a.pocket_pred_hidden_size = 32
    torch.manual_seed(0)
    model = get_model(a, _L(), dev).to(dev)
    out = model(batch, stage=1, train=False)
    loss, terms = compute_loss(out, batch)
    assert out[0].shape == batch.coords.shape and torch.isfinite(out[0]).all() and torch.isfinite(loss)
    coords, _ = model.inference(batch)
    assert torch.isfinite(coords).all()


def test_batch_builder_edge_inputs():
    """Batch of one; a ligand without bonds / LAS edges next to a normal one (empty edge lists, zero-length integer slices);
    a single-atom ligand."""
    from fabind_amd.data import build_batch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)

    def sample(L, nc, bonds=True):
        prot = torch.randn(L, 3, generator=g) * 8.0
        lig = torch.randn(nc, 3, generator=g) + torch.tensor([3.0, 0.0, 0.0])
        idx = torch.arange(nc)
        e = torch.stack([torch.cat([idx[:-1], idx[1:]]), torch.cat([idx[1:], idx[:-1]])], 1) if (bonds and nc > 1) else torch.zeros(0, 2, dtype=torch.long)
        return dict(protein_node_xyz=prot, protein_esm2_feat=torch.randn(L, 16, generator=g), coords=lig,
                    compound_node_features=torch.randn(nc, 8, generator=g), input_atom_edge_list=e,
                    LAS_edge_index=e.t().contiguous(), rdkit_coords=lig - lig.mean(0))
    one = build_batch([sample(40, 7)], dev)
    assert int(one["compound"].batch.max()) == 0 and one["complex"].node_coords.shape[0] == 7 + int(one["pocket"].keepNode.sum()) + 2
    mixed = build_batch([sample(30, 5, bonds=False), sample(35, 6), sample(25, 1)], dev)
    assert mixed["complex", "c2c", "complex"].edge_index.shape == (2, 10)          # only the middle ligand has bonds (5 x 2)
    assert mixed["compound_atom_edge_list"].batch.tolist() == [1] * 10
    n = torch.bincount(mixed["complex"].batch).tolist()
    off1 = n[0]
    e = mixed["complex", "c2c", "complex"].edge_index
    assert int(e.min()) == off1 + 1 and int(e.max()) == off1 + 6                   # shifted into the second complex, after its glb_c
    assert mixed.dis_map.shape[0] == sum(int(k) * c for k, c in zip(torch.bincount(mixed["pocket"].batch, minlength=3).tolist(), (5, 6, 1)))
    assert torch.isfinite(mixed["complex"].node_coords).all() and torch.isfinite(mixed.dis_map).all()
    empty = build_batch([sample(20, 4, bonds=False)], dev)                          # no edge of any kind in the whole batch
    assert empty["complex", "c2c", "complex"].edge_index.shape == (2, 0) and empty["LAS_edge_list"].x.shape == (0, 2)
