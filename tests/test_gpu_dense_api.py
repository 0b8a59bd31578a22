"""GPU: the inner modules called with the REFERENCE'S OWN signatures (dense zero-padded tensors) against the CPU
oracle's restatement of the same functions (oracle/fabind_oracle.py: mha / row_attention / cross_attention / interaction /
att_forward / egnn_forward, pinned to the reference by test_oracle_golden.py), and the INTEGRATION.md alias recipe executed
verbatim.  Reference: FABind/fabind/models/egnn.py:308-333,392-466, cross_att.py:24-54,118-134, model_utils.py:96-159,200-223."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import fabind_oracle as orc
from test_gpu_stack import _args

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def _close(got, ref, tol=1e-4, what=""):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    err = float((got - ref).abs().max())
    assert err <= tol * max(1.0, float(ref.abs().max())), (what, err, float(ref.abs().max()))


def _dense_inputs(counts_p, counts_c, H, seed):
    g = torch.Generator().manual_seed(seed)
    B, Pm, Cm = len(counts_p), max(counts_p), max(counts_c)
    pm = torch.arange(Pm)[None, :] < torch.tensor(counts_p)[:, None]
    cm = torch.arange(Cm)[None, :] < torch.tensor(counts_c)[:, None]
    p = torch.randn(B, Pm, H, generator=g) * pm[..., None]
    c = torch.randn(B, Cm, H, generator=g) * cm[..., None]
    zm = pm[:, :, None] & cm[:, None, :]
    z = torch.randn(B, Pm, Cm, H, generator=g) * zm[..., None]
    return p, pm, c, cm, z, zm


@pytest.fixture(autouse=True)
def _fp32():
    from fabind_amd import engine
    engine.set_precision("fp32")
    yield
    engine.set_precision("fp32")


@pytest.mark.parametrize("Q,Kn", [(37, 9), (1501, 41), (41, 1501)])
def test_attention_forward_reference_signature(Q, Kn):
    """Attention.forward(q_x, kv_x, biases) vs oracle.mha, incl. the headline block shapes (1501 x 41 and 41 x 1501)."""
    from fabind_amd.models.model_utils import Attention
    torch.manual_seed(0)
    H, B = 64, 2
    att = Attention(H, H, H, 32, 4).eval()
    g = torch.Generator().manual_seed(1)
    q_x, kv_x = torch.randn(B, Q, H, generator=g), torch.randn(B, Kn, H, generator=g)
    kmask = torch.ones(B, Q, Kn, dtype=torch.bool)
    kmask[1, :, Kn - 3:] = False
    mask_bias = (1e9 * (kmask.float() - 1))[:, None]
    pair_bias = torch.randn(B, 4, Q, Kn, generator=g)
    ref = orc.mha(_sd(att), "", q_x, kv_x, [mask_bias, pair_bias])
    got = att.to(DEV)(q_x.to(DEV), kv_x.to(DEV), [mask_bias.to(DEV), pair_bias.to(DEV)])
    assert got.shape == ref.shape
    _close(got, ref, 1e-4, "Attention")


def test_row_attention_block_reference_signature():
    from fabind_amd.models.cross_att import RowAttentionBlock
    torch.manual_seed(2)
    H = 64
    blk = RowAttentionBlock(H, H, rm_layernorm=True).eval()
    p, pm, c, cm, z, zm = _dense_inputs([50, 33, 41], [9, 12, 5], H, 3)
    ref = orc.row_attention(_sd(blk), "", p, c, z, zm, pm)
    got = blk.to(DEV)(p.to(DEV), c.to(DEV), z.to(DEV), zm.to(DEV), pm.to(DEV))
    _close(got, ref, 1e-4, "RowAttentionBlock p<-c")
    # the transposed use (ligand queries, cross_att.py:43-47)
    ref = orc.row_attention(_sd(blk), "", c, p, z.transpose(1, 2), zm.transpose(1, 2), cm)
    got = blk(c.to(DEV), p.to(DEV), z.transpose(1, 2).to(DEV), zm.transpose(1, 2).to(DEV), cm.to(DEV))
    _close(got, ref, 1e-4, "RowAttentionBlock c<-p")


def test_interaction_module_reference_signature():
    from fabind_amd.models.model_utils import InteractionModule
    torch.manual_seed(4)
    H = 64
    for hidden in (32, 64):
        im = InteractionModule(H, H, hidden, rm_layernorm=True).eval()
        p, pm, c, cm, _, _ = _dense_inputs([21, 30], [7, 4], H, 5)
        ref, rmask = orc.interaction(_sd(im), "", p, c, pm, cm)
        got, gmask = im.to(DEV)(p.to(DEV), c.to(DEV), pm.to(DEV), cm.to(DEV))
        assert torch.equal(gmask.cpu(), rmask)
        _close(got, ref, 1e-4, "InteractionModule")


def test_cross_attention_module_reference_signature_and_gradients():
    from fabind_amd.models.cross_att import CrossAttentionModule
    torch.manual_seed(6)
    H = 64
    cam = CrossAttentionModule(H, H, rm_layernorm=True).eval()
    p, pm, c, cm, z, zm = _dense_inputs([44, 29, 37], [8, 11, 6], H, 7)
    sd = _sd(cam)
    for v in sd.values():
        v.requires_grad_(True)
    pr, cr, zr = orc.cross_attention(sd, "", p, pm, c, cm, z, zm)
    cam = cam.to(DEV)
    pd, cd, zd = p.to(DEV).requires_grad_(True), c.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
    pg, cg, zg = cam(pd, pm.to(DEV), cd, cm.to(DEV), zd, zm.to(DEV))
    _close(pg[pm], pr[pm], 1e-4, "p'")
    _close(cg[cm], cr[cm], 1e-4, "c'")
    _close(zg, zr, 1e-4, "pair'")                   # padded pairs are zero on both sides
    g = torch.Generator().manual_seed(8)
    cp, cc, cz = torch.randn(pr.shape, generator=g) * pm[..., None], torch.randn(cr.shape, generator=g) * cm[..., None], \
        torch.randn(zr.shape, generator=g)
    ((pr * cp).sum() + (cr * cc).sum() + (zr * cz).sum()).backward()
    ((pg * cp.to(DEV)).sum() + (cg * cc.to(DEV)).sum() + (zg * cz.to(DEV)).sum()).backward()
    for n, prm in cam.named_parameters():
        r = sd[n].grad
        assert prm.grad is not None, n
        assert float((prm.grad.cpu() - r).abs().max()) <= 3e-3 * float(r.abs().max()) + 1e-6, n


def _stack_case(seed, H=64, L=2):
    from fabind_amd import synthetic
    from fabind_amd.models.att_model import EfficientMCAttModel
    torch.manual_seed(seed)
    m = EfficientMCAttModel(_args(H, L, 1), H, H, 1, n_layers=L, n_iter=1, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("coord_mlp.2.weight"):
                p.mul_(300.0)
    inp = synthetic.make_stack_batch([(60, 12), (45, 9), (70, 15)], H, seed=seed)
    lay = orc.Layout(inp["batch_id"], inp["segment_id"])
    sd = _sd(m)
    p, pm, c, cm = lay.dense(inp["H"])
    z0, zm = orc.interaction(sd, "inter_layer.", p, c, pm, cm)
    ctx, inter = orc.construct_edges(inp["X"], inp["batch_id"], inp["segment_id"], inp["is_global"], 2.0, 1.6)
    ctx = torch.cat([inp["compound_edge_index"], ctx], 1)
    return m, sd, inp, lay, z0, zm, ctx, inter


def test_mc_att_l_forward_reference_signature():
    """MC_Att_L.forward(h, edge_index, coord, segment_id=, batch_id=, reduced_tuple=, pair_embed_batched=, pair_mask=) with a
    GENERIC dense pair tensor (not the factored z0 the stack uses) vs oracle.att_forward."""
    m, sd, inp, lay, z0, zm, ctx, inter = _stack_case(21)
    g = torch.Generator().manual_seed(22)
    z = (z0 + 0.3 * torch.randn(z0.shape, generator=g)) * zm[..., None]
    h = torch.randn(inp["X"].shape[0], 64, generator=g) * 0.3
    x = inp["X"]
    B = lay.B
    hr, xr, ar = orc.att_forward(sd, "gnn.att_0.", h, inter[0], inter[1], x[:, 0], lay, z, zm, inp["batch_id"], B, 2.0)
    att = m.gnn.att_0.to(DEV)
    hg, xg, ag = att(h.to(DEV), inter.to(DEV), x.to(DEV), segment_id=inp["segment_id"].to(DEV), batch_id=inp["batch_id"].to(DEV),
                     reduced_tuple=None, pair_embed_batched=z.to(DEV), pair_mask=zm.to(DEV))
    assert xg.shape == x.shape
    _close(hg, hr, 1e-4, "h'")
    _close(xg[:, 0], xr, 1e-5, "coord'")
    _close(ag, ar, 1e-4, "attention weights")


def test_mcattegnn_forward_positional_reference_signature():
    """MCAttEGNN.forward(h, x, ctx_edges, att_edges, LAS_edge_list, batched_complex_coord_LAS, segment_id=, batch_id=, ...,
    pair_embed_batched=, pair_mask=, return_attention=True) vs oracle.egnn_forward."""
    m, sd, inp, lay, z0, zm, ctx, inter = _stack_case(31)
    hr, xr = orc.egnn_forward(sd, "gnn.", inp["H"], inp["X"][:, 0], ctx, inter, inp["LAS_edge_index"], inp["coord_LAS"][:, 0],
                              lay, z0 * zm[..., None], zm, inp["batch_id"], 2, 5.0)
    gnn = m.gnn.to(DEV)
    t = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp.items()}
    las_in = t["coord_LAS"].clone()
    hg, xg, atts = gnn(t["H"], t["X"], ctx.to(DEV), inter.to(DEV), t["LAS_edge_index"], las_in, segment_id=t["segment_id"],
                       batch_id=t["batch_id"], reduced_tuple=None, pair_embed_batched=z0.to(DEV), pair_mask=zm.to(DEV),
                       mask=t["mask"], return_attention=True)
    assert las_in.dim() == 2                                   # squeezed in place like the reference (egnn.py:435)
    assert len(atts) == 2 and atts[0].shape[0] == inter.shape[1]
    lig = inp["mask"].numpy()
    rm = float(np.sqrt((((xg[:, 0].detach().cpu().numpy() - xr.numpy())[lig] * 5) ** 2).sum(-1).mean()))
    assert rm < 1e-4, rm
    _close(hg, hr, 1e-4, "h_out")
    # the same module through the stack's own route gives the same answer
    ctxo = m.to(DEV).context(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"],
                             t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
    with torch.no_grad():
        h2, x2 = m.gnn(t["H"], t["X"], ctx=ctxo)
    _close(h2, hg, 1e-4, "ctx route vs positional route")


def test_integration_recipe_runs_verbatim():
    """alias `models.*` to fabind_amd.models, `from models.model import *`, strict load of a
    state_dict captured from the reference, `model.inference` -- executed in a fresh interpreter from the text of the file."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_integration_recipe.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "integration recipe ok" in out.stdout, out.stdout[-1500:]
