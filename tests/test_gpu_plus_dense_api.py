"""GPU: the FABind+ inner modules called with the REFERENCE'S OWN signatures (dense zero-padded tensors) against the CPU
oracle's restatement of the same functions (oracle/fabind_plus_oracle.py: ln_mlp / row_attention / cross_attention /
gcl_forward / att_forward / egnn_forward, pinned to the reference tree by test_oracle_golden.py).
Reference: FABind_plus/fabind/models/model_utils.py:10-98,150-320, cross_att.py:20-89, egnn.py:100-118,277-300,358-433."""
import numpy as np
import pytest
import torch

import fabind_oracle as orc
import fabind_plus_oracle as porc
from test_gpu_dense_api import _close, _dense_inputs, _sd
from test_gpu_plus import _args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _fp32():
    from fabind_amd import engine
    engine.set_precision("fp32")
    yield
    engine.set_precision("fp32")


def test_plus_ln_mlps_reference_signature():
    """MLP / MLPwithLastAct / MLPwoBias / MLP4Confidence .forward(z) on [B, n, C] and [B, P, C, C'] shaped inputs."""
    from fabind_amd.plus.models.model_utils import MLP, MLP4Confidence, MLPwithLastAct, MLPwoBias
    torch.manual_seed(0)
    a = _args(64, 1, 1)
    a.confidence_use_ln_mlp, a.confidence_dropout = True, 0.1
    g = torch.Generator().manual_seed(1)
    for cls, last, n, out in ((MLP, False, 2, 48), (MLPwithLastAct, True, 1, 64), (MLPwoBias, False, 2, 1),
                              (MLP4Confidence, False, 2, 1)):
        m = cls(a, embedding_channels=64, out_channels=out, n=n).eval()
        for shape in ((3, 17, 64), (2, 9, 5, 64)):
            z = torch.randn(shape, generator=g)
            ref = porc.ln_mlp(_sd(m), "", z, last)
            got = m.to(DEV)(z.to(DEV))
            assert got.shape == ref.shape
            _close(got, ref, 1e-4, cls.__name__)
            m = m.cpu()


def test_plus_attention_primitives_reference_signature():
    """Attention / RowAttentionBlock (mha_permu on and off) / InteractionModule with the FABind+ constructors."""
    from fabind_amd.plus.models.cross_att import RowAttentionBlock
    from fabind_amd.plus.models.model_utils import Attention, InteractionModule
    torch.manual_seed(2)
    a, H = _args(64, 1, 1), 64
    att = Attention(a, H, H, H, 32, 4, mha_permu=True).eval()
    g = torch.Generator().manual_seed(3)
    q_x, kv_x = torch.randn(2, 37, H, generator=g), torch.randn(2, 9, H, generator=g)
    pair_bias = torch.randn(2, 4, 37, 9, generator=g)
    _close(att.to(DEV)(q_x.to(DEV), kv_x.to(DEV), [pair_bias.to(DEV)]), orc.mha(_sd(att), "", q_x, kv_x, [pair_bias]), 1e-4, "Attention")
    p, pm, c, cm, z, zm = _dense_inputs([50, 33, 41], [9, 12, 5], H, 4)
    for permu in (True, False):
        blk = RowAttentionBlock(a, H, H, no_heads=4, rm_layernorm=True, mha_permu=permu).eval()
        ref = porc.row_attention(_sd(blk), "", p, c, z, zm, pm, 4)
        got = blk.to(DEV)(p.to(DEV), c.to(DEV), z.to(DEV), zm.to(DEV), pm.to(DEV))
        _close(got, ref, 1e-4, "RowAttentionBlock")
    im = InteractionModule(H, H, 32, rm_layernorm=True).eval()
    ref, rmask = orc.interaction(_sd(im), "", p, c, pm, cm)
    got, gmask = im.to(DEV)(p.to(DEV), c.to(DEV), pm.to(DEV), cm.to(DEV))
    assert torch.equal(gmask.cpu(), rmask)
    _close(got, ref, 1e-4, "InteractionModule")


def test_plus_cross_attention_module_reference_signature_and_gradients():
    """CrossAttentionModule.forward -> (p', c', UPDATED pair embedding); parameter gradients vs oracle autograd."""
    from fabind_amd.plus.models.cross_att import CrossAttentionModule
    torch.manual_seed(6)
    H = 64
    cam = CrossAttentionModule(_args(H, 1, 1), H, H, rm_layernorm=True).eval()
    p, pm, c, cm, z, zm = _dense_inputs([44, 29, 37], [8, 11, 6], H, 7)
    sd = _sd(cam)
    for v in sd.values():
        v.requires_grad_(True)
    pr, cr, zr = porc.cross_attention(sd, "", p, pm, c, cm, z, zm)
    cam = cam.to(DEV)
    pg, cg, zg = cam(p.to(DEV), pm.to(DEV), c.to(DEV), cm.to(DEV), z.to(DEV), zm.to(DEV))
    _close(pg[pm], pr[pm], 1e-4, "p'")
    _close(cg[cm], cr[cm], 1e-4, "c'")
    _close(zg, zr, 1e-4, "pair'")
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
    from fabind_amd.plus.models.att_model import EfficientMCAttModel
    torch.manual_seed(seed)
    m = EfficientMCAttModel(_args(H, L, 1), H, H, 1, n_layers=L, n_iter=1, normalize_coord=lambda x: x / 5.0,
                            unnormalize_coord=lambda x: x * 5.0).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("coord_mlp.linear2.weight"):
                p.mul_(300.0)                                   # xavier gain 0.001 would hide coordinate errors
    inp = synthetic.make_stack_batch([(60, 12), (45, 9), (70, 15)], H, seed=seed)
    lay = orc.Layout(inp["batch_id"], inp["segment_id"])
    sd = _sd(m)
    p, pm, c, cm = lay.dense(inp["H"])
    z0, zm = orc.interaction(sd, "inter_layer.", p, c, pm, cm)
    ctx, inter = orc.construct_edges(inp["X"], inp["batch_id"], inp["segment_id"], inp["is_global"], 2.0, 1.6)
    ctx = torch.cat([inp["compound_edge_index"], ctx], 1)
    return m, sd, inp, lay, z0, zm, ctx, inter


def test_plus_mc_e_gcl_forward_reference_signature():
    """MC_E_GCL.forward(h, edge_index, coord, batch_id=) on an edge list in arbitrary order vs oracle.gcl_forward."""
    m, sd, inp, lay, z0, zm, ctx, inter = _stack_case(11)
    g = torch.Generator().manual_seed(12)
    ctx = ctx[:, torch.randperm(ctx.shape[1], generator=g)]
    h = torch.randn(inp["X"].shape[0], 64, generator=g) * 0.5
    hr, xr = porc.gcl_forward(sd, "gnn.gcl_0.", h, ctx[0], ctx[1], inp["X"][:, 0], inp["batch_id"], lay.B, 2.0)
    gcl = m.gnn.gcl_0.to(DEV)
    hg, xg = gcl(h.to(DEV), ctx.to(DEV), inp["X"].to(DEV), batch_id=inp["batch_id"].to(DEV))
    assert xg.shape == inp["X"].shape
    _close(hg, hr, 1e-4, "h'")
    _close(xg[:, 0], xr, 1e-5, "coord'")


def test_plus_mc_att_l_forward_reference_signature():
    """MC_Att_L.forward(h, edge_index, coord, segment_id=, batch_id=, reduced_tuple=, pair_embed_batched=, pair_mask=) with a
    generic dense pair tensor vs oracle.att_forward: four outputs, the last the updated pair embedding."""
    m, sd, inp, lay, z0, zm, ctx, inter = _stack_case(21)
    g = torch.Generator().manual_seed(22)
    z = (z0 + 0.3 * torch.randn(z0.shape, generator=g)) * zm[..., None]
    h = torch.randn(inp["X"].shape[0], 64, generator=g) * 0.3
    x = inp["X"]
    hr, xr, ar, zr = porc.att_forward(sd, "gnn.att_0.", h, inter[0], inter[1], x[:, 0], lay, z, zm, inp["batch_id"], lay.B, 2.0)
    att = m.gnn.att_0.to(DEV)
    hg, xg, ag, zg = att(h.to(DEV), inter.to(DEV), x.to(DEV), segment_id=inp["segment_id"].to(DEV),
                         batch_id=inp["batch_id"].to(DEV), reduced_tuple=None, pair_embed_batched=z.to(DEV),
                         pair_mask=zm.to(DEV))
    assert xg.shape == x.shape and zg.shape == z.shape
    _close(hg, hr, 1e-4, "h'")
    _close(xg[:, 0], xr, 1e-5, "coord'")
    _close(ag, ar, 1e-4, "attention weights")
    _close(zg, zr, 1e-4, "pair'")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_plus_mcattegnn_forward_positional_reference_signature(precision):
    """MCAttEGNN.forward(h, x, ctx_edges, att_edges, LAS_edge_list, batched_complex_coord_LAS, segment_id=, batch_id=, ...,
    pair_embed_batched=, pair_mask=, return_attention=True) vs oracle.egnn_forward (fp32: the 1e-4 A gate; bf16 -- which
    takes the fused pair-update kernel -- at the bf16 stack tolerance)."""
    from fabind_amd import engine
    m, sd, inp, lay, z0, zm, ctx, inter = _stack_case(31)
    hr, xr, zr = porc.egnn_forward(sd, "gnn.", inp["H"], inp["X"][:, 0], ctx, inter, inp["LAS_edge_index"],
                                   inp["coord_LAS"][:, 0], lay, z0 * zm[..., None], zm, inp["batch_id"], 2, 5.0)
    engine.set_precision(precision)
    gnn = m.gnn.to(DEV)
    t = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp.items()}
    las_in = t["coord_LAS"].clone()
    with torch.no_grad():
        hg, xg, atts, zg = gnn(t["H"], t["X"], ctx.to(DEV), inter.to(DEV), t["LAS_edge_index"], las_in,
                               segment_id=t["segment_id"], batch_id=t["batch_id"], reduced_tuple=None,
                               pair_embed_batched=z0.to(DEV), pair_mask=zm.to(DEV), mask=t["mask"], return_attention=True)
        out3 = gnn(t["H"], t["X"], ctx.to(DEV), inter.to(DEV), t["LAS_edge_index"], t["coord_LAS"].clone(),
                   segment_id=t["segment_id"], batch_id=t["batch_id"], pair_embed_batched=z0.to(DEV), pair_mask=zm.to(DEV))
    assert las_in.dim() == 2                                   # squeezed in place like the reference (egnn.py:397)
    assert len(out3) == 3 and len(atts) == 2 and atts[0].shape[0] == inter.shape[1]
    lig = inp["mask"].numpy()
    rm = float(np.sqrt((((xg[:, 0].cpu().numpy() - xr.numpy())[lig] * 5) ** 2).sum(-1).mean()))
    tol_x, tol = (1e-4, 1e-4) if precision == "fp32" else (5e-2, 5e-2)
    assert rm < tol_x, rm
    _close(hg, hr, tol, "h_out")
    _close(zg, zr, tol, "pair embedding")
