"""Reference-signature forwards of the FABind+ inner modules: dense, zero-padded tensors in and out, the arithmetic on
the HIP kernels (the FABind+ counterpart of fabind_amd/dense.py).

The FABind+ `MLP*` containers, `CrossAttentionModule`, `MC_E_GCL`, `MC_Att_L` and the positional `MCAttEGNN.forward`
(FABind_plus/fabind/models/model_utils.py:10-98, cross_att.py:20-47, egnn.py:100-118, 277-300, 358-433) exchange
`to_dense_batch`-padded node tensors and a dense `[B, P, C, H]` pair tensor that every attention layer UPDATES and
returns.  The stack (fabind_amd/plus/engine.py) keeps that tensor as a ragged pair list; this module packs a caller's
dense tensors into the list, runs the stack's own layer functions (`engine.cross_part`, `engine.att_layer`,
`engine.gcl_layer`) and scatters the results back into zero-padded tensors of the reference's shapes.
`Attention`, `RowAttentionBlock` and `InteractionModule` compute the same function as in FABind v1 when
`--rel-dis-pair-bias no` (the only built configuration: `mha_permu` matters to the distance term alone,
model_utils.py:109-145), so they share v1's adapters.

No CPU path: every function raises on non-HIP tensors."""
import torch

from .. import dense as v1
from .. import engine as eng1
from .. import ops
from . import engine
from .engine import PairList


def mlp(m, z, last_act):
    """MLP / MLPwithLastAct / MLPwoBias / MLP4Confidence.forward on [..., C] (model_utils.py:10-98)."""
    v1._require_cuda(z, type(m).__name__)
    if isinstance(getattr(m, "dropout", None), torch.nn.Dropout):
        p = m.dropout.p
    elif isinstance(getattr(m, "dropout1", None), torch.nn.Dropout):
        p = m.dropout1.p
    else:
        p = 0.0
    rows = z.reshape(-1, z.shape[-1]).float().contiguous()
    out = engine.mlp_module(m, rows, last_act=last_act, pdrop=p if m.training else 0.0)
    return out.reshape(z.shape[:-1] + (out.shape[-1],))


def _pd(mod):
    return mod.p if mod.training else 0.0


class _Packed:
    """Ragged view of one (p, c, z) triple of dense tensors: Layout, PairList, node-layout rows, pair list."""

    def __init__(self, p_embed, p_mask, c_embed, c_mask, pair_embed, pair_mask, who):
        p_mask, c_mask, pair_mask = p_mask.bool(), c_mask.bool(), pair_mask.bool()
        self.lay = lay = v1.pair_layout(p_mask, c_mask, who)
        if int(pair_mask.sum()) != lay.n_pairs:
            raise ValueError("fabind_amd: %s expects pair_mask = p_mask x c_mask" % who)
        self.pairs = PairList(lay, p_mask.device)
        self.p_mask, self.c_mask, self.pair_mask = p_mask, c_mask, pair_mask
        self.h = v1._to_nodes(lay, p_embed[p_mask], c_embed[c_mask])
        self.z = pair_embed[pair_mask].float().contiguous()


def cross_attention(cam_owner, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask, P=None):
    """CrossAttentionModule.forward (cross_att.py:20-47) -> (p', c', z').  `cam_owner` is the MC_Att_L holding the module
    (its parameter pack covers the module) or the module itself."""
    v1._require_cuda(p_embed_batched, "CrossAttentionModule")
    cam = getattr(cam_owner, "cross_attn_module", cam_owner)
    P = P or (engine.att_params(cam_owner) if cam_owner is not cam else engine.cam_params(cam))
    st = _Packed(p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask, "CrossAttentionModule")
    lay = st.lay
    h, z, _ = engine.cross_part(P, st.h, st.z, lay, st.pairs, _pd(cam.p_attention_block.dropout))
    p_out = torch.zeros_like(p_embed_batched, dtype=torch.float32)
    c_out = torch.zeros_like(c_embed_batched, dtype=torch.float32)
    p_out[st.p_mask] = h.index_select(0, lay.p_index64).float()
    c_out[st.c_mask] = h.index_select(0, lay.c_index64).float()
    # padded node rows: the reference leaves transition(0)-bias terms there, masked out by every consumer; zeros here
    z_out = torch.zeros(st.pair_mask.shape + (z.shape[1],), dtype=torch.float32, device=z.device)
    z_out[st.pair_mask] = z.float()
    return p_out, c_out, z_out


def gcl_forward(m, h, edge_index, coord, batch_id):
    """MC_E_GCL.forward (egnn.py:100-118): h [N, H], edge_index [2, E] (any order), coord [N, 1, 3], batch_id [N]
    (complex-contiguous) -> (h', coord')."""
    v1._require_cuda(h, "MC_E_GCL")
    n = h.shape[0]
    g = eng1.EdgeListGraph(edge_index, n)
    lay = eng1.BatchOnlyLayout(batch_id)
    h2, x2 = engine.gcl_layer(engine.gcl_params(m), h.float().contiguous(), coord.reshape(n, 3).float().contiguous(), lay, g,
                              float(m.coord_change_maximum), m.args.dropout if m.training else 0.0)
    return h2, x2.reshape(coord.shape)


def _z_list(pair_embed_batched, pair_mask, lay, who):
    pm = pair_mask.bool()
    if int(pm.sum()) != lay.n_pairs:
        raise ValueError("fabind_amd: %s expects pair_mask = to_dense_batch masks of the two segments" % who)
    return pair_embed_batched[pm].to(ops.act_dtype()).contiguous(), pm


def _z_dense(z, pm):
    out = torch.zeros(pm.shape + (z.shape[1],), dtype=torch.float32, device=z.device)
    out[pm] = z.float()
    return out


def att_layer(m, h, edge_index, coord, segment_id, batch_id, pair_embed_batched, pair_mask):
    """MC_Att_L.forward (egnn.py:277-300) with the reference's arguments -> (h', coord', alpha [E] in the caller's edge
    order, updated dense pair embedding).  `reduced_tuple` is recomputed from the edge list."""
    v1._require_cuda(h, "MC_Att_L")
    lay = eng1.Layout(batch_id, segment_id)
    g = v1.EdgeListInterGraph(edge_index, lay)
    pairs = PairList(lay, h.device)
    z, pm = _z_list(pair_embed_batched, pair_mask, lay, "MC_Att_L")
    x = coord.reshape(-1, 3).float().contiguous()
    h2, x2, alpha, z2, _ = engine.att_layer(engine.att_params(m), h.float().contiguous(), x, z, lay, g, pairs, batch_id,
                                            float(m.coord_change_maximum), _pd(m.dropout))
    alpha_out = torch.empty_like(alpha)
    alpha_out[g.order] = alpha
    return h2, x2.reshape(coord.shape), alpha_out, _z_dense(z2, pm)


def egnn_forward(gnn, h, x, ctx_edges, att_edges, LAS_edge_list, batched_complex_coord_LAS, segment_id=None, batch_id=None,
                 reduced_tuple=None, pair_embed_batched=None, pair_mask=None, LAS_mask=None, p_p_dist_embed=None,
                 c_c_dist_embed=None, mask=None, ctx_edge_attr=None, att_edge_attr=None, return_attention=False):
    """MCAttEGNN.forward with the reference's positional arguments (egnn.py:358-433) -> (h, x[, atts], pair embedding)."""
    v1._require_cuda(h, "MCAttEGNN")
    assert ctx_edge_attr is None and att_edge_attr is None
    lay = eng1.Layout(batch_id, segment_id)
    pairs = PairList(lay, h.device)
    g_int = v1.EdgeListInterGraph(att_edges, lay)
    g_ctx = eng1.EdgeListGraph(ctx_edges, lay.N)
    z, pm = _z_list(pair_embed_batched, pair_mask, lay, "MCAttEGNN")
    wd = ops.mm_dtype()
    pd = _pd(gnn.dropout)
    h = engine._drop(ops.linear(h.float().contiguous(), gnn.linear_in.weight.to(wd).contiguous(), gnn.linear_in.bias), pd)
    shp = x.shape
    xx = x.reshape(-1, 3).float().contiguous()
    if batched_complex_coord_LAS.dim() == 3:
        batched_complex_coord_LAS.squeeze_(1)                   # the reference squeezes the caller's tensor in place (egnn.py:397)
    x_las = batched_complex_coord_LAS.reshape(-1, 3).float().contiguous()
    las = (LAS_edge_list[0].to(torch.int32).contiguous(), LAS_edge_list[1].to(torch.int32).contiguous())
    las = las + (lay.ranges(las[0]),)
    scale = 10.0 / float(gnn.gcl_0.coord_change_maximum)        # coord_change_maximum = normalize_coord(10)
    atts, bias = [], None
    packs = [engine.att_params(getattr(gnn, "att_%d" % i)) for i in range(gnn.n_layers)]
    for i in range(gnn.n_layers):
        gcl, att = getattr(gnn, "gcl_%d" % i), getattr(gnn, "att_%d" % i)
        h, xx = engine.gcl_layer(engine.gcl_params(gcl), h, xx, lay, g_ctx, float(gcl.coord_change_maximum), pd)
        h, xx, alpha, z, bias = engine.att_layer(packs[i], h, xx, z, lay, g_int, pairs, batch_id,
                                                 float(att.coord_change_maximum), pd, bias,
                                                 packs[i + 1] if i + 1 < gnn.n_layers else None)
        a = torch.empty_like(alpha)
        a[g_int.order] = alpha
        atts.append(a)
        xx = ops.las_step(xx, x_las, las, lay, float(gnn.geometry_reg_step_size), 15.0 / scale)
    ol = gnn.out_layer
    h, xx = engine.gcl_layer(engine.gcl_params(ol), h, xx, lay, g_ctx, float(ol.coord_change_maximum), pd)
    h = ops.linear(engine._drop(h, pd), gnn.linear_out.weight.to(wd).contiguous(), gnn.linear_out.bias)
    xo = xx.reshape(shp)
    zd = _z_dense(z, pm)
    return (h, xo, atts, zd) if return_attention else (h, xo, zd)
