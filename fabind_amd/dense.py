"""Reference-signature forwards of the inner modules: dense, zero-padded tensors in and out, the
arithmetic on the HIP kernels.

The reference's `MC_Att_L`, `CrossAttentionModule`, `RowAttentionBlock`, `Attention`, `InteractionModule` and the
positional `MCAttEGNN.forward` (FABind/fabind/models/egnn.py:308-333,392-466, cross_att.py:24-54,118-134,
model_utils.py:96-159,200-223) exchange `to_dense_batch`-padded `[B, P, H]` / `[B, C, H]` node tensors and a dense
`[B, P, C, H]` pair tensor with boolean masks.  The stack itself (fabind_amd/engine.py) never builds those; this module
is the adapter for callers that do: it packs the valid rows / pairs into the ragged lists the kernels consume
(index glue: boolean-mask gathers and scatters, the packing the FABind+ engine already uses), runs the same kernels as
the stack, and scatters the result back into zero-padded tensors of the reference's shapes.  Masks are `to_dense_batch`
masks (valid entries first in every row); anything else raises.

No CPU path: every function raises on non-HIP tensors through `_lib.ptr`."""
import math

import numpy as np
import torch

from . import engine
from . import kernels as K
from . import ops


def _require_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError("fabind_amd: %s runs on a HIP device only (no CPU fallback); got %s" % (who, t.device))


def _prefix_counts(mask, who):
    """Row counts of a to_dense_batch mask [B, n]; raises unless the valid entries come first in every row."""
    cnt = mask.sum(1)
    ok = (mask == (torch.arange(mask.shape[1], device=mask.device)[None, :] < cnt[:, None])).all()
    if not bool(ok):
        raise ValueError("fabind_amd: %s expects to_dense_batch masks (valid entries first in every row)" % who)
    return cnt


def pair_layout(i_mask, j_mask, who="dense adapter"):
    """engine.Layout of a synthetic node order [j rows..., i rows...] per batch element: the `p` side of the layout is the
    query side i, the `c` side the key side j, pairs are i-major (pair = pair_off[b] + i * J_b + j)."""
    I, J = _prefix_counts(i_mask, who), _prefix_counts(j_mask, who)
    B, dev = i_mask.shape[0], i_mask.device
    n = I + J
    batch_id = torch.repeat_interleave(torch.arange(B, device=dev), n)
    off = torch.cumsum(n, 0) - n
    loc = torch.arange(batch_id.shape[0], device=dev) - off[batch_id]
    segment = (loc >= J[batch_id]).float()
    lay = engine.Layout(batch_id, segment)
    if int(lay.P.min()) < 1 or int(lay.C.min()) < 1:
        raise ValueError("fabind_amd: %s needs at least one valid row on both sides of every batch element" % who)
    return lay


def _to_nodes(lay, i_list, j_list):
    """[N, H] node-layout array ([j rows | i rows] per element) from the two compact lists (row scatter: data movement)."""
    h = torch.zeros((lay.N, i_list.shape[1]), dtype=torch.float32, device=i_list.device)
    return h.index_copy(0, lay.c_index64, j_list.float()).index_copy(0, lay.p_index64, i_list.float())


def _row_bias(blk, z_list):
    """[pairs, 8] = (linear | linear_g)(z): the kernel evaluates lin * sigmoid(gate) itself (cross_att.py:125)."""
    wd = ops.mm_dtype()
    Wb = torch.cat([blk.linear.weight, blk.linear_g.weight]).to(wd).contiguous()
    bb = torch.cat([blk.linear.bias, blk.linear_g.bias]).contiguous()
    return ops.linear(z_list, Wb, bb)


def _check_mha(att, who):
    if att.no_heads != 4 or att.c_hidden != 32 or att.linear_g is None:
        raise NotImplementedError("%s: the attention kernels are built for 4 gated heads x 32 channels (the reference's only "
                                  "configuration, cross_att.py:105,116)" % who)


# ------------------------------------------------------------------------------------------------
# Attention.forward (model_utils.py:96-159)
# ------------------------------------------------------------------------------------------------
def attention(att, q_x, kv_x, biases=None):
    """[*, Q, Cq], [*, K, Ck], biases broadcastable to [*, heads, Q, K] -> [*, Q, Cq].  Every (query, key) pair of the
    padded tensors takes part, exactly like the reference: masking is whatever the caller put into `biases` (-1e9)."""
    _require_cuda(q_x, "Attention")
    _check_mha(att, "Attention")
    lead, Q, Kn = q_x.shape[:-2], q_x.shape[-2], kv_x.shape[-2]
    Bn = int(np.prod(lead)) if len(lead) else 1
    dev = q_x.device
    wd = ops.mm_dtype()
    qf, kf = q_x.reshape(Bn * Q, -1).float().contiguous(), kv_x.reshape(Bn * Kn, -1).float().contiguous()
    Wqg = torch.cat([att.linear_q.weight, att.linear_g.weight]).to(wd).contiguous()
    bqg = torch.cat([torch.zeros_like(att.linear_g.bias), att.linear_g.bias]).contiguous()
    Wkv = torch.cat([att.linear_k.weight, att.linear_v.weight]).to(wd).contiguous()
    qg = ops.linear(qf, Wqg, bqg)
    kv = ops.linear(kf, Wkv)
    # the summed bias as the kernel's (lin, gate) pair with gate = +40: sigmoid(40) == 1.0f exactly
    tot = torch.zeros(lead + (4, Q, Kn), dtype=torch.float32, device=dev)
    for b in (biases or []):
        tot = tot + b.float()
    lin = tot.reshape(Bn, 4, Q, Kn).permute(0, 2, 3, 1).reshape(Bn * Q * Kn, 4)
    bias8 = torch.cat([lin, torch.full_like(lin, 40.0)], 1).contiguous()
    b = np.arange(Bn, dtype=np.int64)
    po = b * Q * Kn
    desc = np.stack([b * Q, np.full(Bn, Q), b * Kn, np.full(Bn, Kn), po & 0xFFFFFFFF, po >> 32, np.full(Bn, Kn),
                     np.ones(Bn, dtype=np.int64)], 1).reshape(-1)
    desc = torch.from_numpy((desc & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).to(dev).view(-1, 8)
    og = ops.cross_attn(qg, kv, bias8, 0, 4, desc, Bn, Q, Kn, 1.0 / math.sqrt(att.c_hidden))
    o = ops.linear(og, att.linear_o.weight.to(wd).contiguous(), att.linear_o.bias)
    return o.reshape(lead + (Q, -1))


# ------------------------------------------------------------------------------------------------
# RowAttentionBlock.forward (cross_att.py:118-134)
# ------------------------------------------------------------------------------------------------
def row_attention(blk, node_embed_i, node_embed_j, pair_embed, pair_mask, node_mask_i):
    _require_cuda(node_embed_i, "RowAttentionBlock")
    if not blk.rm_layernorm:
        raise NotImplementedError("RowAttentionBlock: only --rm-layernorm (production) is built")
    _check_mha(blk.mha, "RowAttentionBlock")
    mask_j = pair_mask.any(1)
    lay = pair_layout(node_mask_i, mask_j, "RowAttentionBlock")
    if int(pair_mask.sum()) != lay.n_pairs:
        raise ValueError("fabind_amd: RowAttentionBlock expects pair_mask = node_mask_i x node_mask_j")
    xi, xj = node_embed_i[node_mask_i].float(), node_embed_j[mask_j].float()
    bias = _row_bias(blk, pair_embed[pair_mask].float().contiguous())
    a = blk.mha
    wd = ops.mm_dtype()
    Wqg = torch.cat([a.linear_q.weight, a.linear_g.weight]).to(wd).contiguous()
    bqg = torch.cat([torch.zeros_like(a.linear_g.bias), a.linear_g.bias]).contiguous()
    Wkv = torch.cat([a.linear_k.weight, a.linear_v.weight]).to(wd).contiguous()
    qg, kv = ops.linear(xi.contiguous(), Wqg, bqg), ops.linear(xj.contiguous(), Wkv)
    og = ops.cross_attn(qg, kv, bias, 0, 4, lay.desc_p, lay.B, lay.max_P, lay.max_C, 1.0 / math.sqrt(32.0))
    upd = ops.linear(og, a.linear_o.weight.to(wd).contiguous(), a.linear_o.bias)
    pd = blk.dropout.p if blk.training else 0.0
    xi = xi + engine._drop(upd, pd)
    out = node_embed_i.float().clone()            # padded query rows: update * node_mask_i = 0 in the reference
    out[node_mask_i] = xi
    return out


# ------------------------------------------------------------------------------------------------
# InteractionModule.forward (model_utils.py:200-223)
# ------------------------------------------------------------------------------------------------
def _interaction_list(im, p_list, c_list, lay, p_node, c_node):
    """linear_out(linear_p(p_i) * linear_c(c_j)) on a pair list; p_node / c_node index the concatenated [p rows; c rows]."""
    wd = ops.mm_dtype()
    Hh = im.hidden_dim
    Wpc = torch.cat([im.linear_p.weight, im.linear_c.weight]).to(wd).contiguous()
    bpc = torch.cat([im.linear_p.bias, im.linear_c.bias]).contiguous()
    T = ops.linear(torch.cat([p_list, c_list], 0).contiguous(), Wpc, bpc)          # [sumP + sumC, 2 Hh]: (a | b) of every row
    from .plus.engine import pair_had
    hd = pair_had(T, Hh, p_node, c_node, ops.act_dtype())
    return hd, im.linear_out.weight.to(wd).contiguous(), im.linear_out.bias


def interaction(im, p_embed, c_embed, p_mask=None, c_mask=None):
    _require_cuda(p_embed, "InteractionModule")
    if not im.rm_layernorm:
        raise NotImplementedError("InteractionModule: only --rm-layernorm (production) is built")
    squeeze = p_embed.dim() == 2
    if squeeze:
        p_embed, c_embed = p_embed[None], c_embed[None]
        p_mask = None if p_mask is None else p_mask[None]
        c_mask = None if c_mask is None else c_mask[None]
    if p_mask is None:
        p_mask = torch.ones(p_embed.shape[:-1], dtype=torch.bool, device=p_embed.device)
    if c_mask is None:
        c_mask = torch.ones(c_embed.shape[:-1], dtype=torch.bool, device=c_embed.device)
    p_mask, c_mask = p_mask.bool(), c_mask.bool()
    lay = pair_layout(p_mask, c_mask, "InteractionModule")
    from .plus.engine import PairList
    pairs = PairList(lay, p_embed.device)
    p_list, c_list = p_embed[p_mask].float(), c_embed[c_mask].float()
    # rows of the concatenated [p_list; c_list]: p row = poff[b] + i, c row = sumP + coff[b] + j
    poff = torch.as_tensor(np.concatenate([[0], np.cumsum(lay.P)])[:-1], device=p_embed.device)
    coff = torch.as_tensor(lay.coff[:-1], device=p_embed.device)
    p_node = (poff[pairs.b] + pairs.i).to(torch.int32)
    c_node = (lay.sumP + coff[pairs.b] + pairs.j).to(torch.int32)
    hd, Wo, bo = _interaction_list(im, p_list, c_list, lay, p_node, c_node)
    z = ops.linear(hd, Wo, bo)
    inter_mask = p_mask[:, :, None] & c_mask[:, None, :]
    out = torch.zeros(inter_mask.shape + (z.shape[1],), dtype=torch.float32, device=z.device)
    out[inter_mask] = z.float()
    if squeeze:
        return out[0], inter_mask[0]
    return out, inter_mask


# ------------------------------------------------------------------------------------------------
# CrossAttentionModule.forward (cross_att.py:24-54)
# ------------------------------------------------------------------------------------------------
class _CrossState:
    """Packed view of one (p, c, z) triple: layout, pair list and the compact tensors."""

    def __init__(self, p_mask, c_mask, pair_mask, who):
        p_mask, c_mask = p_mask.bool(), c_mask.bool()
        self.lay = lay = pair_layout(p_mask, c_mask, who)
        if int(pair_mask.sum()) != lay.n_pairs:
            raise ValueError("fabind_amd: %s expects pair_mask = p_mask x c_mask" % who)
        from .plus.engine import PairList
        self.pairs = PairList(lay, p_mask.device)
        self.p_mask, self.c_mask, self.pair_mask = p_mask, c_mask, pair_mask.bool()


def _cross_nodes(cam, h, lay, z_list, pdrop):
    """Node path of the module on the node layout: both RowAttentionBlocks + both transitions (engine.cross_attention)."""
    P = engine.cam_node_params(cam)
    bias = [_row_bias(cam.p_attention_block, z_list), _row_bias(cam.c_attention_block, z_list)]
    return engine.cross_attention(P, h, lay, bias, 0, pdrop), P


def _pair_update_rows(cam, P, h, z_rows, p_node, c_node):
    """pair_transition(z + inter_layer(p', c')) at the given pairs (node ids in the node layout): cross_att.py:51-53."""
    wd = ops.mm_dtype()
    ab32 = ops.linear(h, P["W_ab32"], P["b_ab32"])                              # [N, 128] = (a32 | 0 | b32 | 0)
    from .plus.engine import pair_had
    hd = pair_had(ab32, 64, p_node, c_node, ops.act_dtype())                    # [n, 64]
    i32, pt = cam.inter_layer, cam.pair_transition
    Wo = torch.cat([i32.linear_out.weight, torch.zeros_like(i32.linear_out.weight)], 1).to(wd).contiguous()
    z1 = ops.linear(hd, Wo, i32.linear_out.bias, residual=z_rows.float().contiguous())
    t = ops.linear(z1, pt.linear_1.weight.to(wd).contiguous(), pt.linear_1.bias, act_epi=K.ACT_RELU, out_dtype=ops.act_dtype())
    return ops.linear(t, pt.linear_2.weight.to(wd).contiguous(), pt.linear_2.bias)


def cross_attention(cam, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask, c_c_dist_embed=None,
                    p_p_dist_embed=None):
    _require_cuda(p_embed_batched, "CrossAttentionModule")
    st = _CrossState(p_mask, c_mask, pair_mask, "CrossAttentionModule")
    lay = st.lay
    h = _to_nodes(lay, p_embed_batched[st.p_mask], c_embed_batched[st.c_mask])
    z_list = pair_embed[st.pair_mask].float().contiguous()
    pd = cam.p_attention_block.dropout.p if cam.training else 0.0
    h, P = _cross_nodes(cam, h, lay, z_list, pd)
    z_new = _pair_update_rows(cam, P, h, z_list, st.pairs.p_node, st.pairs.c_node)
    p_out = torch.zeros_like(p_embed_batched, dtype=torch.float32)
    c_out = torch.zeros_like(c_embed_batched, dtype=torch.float32)
    p_out[st.p_mask] = h.index_select(0, lay.p_index64)
    c_out[st.c_mask] = h.index_select(0, lay.c_index64)
    # the reference leaves the padded node rows at (input + transition(input)) of the zero padding = transition bias terms;
    # they are masked out by every consumer (to_dense_batch masks), here they are zeros
    z_out = torch.zeros(st.pair_mask.shape + (z_new.shape[1],), dtype=torch.float32, device=z_new.device)
    z_out[st.pair_mask] = z_new.float()
    return p_out, c_out, z_out


# ------------------------------------------------------------------------------------------------
# MC_Att_L.forward (egnn.py:308-333) and the positional MCAttEGNN.forward (egnn.py:392-466)
# ------------------------------------------------------------------------------------------------
class EdgeListInterGraph:
    """The inter graph of a caller-supplied edge list [2, E] (both directions present) under the attribute names the
    kernels read: rows sorted, columns sorted inside a row (what fabind_inter_meta's mirror search needs), plus the
    permutation back to the caller's edge order."""

    def __init__(self, edge_index, lay):
        row, col = edge_index[0].long(), edge_index[1].long()
        N = lay.N
        key, order = torch.sort(row * N + col)
        self.order = order
        srow, scol = row[order], col[order]
        self.row_int, self.col_int = srow.to(torch.int32).contiguous(), scol.to(torch.int32).contiguous()
        rp = torch.zeros(N + 1, dtype=torch.int32, device=row.device)
        rp[1:] = torch.cumsum(torch.bincount(srow, minlength=N), 0).to(torch.int32)
        self.rp_int, self.E_int, self.N = rp, int(row.shape[0]), N
        k2 = torch.sort(scol * N + srow)[0]
        if self.E_int % 2 or not bool((key == k2).all()):
            raise ValueError("fabind_amd: MC_Att_L expects a symmetric inter-edge list (every (u, v) with its (v, u))")
        self.red_off, self.red_idx, self.red_c, self.red_p, self.mirror = K.inter_meta(
            lay.node_off, lay.c_cnt, lay.B, rp, self.col_int, self.row_int)

    def int_by_col(self):
        return self.rp_int, self.mirror


def att_layer(m, h, edge_index, coord, segment_id, batch_id, pair_embed_batched, pair_mask, lay=None, g=None, z_list=None,
              pairs=None):
    """MC_Att_L.forward with the reference's arguments -> (h', coord', alpha [E] in the caller's edge order)."""
    _require_cuda(h, "MC_Att_L")
    lay = lay or engine.Layout(batch_id, segment_id)
    g = g or EdgeListInterGraph(edge_index, lay)
    if pairs is None:
        from .plus.engine import PairList
        pairs = PairList(lay, h.device)
    if z_list is None:
        pm = pair_mask.bool()
        if int(pm.sum()) != lay.n_pairs:
            raise ValueError("fabind_amd: MC_Att_L expects pair_mask = to_dense_batch masks of the two segments")
        z_list = pair_embed_batched[pm].float().contiguous()
    H = h.shape[1]
    x = coord.reshape(-1, 3).float().contiguous()
    cam = m.cross_attn_module
    training = m.training
    h2, P = _cross_nodes(cam, h.float().contiguous(), lay, z_list, cam.p_attention_block.dropout.p if training else 0.0)
    # pair embedding update at the inter-edge pairs only (all trio_encoder gathers, egnn.py:286-304) -> scalar attention bias
    ridx = pairs.index_of(g.red_p, g.red_c, batch_id)
    z_red = _pair_update_rows(cam, P, h2, ops.take_rows(z_list, ridx), g.red_p, g.red_c)          # [n_red, H]
    wb = torch.zeros((8, H), dtype=m.attn_bias_proj.weight.dtype, device=h.device)
    wb = torch.cat([m.attn_bias_proj.weight, wb[1:]], 0)
    bb = torch.cat([m.attn_bias_proj.bias, torch.zeros(7, dtype=m.attn_bias_proj.bias.dtype, device=h.device)])
    bias_part = ops.linear(z_red, wb.to(ops.mm_dtype()).contiguous(), bb.contiguous())          # [n_red, 8]: column 0
    A = engine.att_edge_params(m)
    qkv = ops.linear(h2, A["Wqkv"], A["bqkv"])
    d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay, g.int_by_col)
    h_new, x_new, alpha = ops.inter_attn(qkv, None, H, h2, x, d, rhohat, g, bias_part, A["w_rk"], A["w_rv"], A["wcr"],
                                         A["w3"], float(m.coord_change_maximum), Wc=A["Wc"], bc=A["bc"])
    if training and m.dropout.p > 0.0:
        h_new = h2 + engine._drop(h_new - h2, m.dropout.p)
    alpha_out = torch.empty_like(alpha)
    alpha_out[g.order] = alpha
    return h_new, x_new.reshape(coord.shape), alpha_out


def egnn_forward(gnn, h, x, ctx_edges, att_edges, LAS_edge_list, batched_complex_coord_LAS, segment_id=None, batch_id=None,
                 reduced_tuple=None, pair_embed_batched=None, pair_mask=None, LAS_mask=None, p_p_dist_embed=None,
                 c_c_dist_embed=None, mask=None, ctx_edge_attr=None, att_edge_attr=None, return_attention=False):
    """MCAttEGNN.forward with the reference's positional arguments (egnn.py:392-466)."""
    _require_cuda(h, "MCAttEGNN")
    assert ctx_edge_attr is None and att_edge_attr is None
    a = gnn.args
    if getattr(a, "fix_pocket", False) or getattr(a, "rm_LAS_constrained_optim", False) or getattr(gnn, "dense", False):
        # (the constructor refuses these too; this catches flags flipped on a built module -- egnn.py:404-405,433,458-460)
        raise NotImplementedError("MCAttEGNN.forward: fix_pocket / rm_LAS_constrained_optim / dense are not built (production flags only)")
    lay = engine.Layout(batch_id, segment_id)
    from .plus.engine import PairList
    pairs = PairList(lay, h.device)
    g_int = EdgeListInterGraph(att_edges, lay)
    g_ctx = engine.EdgeListGraph(ctx_edges, lay.N)
    pm = pair_mask.bool()
    z_list = pair_embed_batched[pm].float().contiguous()
    wd = ops.mm_dtype()
    training = gnn.training
    h = engine._drop(ops.linear(h.float().contiguous(), gnn.linear_in.weight.to(wd).contiguous(), gnn.linear_in.bias),
                     gnn.dropout.p if training else 0.0)
    shp = x.shape
    xx = x.reshape(-1, 3).float().contiguous()
    if batched_complex_coord_LAS.dim() == 3:
        batched_complex_coord_LAS.squeeze_(1)                   # the reference squeezes the caller's tensor in place (egnn.py:435)
    x_las = batched_complex_coord_LAS.reshape(-1, 3).float().contiguous()
    las = (LAS_edge_list[0].to(torch.int32).contiguous(), LAS_edge_list[1].to(torch.int32).contiguous())
    las = las + (lay.ranges(las[0]),)
    scale = 10.0 / float(gnn.gcl_0.coord_change_maximum)        # coord_change_maximum = normalize_coord(10)
    atts = []
    for i in range(gnn.n_layers):
        gcl, att = getattr(gnn, "gcl_%d" % i), getattr(gnn, "att_%d" % i)
        h, xx = engine.gcl_layer(engine.gcl_params(gcl), h, xx, lay, g_ctx, float(gcl.coord_change_maximum),
                                 gcl.dropout.p if training else 0.0)
        h, xx, alpha = att_layer(att, h, att_edges, xx, segment_id, batch_id, None, None, lay=lay, g=g_int, z_list=z_list,
                                 pairs=pairs)
        atts.append(alpha)
        for _ in range(int(gnn.geom_reg_steps)):                # egnn.py:436 (the reference fixes it to 1, egnn.py:357)
            xx = ops.las_step(xx, x_las, las, lay, float(gnn.geometry_reg_step_size), 15.0 / scale)
    ol = gnn.out_layer
    h, xx = engine.gcl_layer(engine.gcl_params(ol), h, xx, lay, g_ctx, float(ol.coord_change_maximum),
                             ol.dropout.p if training else 0.0)
    h = ops.linear(engine._drop(h, gnn.dropout.p if training else 0.0), gnn.linear_out.weight.to(wd).contiguous(),
                   gnn.linear_out.bias)
    xo = xx.reshape(shp)
    return (h, xo, atts) if return_attention else (h, xo)
