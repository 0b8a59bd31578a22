"""TEST INFRASTRUCTURE -- CPU restatement of the FABind+ docking stack (the ORACLE for SURVEY.md row a18).

Checker only (see fabind_oracle.py for the rules).  Restates, on top of the FABind-v1 helpers, what changes in
`/root/reference/FABind_plus/fabind/models/` under the FABind+ production flags (SURVEY.md App. C:
--use-ln-mlp --mlp-hidden-scale 1 --mean-layers 5, rm_layernorm, rel_dis_pair_bias='no', mha_heads=4,
inter_additional_mlp off, only_last_LAS off; dropout = identity in eval):

  * MLP / MLPwithLastAct / MLPwoBias                  FABind_plus/fabind/models/model_utils.py:10-74
  * MC_E_GCL with LN-MLPs                             FABind_plus/fabind/models/egnn.py:20-118
  * MC_Att_L (pair embedding threaded through)        FABind_plus/fabind/models/egnn.py:119-300
  * CrossAttentionModule (LN-MLP transitions, pair)   FABind_plus/fabind/models/cross_att.py:7-47
  * MCAttEGNN.forward -> (h, x, pair_embed)           FABind_plus/fabind/models/egnn.py:358-433
  * EfficientMCAttModel.forward -> (X, H, pair_embed) FABind_plus/fabind/models/att_model.py:165-223

PARITY PIN: pinned against outputs of the reference itself executed in the build container
(`oracle/make_golden.py plus` -> `tests/golden/plus_stack_*.npz`; `tests/test_oracle_golden.py`).
"""
import torch
import torch.nn.functional as F

from fabind_oracle import (Layout, _lin, construct_edges, coord2radial, interaction, las_step, mha, seg_softmax, seg_sum)


def ln_mlp(sd, pre, x, last_act):
    """MLP (last_act=False, also MLPwoBias: its linear2 simply has no bias) / MLPwithLastAct (last_act=True) with
    use_ln_mlp: LayerNorm -> linear1 -> relu -> linear2 (-> relu).  model_utils.py:10-74."""
    w = sd[pre + "layernorm.weight"]
    x = F.layer_norm(x, (w.shape[0],), w, sd[pre + "layernorm.bias"], 1e-5)
    y = _lin(sd, pre + "linear2", torch.relu(_lin(sd, pre + "linear1", x)))
    return torch.relu(y) if last_act else y


def gcl_forward(sd, pre, h, row, col, x, batch_id, B, clamp):
    """MC_E_GCL.forward, FABind_plus egnn.py:104-118."""
    N = h.shape[0]
    rad, d = coord2radial(row, col, x, batch_id, B)
    m = ln_mlp(sd, pre + "edge_mlp.", torch.cat([h[row], h[col], rad[:, None]], 1), True)
    s = ln_mlp(sd, pre + "coord_mlp.", m, False)                                   # [E,1]
    trans = d * s
    cnt = seg_sum(torch.ones_like(trans), row, N).clamp(min=1)
    x_new = x + (seg_sum(trans, row, N) / cnt).clamp(-clamp, clamp)
    agg = seg_sum(m, row, N)
    h_new = h + ln_mlp(sd, pre + "node_mlp.", torch.cat([h, agg], 1), True)
    return h_new, x_new


def row_attention(sd, pre, xi, xj, z, zmask, mask_i, heads):
    """RowAttentionBlock.forward (rm_layernorm, rel_dis_pair_bias='no'), cross_att.py:72-89."""
    mask_bias = (1e9 * (zmask.to(xi.dtype) - 1))[:, None, :, :]
    pb = _lin(sd, pre + "linear", z) * torch.sigmoid(_lin(sd, pre + "linear_g", z))
    pb = pb.permute(0, 3, 1, 2)
    return xi + mha(sd, pre + "mha.", xi, xj, [mask_bias, pb], heads) * mask_i.to(xi.dtype)[..., None]


def cross_attention(sd, pre, p, pm, c, cm, z, zm, heads=4):
    """CrossAttentionModule.forward, FABind_plus cross_att.py:20-47: the pair embedding is UPDATED and returned
    (no residual around pair_transition)."""
    p = row_attention(sd, pre + "p_attention_block.", p, c, z, zm, pm, heads)
    c = row_attention(sd, pre + "c_attention_block.", c, p, z.transpose(1, 2), zm.transpose(1, 2), cm, heads)
    p = p + ln_mlp(sd, pre + "p_transition.", p, True)
    c = c + ln_mlp(sd, pre + "c_transition.", c, True)
    z = z + interaction(sd, pre + "inter_layer.", p, c, pm, cm)[0]
    z = ln_mlp(sd, pre + "pair_transition.", z, True) * zm.to(z.dtype)[..., None]
    return p, c, z


def att_forward(sd, pre, h, row, col, x, lay, z, zm, batch_id, B, clamp):
    """MC_Att_L.forward, FABind_plus egnn.py:277-300 -> (h, x, alpha, z_updated)."""
    N = h.shape[0]
    p, pm, c, cm = lay.dense(h)
    p, c, z = cross_attention(sd, pre + "cross_attn_module.", p, pm, c, cm, z, zm)
    h = lay.undense(p, c)
    eb = batch_id[row]
    lig_row = lay.is_c[row]
    cl = torch.where(lig_row, lay.c_local[row], lay.c_local[col])
    pl = torch.where(lig_row, lay.p_local[col], lay.p_local[row])
    pair_off = z[eb, pl, cl]                                                        # [E,H]
    rad, d = coord2radial(row, col, x, batch_id, B)
    q = _lin(sd, pre + "linear_q", h[row])
    kv = _lin(sd, pre + "linear_kv", torch.cat([rad[:, None], h[col]], 1))
    k, v = kv[:, 0::2], kv[:, 1::2]
    alpha = (q * k).sum(1) + _lin(sd, pre + "attn_bias_proj", pair_off)[:, 0]
    alpha = seg_softmax(alpha, row, N)
    h_new = h + seg_sum(alpha[:, None] * v, row, N)
    cv = alpha[:, None] * ln_mlp(sd, pre + "coord_mlp.", v, False)
    x_new = x + seg_sum(d * cv, row, N).clamp(-clamp, clamp)
    return h_new, x_new, alpha, z


def egnn_forward(sd, pre, h, x, ctx, inter, las, x_las, lay, z0, zm, batch_id, n_layers, scale, step=1e-3, capture=None):
    """MCAttEGNN.forward, FABind_plus egnn.py:358-433 -> (h_out, x, pair_embed)."""
    B = lay.B
    clamp = 10.0 / scale
    h = _lin(sd, pre + "linear_in", h)
    z = z0
    for i in range(n_layers):
        h, x = gcl_forward(sd, pre + "gcl_%d." % i, h, ctx[0], ctx[1], x, batch_id, B, clamp)
        if capture is not None:
            capture["gcl_%d.h" % i], capture["gcl_%d.x" % i] = h.clone(), x.clone()
        h, x, alpha, z = att_forward(sd, pre + "att_%d." % i, h, inter[0], inter[1], x, lay, z, zm, batch_id, B, clamp)
        if capture is not None:
            capture["att_%d.h" % i], capture["att_%d.x" % i], capture["att_%d.alpha" % i] = h.clone(), x.clone(), alpha.clone()
        x = las_step(x, x_las, las, step, 15.0 / scale)
    h, x = gcl_forward(sd, pre + "out_layer.", h, ctx[0], ctx[1], x, batch_id, B, clamp)
    return _lin(sd, pre + "linear_out", h), x, z


def stack_forward(sd, pre, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index, coord_LAS,
                  n_layers, n_iter, scale=5.0, inter_cutoff=10.0, intra_cutoff=8.0, step=1e-3, capture=None):
    """EfficientMCAttModel.forward (refine_coord), FABind_plus att_model.py:165-223.
    Every refinement iteration starts again from z0 = inter_layer(H_in); the pair embedding of the LAST iteration is
    returned.  -> (X_out [N,1,3], H_out, pair_embed [B,Pmax,Cmax,H])."""
    lay = Layout(batch_id, segment_id)
    p, pm, c, cm = lay.dense(H)
    z0, zm = interaction(sd, pre + "inter_layer.", p, c, pm, cm)
    z0 = z0 * zm.to(z0.dtype)[..., None]
    x = X[:, 0].clone()
    x_las = coord_LAS[:, 0]
    Hout = zout = None
    for r in range(n_iter):
        ctx, inter = construct_edges(x[:, None], batch_id, segment_id, is_global, inter_cutoff / scale, intra_cutoff / scale)
        ctx = torch.cat([compound_edge_index, ctx], 1)
        last = r == n_iter - 1
        with torch.set_grad_enabled(last and torch.is_grad_enabled()):
            Hout, z, zout = egnn_forward(sd, pre + "gnn.", H, x, ctx, inter, LAS_edge_index, x_las, lay, z0, zm, batch_id,
                                         n_layers, scale, step, capture if last else None)
            x = torch.where(mask[:, None], z, x)
    return x[:, None], Hout, zout
