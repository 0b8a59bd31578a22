"""TEST INFRASTRUCTURE -- CPU restatement of the FABind docking hot path (the ORACLE).

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product path (`fabind_amd/`) must never call
into it; it fails loudly when the HIP library is missing instead.

It restates, in plain PyTorch CPU ops and from scratch (functional style over a `state_dict`,
vectorised index arithmetic instead of the reference's per-sample python loops), the algorithm of

  * ComplexGraph.construct_edges / _radial_edges   FABind/fabind/models/att_model.py:37-128
  * EfficientMCAttModel.forward                    FABind/fabind/models/att_model.py:170-246
  * MCAttEGNN.forward (+ LAS step)                 FABind/fabind/models/egnn.py:392-466
  * MC_E_GCL (edge/coord/node model)               FABind/fabind/models/egnn.py:68-144
  * MC_Att_L (trio_encoder/att/node/coord model)   FABind/fabind/models/egnn.py:186-333
  * coord2radial, unsorted_segment_sum/mean        FABind/fabind/models/egnn.py:767-821
  * CrossAttentionModule / RowAttentionBlock       FABind/fabind/models/cross_att.py:24-54,118-134
  * Attention/_attention/Transition/InteractionModule  FABind/fabind/models/model_utils.py:21-223
  * IaBNet...forward / .inference                  FABind/fabind/models/model.py:82-580
  * get_keepNode_tensor, gumbel_softmax_no_random  FABind/fabind/utils/utils.py:150-158,687-699
  * train-step loss                                FABind/fabind/main_fabind.py:398-417

in the production configuration (rm_layernorm, add_attn_pair_bias,
explicit_pair_embed, add_cross_attn_layer, norm_type=per_sample, refine=refine_coord, n_channel=1).
Third-party arithmetic the reference reaches through torch_scatter 2.1.0 (scatter_sum/add/mean/
softmax) and torch_geometric 2.4.0 (to_dense_batch) is restated from the documented semantics.

PARITY PIN: the reference ships no tests or golden vectors.  This oracle is
pinned against outputs of the reference itself executed in the build container
(`oracle/make_golden.py` -> `tests/golden/*.npz`; `tests/test_oracle_golden.py`).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    w = sd[name + ".weight"]
    y = x @ w.T
    b = sd.get(name + ".bias")
    return y if b is None else y + b


def _silu(x):
    return x * torch.sigmoid(x)


def seg_sum(src, index, n):
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    return out.index_add_(0, index, src)


def seg_softmax(logit, index, n):
    mx = logit.new_full((n,), float("-inf")).scatter_reduce(0, index, logit, reduce="amax")
    e = (logit - mx[index]).exp()
    return e / seg_sum(e, index, n)[index]


class Layout:
    """Index arithmetic for the fixed node order [glb_c, lig..., glb_p, prot...] per complex."""

    def __init__(self, batch_id, segment_id):
        self.B = int(batch_id.max()) + 1
        self.n = torch.bincount(batch_id, minlength=self.B)
        self.off = torch.cumsum(self.n, 0) - self.n
        seg1 = segment_id.to(torch.long)
        self.P = torch.bincount(batch_id, weights=seg1.double(), minlength=self.B).long()  # incl. glb_p
        self.C = self.n - self.P                                                            # incl. glb_c
        N = batch_id.shape[0]
        local = torch.arange(N) - self.off[batch_id]
        self.is_c = seg1 == 0
        self.c_local = local                       # valid where is_c
        self.p_local = local - self.C[batch_id]    # valid where ~is_c
        self.batch_id = batch_id
        self.Pmax, self.Cmax = int(self.P.max()), int(self.C.max())

    def dense(self, h):
        """to_dense_batch of the two segments: p [B,Pmax,H], c [B,Cmax,H] and masks."""
        H = h.shape[-1]
        p = h.new_zeros(self.B, self.Pmax, H)
        c = h.new_zeros(self.B, self.Cmax, H)
        pm = torch.zeros(self.B, self.Pmax, dtype=torch.bool)
        cm = torch.zeros(self.B, self.Cmax, dtype=torch.bool)
        b = self.batch_id
        p[b[~self.is_c], self.p_local[~self.is_c]] = h[~self.is_c]
        c[b[self.is_c], self.c_local[self.is_c]] = h[self.is_c]
        pm[b[~self.is_c], self.p_local[~self.is_c]] = True
        cm[b[self.is_c], self.c_local[self.is_c]] = True
        return p, pm, c, cm

    def undense(self, p, c):
        H = p.shape[-1]
        h = p.new_zeros(self.batch_id.shape[0], H)
        b = self.batch_id
        h[~self.is_c] = p[b[~self.is_c], self.p_local[~self.is_c]]
        h[self.is_c] = c[b[self.is_c], self.c_local[self.is_c]]
        return h


# ------------------------------------------------------------------------------------------------
# edge construction (att_model.py:37-128); returns reference ordering
# ------------------------------------------------------------------------------------------------
def construct_edges(X, batch_id, segment_id, is_global, inter_cutoff, intra_cutoff):
    x = X[:, 0]
    B = int(batch_id.max()) + 1
    n = torch.bincount(batch_id, minlength=B)
    off = torch.cumsum(n, 0) - n
    seg = segment_id.to(torch.long)
    radial, inter, gnorm, gglob, cand = [], [], [], [], None
    for b in range(B):
        lo, hi = int(off[b]), int(off[b] + n[b])
        xb, sb, gb = x[lo:hi], seg[lo:hi], is_global[lo:hi]
        d = torch.norm(xb[:, None, :] - xb[None, :, :], dim=-1)
        nb = hi - lo
        notself = ~torch.eye(nb, dtype=torch.bool)
        same_seg = sb[:, None] == sb[None, :]
        anyg = gb[:, None] | gb[None, :]
        r_m = notself & same_seg & (sb[:, None] == 1) & ~anyg & (d <= intra_cutoff)
        i_all = notself & ~same_seg & ~anyg
        i_m = i_all & (d <= inter_cutoff)
        gn_m = notself & same_seg & anyg
        gg_m = notself & gb[:, None] & gb[None, :]
        if cand is None and i_all.any():
            cand = torch.nonzero(i_all)[0] + lo
        for m, lst in ((r_m, radial), (i_m, inter), (gn_m, gnorm), (gg_m, gglob)):
            lst.append(torch.nonzero(m).T + lo)
    cat = lambda l: torch.cat(l, 1)
    inter = cat(inter)
    if inter.shape[1] == 0:  # att_model.py:85-86 fallback
        inter = torch.stack([torch.stack([cand[0], cand[1]]), torch.stack([cand[1], cand[0]])])
    ctx = torch.cat([cat(radial), cat(gnorm), cat(gglob)], 1)
    return ctx, inter


def coord2radial(row, col, x, batch_id, B):
    """egnn.py:767-787 with norm_type='per_sample', n_channel=1.  x [N,3]."""
    d = x[row] - x[col]
    rad = (d * d).sum(-1)
    eb = batch_id[row]
    norm = seg_sum(rad * rad, eb, B).sqrt()
    return rad / norm[eb], d


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def gcl_forward(sd, pre, h, row, col, x, batch_id, B, clamp):
    """MC_E_GCL.forward, egnn.py:130-144 (dropout = identity in eval)."""
    N = h.shape[0]
    rad, d = coord2radial(row, col, x, batch_id, B)
    e_in = torch.cat([h[row], h[col], rad[:, None]], 1)
    m = _silu(_lin(sd, pre + "edge_mlp.2", _silu(_lin(sd, pre + "edge_mlp.0", e_in))))
    s = _lin(sd, pre + "coord_mlp.2", _silu(_lin(sd, pre + "coord_mlp.0", m)))  # [E,1]
    trans = d * s
    cnt = seg_sum(torch.ones_like(trans), row, N).clamp(min=1)
    x_new = x + (seg_sum(trans, row, N) / cnt).clamp(-clamp, clamp)
    agg = seg_sum(m, row, N)
    h_new = h + _lin(sd, pre + "node_mlp.2", _silu(_lin(sd, pre + "node_mlp.0", torch.cat([h, agg], 1))))
    return h_new, x_new


def interaction(sd, pre, p, c, pm, cm):
    """InteractionModule.forward (opm=False, rm_layernorm), model_utils.py:200-223."""
    a = _lin(sd, pre + "linear_p", p)
    b = _lin(sd, pre + "linear_c", c)
    z = torch.einsum("bik,bjk->bijk", a, b)
    mask = pm[:, :, None] & cm[:, None, :]
    return _lin(sd, pre + "linear_out", z) * mask[..., None], mask


def mha(sd, pre, q_x, kv_x, biases, heads=4):
    """Attention.forward, model_utils.py:96-159."""
    q, k, v = _lin(sd, pre + "linear_q", q_x), _lin(sd, pre + "linear_k", kv_x), _lin(sd, pre + "linear_v", kv_x)
    Bn, Q, D = q.shape
    hd = D // heads
    q = q.view(Bn, Q, heads, hd) / math.sqrt(hd)
    k = k.view(Bn, -1, heads, hd)
    v = v.view(Bn, -1, heads, hd)
    a = torch.einsum("bqhd,bkhd->bhqk", q, k)
    for bias in biases:
        a = a + bias
    a = torch.softmax(a, -1)
    o = torch.einsum("bhqk,bkhd->bqhd", a, v)
    g = torch.sigmoid(_lin(sd, pre + "linear_g", q_x)).view(Bn, Q, heads, hd)
    return _lin(sd, pre + "linear_o", (o * g).reshape(Bn, Q, D))


def row_attention(sd, pre, xi, xj, z, zmask, mask_i):
    """RowAttentionBlock.forward (rm_layernorm), cross_att.py:118-134."""
    mask_bias = (1e9 * (zmask.to(xi.dtype) - 1))[:, None, :, :]
    pb = _lin(sd, pre + "linear", z) * torch.sigmoid(_lin(sd, pre + "linear_g", z))
    pb = pb.permute(0, 3, 1, 2)
    return xi + mha(sd, pre + "mha.", xi, xj, [mask_bias, pb]) * mask_i.to(xi.dtype)[..., None]


def transition(sd, pre, x):
    return _lin(sd, pre + "linear_2", torch.relu(_lin(sd, pre + "linear_1", x)))


def cross_attention(sd, pre, p, pm, c, cm, z, zm):
    """CrossAttentionModule.forward (no triangle blocks), cross_att.py:24-54."""
    p = row_attention(sd, pre + "p_attention_block.", p, c, z, zm, pm)
    c = row_attention(sd, pre + "c_attention_block.", c, p, z.transpose(1, 2), zm.transpose(1, 2), cm)
    p = p + transition(sd, pre + "p_transition.", p)
    c = c + transition(sd, pre + "c_transition.", c)
    z = z + interaction(sd, pre + "inter_layer.", p, c, pm, cm)[0]
    z = transition(sd, pre + "pair_transition.", z) * zm.to(z.dtype)[..., None]
    return p, c, z


def att_forward(sd, pre, h, row, col, x, lay, z0, zm, batch_id, B, clamp, return_pair=False):
    """MC_Att_L.forward, egnn.py:308-333 (trio_encoder 254-305, att/node/coord model 186-252)."""
    N = h.shape[0]
    p, pm, c, cm = lay.dense(h)
    p, c, z = cross_attention(sd, pre + "cross_attn_module.", p, pm, c, cm, z0, zm)
    h = lay.undense(p, c)
    eb = batch_id[row]
    lig_row = lay.is_c[row]
    cl = torch.where(lig_row, lay.c_local[row], lay.c_local[col])
    pl = torch.where(lig_row, lay.p_local[col], lay.p_local[row])
    pair_off = z[eb, pl, cl]                                         # [E,H]
    rad, d = coord2radial(row, col, x, batch_id, B)
    q = _lin(sd, pre + "linear_q", h[row])
    kv = _lin(sd, pre + "linear_kv", torch.cat([rad[:, None], h[col]], 1))
    k, v = kv[:, 0::2], kv[:, 1::2]
    alpha = (q * k).sum(1) + _lin(sd, pre + "attn_bias_proj", pair_off)[:, 0]
    alpha = seg_softmax(alpha, row, N)
    h_new = h + seg_sum(alpha[:, None] * v, row, N)
    cv = alpha[:, None] * _lin(sd, pre + "coord_mlp.2", _silu(_lin(sd, pre + "coord_mlp.0", v)))
    x_new = x + seg_sum(d * cv, row, N).clamp(-clamp, clamp)
    if return_pair:
        return h_new, x_new, alpha, z
    return h_new, x_new, alpha


def las_step(x, x_las, las, step, clamp):
    """egnn.py:433-449 (one geometry-regularisation step)."""
    i, j = las[0], las[1]
    dcur = x[i] - x[j]
    cur = (dcur ** 2).sum(1)
    true = ((x_las[i] - x_las[j]) ** 2).sum(1)
    force = 2 * (cur - true)[:, None] * (2 * dcur)
    delta = seg_sum(force, j, x.shape[0])
    return x + (delta * step).clamp(-clamp, clamp)


def egnn_forward(sd, pre, h, x, ctx, inter, las, x_las, lay, z0, zm, batch_id, n_layers, scale,
                 step=1e-3, capture=None):
    """MCAttEGNN.forward, egnn.py:392-466.  x, x_las are [N,3] (normalised)."""
    B = lay.B
    clamp = 10.0 / scale
    h = _lin(sd, pre + "linear_in", h)
    for i in range(n_layers):
        h, x = gcl_forward(sd, pre + "gcl_%d." % i, h, ctx[0], ctx[1], x, batch_id, B, clamp)
        if capture is not None:
            capture["gcl_%d.h" % i], capture["gcl_%d.x" % i] = h.clone(), x.clone()
        h, x, alpha = att_forward(sd, pre + "att_%d." % i, h, inter[0], inter[1], x, lay, z0, zm, batch_id, B, clamp)
        if capture is not None:
            capture["att_%d.h" % i], capture["att_%d.x" % i], capture["att_%d.alpha" % i] = h.clone(), x.clone(), alpha.clone()
        x = las_step(x, x_las, las, step, 15.0 / scale)
    h, x = gcl_forward(sd, pre + "out_layer.", h, ctx[0], ctx[1], x, batch_id, B, clamp)
    return _lin(sd, pre + "linear_out", h), x


def stack_forward(sd, pre, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                  coord_LAS, n_layers, n_iter, scale=5.0, inter_cutoff=10.0, intra_cutoff=8.0, step=1e-3,
                  capture=None):
    """EfficientMCAttModel.forward (refine_coord), att_model.py:170-246.

    X [N,1,3] normalised -- NOT mutated (the reference mutates; the host mirror does too).
    Returns (X_out [N,1,3], H_out [N,Hout])."""
    lay = Layout(batch_id, segment_id)
    p, pm, c, cm = lay.dense(H)
    z0, zm = interaction(sd, pre + "inter_layer.", p, c, pm, cm)
    z0 = z0 * zm.to(z0.dtype)[..., None]
    x = X[:, 0].clone()
    x_las = coord_LAS[:, 0]
    Hout = None
    for r in range(n_iter):
        ctx, inter = construct_edges(x[:, None], batch_id, segment_id, is_global, inter_cutoff / scale,
                                     intra_cutoff / scale)
        ctx = torch.cat([compound_edge_index, ctx], 1)
        if capture is not None and r == 0:
            capture["ctx_edges"], capture["inter_edges"] = ctx.clone(), inter.clone()
        last = r == n_iter - 1
        with torch.set_grad_enabled(last and torch.is_grad_enabled()):   # att_model.py:228-236: no_grad
            Hout, z = egnn_forward(sd, pre + "gnn.", H, x, ctx, inter, LAS_edge_index, x_las, lay, z0, zm,
                                   batch_id, n_layers, scale, step, capture if last else None)
            x = torch.where(mask[:, None], z, x)
    return x[:, None], Hout


# ------------------------------------------------------------------------------------------------
# full model (model.py:82-580)
# ------------------------------------------------------------------------------------------------
def gumbel_softmax_no_random(logits, tau=1.0, hard=False):
    y = (logits / tau).softmax(-1)
    if hard:
        idx = y.max(-1, keepdim=True)[1]
        yh = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
        return yh - y.detach() + y
    return y


def get_keepNode_tensor(xyz, radius, center):
    return torch.sqrt(((xyz - center[None]) ** 2).sum(-1)) < radius


def _to_dense(x, batch, B=None):
    B = int(batch.max()) + 1 if B is None else B
    cnt = torch.bincount(batch, minlength=B)
    off = torch.cumsum(cnt, 0) - cnt
    loc = torch.arange(batch.shape[0]) - off[batch]
    out = x.new_zeros((B, int(cnt.max())) + tuple(x.shape[1:]))
    m = torch.zeros(B, int(cnt.max()), dtype=torch.bool)
    out[batch, loc] = x
    m[batch, loc] = True
    return out, m


def _assemble(glb_c, glb_p, c_emb, c_batch, p_emb, p_batch, B):
    rows = []
    for b in range(B):
        rows += [glb_c, c_emb[c_batch == b], glb_p, p_emb[p_batch == b]]
    return torch.cat(rows, 0)


def _pocket_head(sd, cfg, data, stack_capture=None):
    """model.py:98-158: whole-protein pocket model, classifier, soft (no-noise) Gumbel centre."""
    scale = cfg["coordinate_scale"]
    c_batch, pw_batch = data["compound"].batch, data["protein_whole"].batch
    w = data["complex_whole_protein"]
    B = int(w.batch.max()) + 1
    Xw = (w.node_coords / scale)[:, None]
    Xw_las = (w.node_coords_LAS / scale)[:, None]
    c_emb = _lin(sd, "compound_linear_whole_protein", data["compound"].node_feats)
    p_emb = _lin(sd, "protein_linear_whole_protein", data["protein_whole"].node_feats)
    h0 = _lin(sd, "embedding_shrink", _assemble(sd["glb_c"], sd["glb_p"], c_emb, c_batch, p_emb, pw_batch, B))
    _, hw = stack_forward(sd, "pocket_pred_model.", Xw, h0, w.batch, w.segment, w.mask, w.is_global,
                          data["complex_whole_protein", "c2c", "complex_whole_protein"].edge_index,
                          data["complex_whole_protein", "LAS", "complex_whole_protein"].edge_index, Xw_las,
                          cfg["pocket_pred_layers"], cfg["pocket_pred_n_iter"], scale, cfg["inter_cutoff"],
                          cfg["intra_cutoff"], cfg["geometry_reg_step_size"])
    hw = _lin(sd, "embedding_enlarge", hw)
    segw = w.segment.to(torch.long)
    c_out = hw[(segw == 0) & ~w.is_global]
    p_out = hw[(segw == 1) & ~w.is_global]
    p_dense, p_mask = _to_dense(p_out, pw_batch, B)
    t = F.layer_norm(p_dense, (p_dense.shape[-1],), sd["protein_to_pocket.layernorm.weight"],
                     sd["protein_to_pocket.layernorm.bias"])
    logits = _lin(sd, "protein_to_pocket.linear2", torch.relu(_lin(sd, "protein_to_pocket.linear1", t)))[..., 0]
    logits = logits * p_mask
    xyz_dense, _ = _to_dense(data.node_xyz_whole, pw_batch, B)
    return dict(B=B, c_out=c_out, p_out=p_out, p_dense=p_dense, p_mask=p_mask, logits=logits, xyz_dense=xyz_dense)


def _soft_center(logits, p_mask, xyz_dense, tau, hard):
    pt = torch.sigmoid(logits)[..., None]
    prob = torch.cat([1.0 - pt, pt], -1).clamp(1e-6, 1 - 1e-6)
    y = gumbel_softmax_no_random(prob.log(), tau, hard)
    wgt = (y[..., 1] * p_mask)[..., None]
    return (wgt * xyz_dense).sum(1) / wgt.sum(1)


def _stage2_graph(sd, cfg, data, head, center):
    """model.py:173-300: crop a pocket of `pocket_radius` around `center`, rebuild the complex graph."""
    B = head["B"]
    c_batch, pw_batch = data["compound"].batch, data["protein_whole"].batch
    hs, xs, xl, seg, msk, glb, bid, c2c, las, pk, pkb, dm = [], [], [], [], [], [], [], [], [], [], [], []
    off, less5 = 0, 0
    for b in range(B):
        prot = data.node_xyz_whole[pw_batch == b]
        keep = get_keepNode_tensor(prot, cfg["pocket_radius"], center[b].detach())
        if keep.sum() < 5:
            keep[:100] = True
            less5 += 1
        pemb = head["p_dense"][b][head["p_mask"][b]][keep]
        cemb = head["c_out"][c_batch == b]
        pc = prot[keep]
        li = data["compound"].node_coords[c_batch == b]
        nl, npk = li.shape[0], pc.shape[0]
        n = nl + npk + 2
        hs += [sd["glb_c"], cemb, sd["glb_p"], pemb]
        z1 = torch.zeros(1, 3, dtype=pc.dtype)
        xs += [z1, li - li.mean(0, keepdim=True) + pc.mean(0, keepdim=True), z1, pc]
        xl += [z1, data["compound"].rdkit_coords[c_batch == b], z1, torch.zeros_like(pc)]
        s = torch.zeros(n)
        s[nl + 1:] = 1
        m = torch.zeros(n, dtype=torch.bool)
        m[:nl + 2] = True
        g = torch.zeros(n, dtype=torch.bool)
        g[0] = True
        g[nl + 1] = True
        seg.append(s), msk.append(m), glb.append(g), bid.append(torch.full((n,), b, dtype=torch.long))
        c2c.append(data["compound_atom_edge_list"].x[data["compound_atom_edge_list"].batch == b].T + off)
        las.append(data["LAS_edge_list"].x[data["LAS_edge_list"].batch == b].T + off)
        pk.append(pc), pkb.append(torch.full((npk,), b, dtype=torch.long))
        dm.append(torch.cdist(pc, li.to(pc.dtype)).flatten().clamp(max=10.0))
        off += n
    return dict(H=torch.cat(hs), X=torch.cat(xs).float(), XL=torch.cat(xl).float(), segment=torch.cat(seg),
                mask=torch.cat(msk), is_global=torch.cat(glb), batch=torch.cat(bid), c2c=torch.cat(c2c, 1),
                LAS=torch.cat(las, 1), pocket_xyz=torch.cat(pk), pocket_batch=torch.cat(pkb),
                dis_map=torch.cat(dm), less5=less5)


def _complex_and_heads(sd, cfg, data, g, c_batch):
    scale = cfg["coordinate_scale"]
    Xo, Ho = stack_forward(sd, "complex_model.", (g["X"] / scale)[:, None], g["H"], g["batch"], g["segment"],
                           g["mask"], g["is_global"], g["c2c"], g["LAS"], (g["XL"] / scale)[:, None],
                           cfg["mean_layers"], cfg["n_iter"], scale, cfg["inter_cutoff"], cfg["intra_cutoff"],
                           cfg["geometry_reg_step_size"])
    segl = g["segment"].to(torch.long)
    cflag = (segl == 0) & ~g["is_global"]
    pflag = (segl == 1) & ~g["is_global"]
    coords_n = Xo[cflag][:, 0]
    return Xo, Ho, cflag, pflag, coords_n


def model_forward(sd, cfg, data, stage=1):
    """IaBNet...forward(data, stage, train=False) in eval mode -> the reference's 11-tuple."""
    scale = cfg["coordinate_scale"]
    head = _pocket_head(sd, cfg, data)
    B = head["B"]
    c_batch = data["compound"].batch
    center = _soft_center(head["logits"], head["p_mask"], head["xyz_dense"], cfg["gs_tau"], cfg["gs_hard"])
    pocket_cls, _ = _to_dense(data.pocket_idx, data["protein_whole"].batch, B)
    if stage == 2:
        g = _stage2_graph(sd, cfg, data, head, center)
    else:
        cx = data["complex"]
        pemb = head["p_out"][data["pocket"].keepNode]
        g = dict(H=_assemble(sd["glb_c"], sd["glb_p"], head["c_out"], c_batch, pemb, data["pocket"].batch, B),
                 X=cx.node_coords, XL=cx.node_coords_LAS, segment=cx.segment, mask=cx.mask,
                 is_global=cx.is_global, batch=cx.batch, c2c=data["complex", "c2c", "complex"].edge_index,
                 LAS=data["complex", "LAS", "complex"].edge_index, pocket_xyz=data.node_xyz,
                 pocket_batch=data["pocket"].batch, dis_map=data.dis_map, less5=0)
    Xo, Ho, cflag, pflag, coords_n = _complex_and_heads(sd, cfg, data, g, c_batch)
    p_d, p_m = _to_dense(Ho[pflag], g["pocket_batch"], B)
    c_d, c_m = _to_dense(Ho[cflag], c_batch, B)
    xc_d, _ = _to_dense(coords_n, c_batch, B)
    xp_d, _ = _to_dense(g["pocket_xyz"] / scale, g["pocket_batch"], B)
    dist = torch.cdist(xp_d, xc_d)
    ln = lambda t: F.layer_norm(t, (t.shape[-1],), sd["layernorm.weight"], sd["layernorm.bias"])
    z = torch.einsum("bik,bjk->bijk", ln(p_d), ln(c_d))
    zm = p_m[:, :, None] & c_m[:, None, :]
    bmap = _lin(sd, "distmap_mlp.2", torch.relu(_lin(sd, "distmap_mlp.0", z)))[..., 0]
    y_pred = torch.sigmoid(bmap[zm]) * 10
    y_by = (dist[zm] * scale).clamp(0, 10)
    return (coords_n * scale, c_batch, y_pred, y_by, head["logits"], pocket_cls, head["p_mask"],
            head["xyz_dense"], center, g["dis_map"], g["less5"])


def model_inference(sd, cfg, data):
    """IaBNet...inference(data), model.py:371-580 -> (coords [sumNc,3] A, compound_batch)."""
    scale = cfg["coordinate_scale"]
    head = _pocket_head(sd, cfg, data)
    B = head["B"]
    center = torch.zeros(B, 3)
    nres = head["p_mask"].sum(1)
    for b in range(B):
        j = int(nres[b])
        lg = head["logits"][b, :j]
        hard = torch.sigmoid(lg).round().int() == 1
        if hard.sum() != 0:
            center[b] = head["xyz_dense"][b, :j][hard].mean(0)
        else:
            pt = torch.sigmoid(lg)[:, None]
            y = gumbel_softmax_no_random(torch.cat([1.0 - pt, pt], -1).log(), cfg["gs_tau"], cfg["gs_hard"])
            wgt = y[:, 1:2]
            center[b] = (wgt * head["xyz_dense"][b, :j]).sum(0) / wgt.sum(0)
    g = _stage2_graph(sd, cfg, data, head, center)
    _, _, _, _, coords_n = _complex_and_heads(sd, cfg, data, g, data["compound"].batch)
    return coords_n * scale, data["compound"].batch


def compute_loss(out, data, w=None):
    """Train-step loss, main_fabind.py:398-417 (production weights: 1,1,1,1 and 0.05 for the centre)."""
    w = dict(coord=1.0, pair=1.0, distill=1.0, cls=1.0, center=0.05, huber_delta=3.0) if w is None else w
    coords, c_batch, y_pred, y_by, logits, pocket_cls, p_mask, _, center, dis_map, _ = out
    cls = w["cls"] * F.binary_cross_entropy_with_logits(logits, pocket_cls.float()) * (p_mask.numel() / p_mask.sum())
    cen = w["center"] * F.huber_loss(center, data.coords_center, delta=w["huber_delta"])
    contact = w["pair"] * F.mse_loss(y_pred, dis_map)
    contact_by = w["pair"] * F.mse_loss(y_by, dis_map)
    distill = w["distill"] * F.mse_loss(y_by, y_pred)
    coord = w["coord"] * F.smooth_l1_loss(coords, data.coords)
    total = coord + contact + contact_by + distill + cls + cen
    return total, dict(coord=coord, contact=contact, contact_by_pred=contact_by, distill=distill, pocket_cls=cls,
                       pocket_center=cen)


def rmsd_per_complex(pred, true, batch):
    """utils/utils.py:489-490: sqrt(scatter_mean(sum((pred-true)^2, -1)))."""
    sd_ = ((pred - true) ** 2).sum(-1)
    B = int(batch.max()) + 1
    return (seg_sum(sd_, batch, B) / torch.bincount(batch, minlength=B)).sqrt()


DEFAULT_CFG = dict(coordinate_scale=5.0, mean_layers=4, n_iter=8, pocket_pred_layers=1, pocket_pred_n_iter=1,
                   inter_cutoff=10.0, intra_cutoff=8.0, pocket_radius=20.0, gs_tau=1.0, gs_hard=False,
                   geometry_reg_step_size=1e-3)
