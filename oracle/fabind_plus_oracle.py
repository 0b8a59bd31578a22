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


# ------------------------------------------------------------------------------------------------
# full model: FABindPlus.forward / .inference (FABind_plus/fabind/models/model.py:63-670), eval mode,
# use_for_radius_pred='ligand', no clustering, no confidence head
# ------------------------------------------------------------------------------------------------
from fabind_oracle import _assemble, _soft_center, _to_dense, get_keepNode_tensor, gumbel_softmax_no_random  # noqa: E402

PLUS_CFG = dict(coordinate_scale=5.0, mean_layers=5, n_iter=8, pocket_pred_layers=1, pocket_pred_n_iter=1, inter_cutoff=10.0,
                intra_cutoff=8.0, gs_tau=1.0, gs_hard=False, geometry_reg_step_size=1e-3, pocket_radius_buffer=5.0,
                min_pocket_radius=20.0, dis_map_thres=15.0)


def _pocket_head(sd, cfg, data):
    """model.py:72-146: whole-protein pocket model, classifier (LN-MLP), radius head, soft Gumbel centre."""
    scale = cfg["coordinate_scale"]
    c_batch, pw_batch = data["compound"].batch, data["protein_whole"].batch
    w = data["complex_whole_protein"]
    B = int(w.batch.max()) + 1
    c_emb = _lin(sd, "compound_linear_whole_protein", data["compound"].node_feats)
    p_emb = _lin(sd, "protein_linear_whole_protein", data["protein_whole"].node_feats)
    h0 = _lin(sd, "embedding_shrink", _assemble(sd["glb_c"], sd["glb_p"], c_emb, c_batch, p_emb, pw_batch, B))
    _, hw, _ = stack_forward(sd, "pocket_pred_model.", (w.node_coords / scale)[:, None], h0, w.batch, w.segment, w.mask,
                             w.is_global, data["complex_whole_protein", "c2c", "complex_whole_protein"].edge_index,
                             data["complex_whole_protein", "LAS", "complex_whole_protein"].edge_index,
                             (w.node_coords_LAS / scale)[:, None], cfg["pocket_pred_layers"], cfg["pocket_pred_n_iter"], scale,
                             cfg["inter_cutoff"], cfg["intra_cutoff"], cfg["geometry_reg_step_size"])
    hw = _lin(sd, "embedding_enlarge", hw)
    segw = w.segment.to(torch.long)
    c_out = hw[(segw == 0) & ~w.is_global]
    p_out = hw[(segw == 1) & ~w.is_global]
    c_dense, _ = _to_dense(c_out, c_batch, B)
    radius = torch.relu(ln_mlp(sd, "pocket_radius_head.", c_dense.sum(1), False))            # [B,1]
    p_dense, p_mask = _to_dense(p_out, pw_batch, B)
    logits = ln_mlp(sd, "protein_to_pocket.", p_dense, False)[..., 0] * p_mask
    xyz_dense, _ = _to_dense(data.node_xyz_whole, pw_batch, B)
    return dict(B=B, c_out=c_out, p_out=p_out, p_dense=p_dense, p_mask=p_mask, logits=logits, xyz_dense=xyz_dense,
                radius=radius)


def _stage2_graph(sd, cfg, data, head, center, shift_coords):
    """model.py:212-330 (and inference 518-600): radius-based crop, pocket-centred frame."""
    B = head["B"]
    c_batch, pw_batch = data["compound"].batch, data["protein_whole"].batch
    hs, xs, xl, seg, msk, glb, bid, c2c, las, pk, pkb, dm, bias = [], [], [], [], [], [], [], [], [], [], [], [], []
    off, less5 = 0, 0
    coords = data.coords.clone()
    for b in range(B):
        prot = data.node_xyz_whole[pw_batch == b]
        buf = cfg["pocket_radius_buffer"]
        r = float(head["radius"][b] * buf) if buf <= 2.0 else float(head["radius"][b] + buf)
        r = max(r, cfg["min_pocket_radius"])
        keep = get_keepNode_tensor(prot, r, center[b].detach())
        if keep.sum() < 5:
            keep[:100] = True
            less5 += 1
        pemb = head["p_dense"][b][head["p_mask"][b]][keep]
        cemb = head["c_out"][c_batch == b]
        pc = prot[keep]
        ctr = pc.mean(0, keepdim=True)
        pc = pc - ctr
        bias.append(ctr[0])
        if shift_coords:
            coords[c_batch == b] = coords[c_batch == b] - ctr
        li = data["compound"].node_coords[c_batch == b]
        nl, npk = li.shape[0], pc.shape[0]
        n = nl + npk + 2
        hs += [sd["glb_c"], cemb, sd["glb_p"], pemb]
        z1 = torch.zeros(1, 3, dtype=pc.dtype)
        xs += [z1, li - li.mean(0, keepdim=True) + pc.mean(0, keepdim=True), z1, pc]
        xl += [z1, data["compound"].rdkit_coords[c_batch == b], z1, torch.zeros_like(pc)]
        s = torch.zeros(n); s[nl + 1:] = 1
        m = torch.zeros(n, dtype=torch.bool); m[:nl + 2] = True
        g = torch.zeros(n, dtype=torch.bool); g[0] = True; g[nl + 1] = True
        seg.append(s), msk.append(m), glb.append(g), bid.append(torch.full((n,), b, dtype=torch.long))
        c2c.append(data["compound_atom_edge_list"].x[data["compound_atom_edge_list"].batch == b].T + off)
        las.append(data["LAS_edge_list"].x[data["LAS_edge_list"].batch == b].T + off)
        pk.append(pc), pkb.append(torch.full((npk,), b, dtype=torch.long))
        dm.append(torch.cdist(pc, li.to(torch.float32) - ctr).flatten().clamp(max=cfg["dis_map_thres"]))
        off += n
    return dict(H=torch.cat(hs), X=torch.cat(xs).float(), XL=torch.cat(xl).float(), segment=torch.cat(seg),
                mask=torch.cat(msk), is_global=torch.cat(glb), batch=torch.cat(bid), c2c=torch.cat(c2c, 1),
                LAS=torch.cat(las, 1), pocket_xyz=torch.cat(pk), pocket_batch=torch.cat(pkb), dis_map=torch.cat(dm),
                less5=less5, bias=torch.stack(bias), coords=coords)


def _complex(sd, cfg, g):
    scale = cfg["coordinate_scale"]
    Xo, Ho, Z = stack_forward(sd, "complex_model.", (g["X"] / scale)[:, None], g["H"], g["batch"], g["segment"], g["mask"],
                              g["is_global"], g["c2c"], g["LAS"], (g["XL"] / scale)[:, None], cfg["mean_layers"], cfg["n_iter"],
                              scale, cfg["inter_cutoff"], cfg["intra_cutoff"], cfg["geometry_reg_step_size"])
    segl = g["segment"].to(torch.long)
    cflag, pflag = (segl == 0) & ~g["is_global"], (segl == 1) & ~g["is_global"]
    return Xo, Ho, Z, cflag, pflag, Xo[cflag][:, 0]


def model_forward(sd, cfg, data, stage=2, confidence=False, stack_mlp=False, clustering=None):
    """FABindPlus.forward(data, stage, train=False), eval mode -> the reference's 13-tuple plus the shifted data.coords
    (the reference shifts `data.coords` in place and the caller's loss reads it afterwards, utils/training.py:55-60).
    confidence=True: the 7-tuple of confidence training (model.py:393-399); clustering = (eps, min_samples,
    choose_cluster_prob): the DBSCAN centre choice of model.py:147-167 (python `random`)."""
    scale, thres = cfg["coordinate_scale"], cfg["dis_map_thres"]
    head = _pocket_head(sd, cfg, data)
    B = head["B"]
    c_batch = data["compound"].batch
    center = _soft_center(head["logits"], head["p_mask"], head["xyz_dense"], cfg["gs_tau"], cfg["gs_hard"])
    pocket_cls, _ = _to_dense(data.pocket_idx, data["protein_whole"].batch, B)
    if clustering is not None:
        center = cluster_centers(head, center, *clustering)
    if stage == 1:
        cx = data["complex"]
        X = cx.node_coords.clone()
        coords = data.coords.clone()
        nl = torch.bincount(c_batch, minlength=B)
        for b in range(B):
            idx = torch.nonzero(cx.batch == b)[:, 0]
            n = int(nl[b])
            lig, poc = idx[1:n + 1], idx[n + 2:]
            X[lig] = X[lig] - X[lig].mean(0)
            X[poc] = X[poc] - data.pocket_residue_center[b][None]
            coords[c_batch == b] = coords[c_batch == b] - data.pocket_residue_center[b][None]
        pemb = head["p_out"][data["pocket"].keepNode]
        g = dict(H=_assemble(sd["glb_c"], sd["glb_p"], head["c_out"], c_batch, pemb, data["pocket"].batch, B), X=X,
                 XL=cx.node_coords_LAS, segment=cx.segment, mask=cx.mask, is_global=cx.is_global, batch=cx.batch,
                 c2c=data["complex", "c2c", "complex"].edge_index, LAS=data["complex", "LAS", "complex"].edge_index,
                 pocket_xyz=data.node_xyz, pocket_batch=data["pocket"].batch, dis_map=data.dis_map, less5=0,
                 bias=torch.zeros(B, 3), coords=coords)
    else:
        g = _stage2_graph(sd, cfg, data, head, center, shift_coords=True)
    Xo, Ho, Z, cflag, pflag, coords_n = _complex(sd, cfg, g)
    if confidence:
        pooled = torch.zeros(B, Ho.shape[1]).index_add_(0, g["batch"], Ho)
        if stack_mlp:
            pooled = torch.relu(_mlp4conf(sd, "ranking_mlp_pre.", pooled))
        score = _mlp4conf(sd, "ranking_score_mlp.", pooled)[:, 0]
        return (coords_n * scale, c_batch, head["logits"], head["p_mask"], g["less5"], score, g["bias"]), g["coords"]
    _, p_m = _to_dense(Ho[pflag], g["pocket_batch"], B)
    _, c_m = _to_dense(Ho[cflag], c_batch, B)
    xc_d, _ = _to_dense(coords_n, c_batch, B)
    xp_d, _ = _to_dense(g["pocket_xyz"] / scale, g["pocket_batch"], B)
    zm = p_m[:, :, None] & c_m[:, None, :]
    bmap = ln_mlp(sd, "distmap_mlp.", Z[:, 1:, 1:], False)[..., 0]
    y_pred = torch.sigmoid(bmap[zm]) * thres
    y_by = (torch.cdist(xp_d, xc_d)[zm] * scale).clamp(0, thres)
    return (coords_n * scale, c_batch, y_pred, y_by, head["logits"], pocket_cls, head["p_mask"], head["xyz_dense"], center,
            g["dis_map"], g["less5"], head["radius"], g["bias"]), g["coords"]


def _mlp4conf(sd, pre, x):
    """MLP4Confidence (model_utils.py:77-98), eval: [LN] -> linear1 -> relu -> linear2."""
    if pre + "layernorm.weight" in sd:
        w = sd[pre + "layernorm.weight"]
        x = F.layer_norm(x, (w.shape[0],), w, sd[pre + "layernorm.bias"], 1e-5)
    return _lin(sd, pre + "linear2", torch.relu(_lin(sd, pre + "linear1", x)))


def cluster_centers(head, center, eps, min_samples, choose_prob):
    """model.py:147-167 / 487-507: DBSCAN (scikit-learn, the reference's own dependency) over the residues predicted to
    be pocket; python's `random` decides which cluster / whether to keep the soft centre."""
    import random
    from sklearn.cluster import DBSCAN
    db = DBSCAN(eps=eps, min_samples=min_samples)
    centers = torch.zeros_like(center)
    prob = torch.sigmoid(head["logits"])
    for i in range(center.shape[0]):
        pp = prob[i]
        sel = pp > 0.5
        if int(sel.sum()) < 50:
            top = torch.argsort(pp)[-50:]
            sel = torch.zeros_like(sel)
            sel[top] = True
        pts = head["xyz_dense"][i][sel].numpy()
        cl = db.fit(pts)
        cid = random.randint(0, cl.labels_.max())
        if random.random() < choose_prob:
            centers[i] = torch.tensor(pts[cl.labels_ == cid].mean(axis=0))
        else:
            centers[i] = center[i]
    return centers


def model_inference(sd, cfg, data, confidence=False, stack_mlp=False, clustering=None):
    """FABindPlus.inference(data), model.py:403-670 -> (coords in the whole-protein frame [sum Nc, 3] A, compound_batch
    [, confidence score per complex]).  clustering = (eps, min_samples, choose_cluster_prob) or None."""
    scale = cfg["coordinate_scale"]
    head = _pocket_head(sd, cfg, data)
    center = _soft_center(head["logits"], head["p_mask"], head["xyz_dense"], cfg["gs_tau"], cfg["gs_hard"])
    if clustering is not None:
        center = cluster_centers(head, center, *clustering)
    g = _stage2_graph(sd, cfg, data, head, center, shift_coords=False)
    _, Ho, _, _, _, coords_n = _complex(sd, cfg, g)
    c_batch = data["compound"].batch
    coords = coords_n * scale + g["bias"][c_batch]
    if not confidence:
        return coords, c_batch
    pooled = torch.zeros(head["B"], Ho.shape[1]).index_add_(0, g["batch"], Ho)
    if stack_mlp:
        pooled = torch.relu(_mlp4conf(sd, "ranking_mlp_pre.", pooled))
    return coords, c_batch, _mlp4conf(sd, "ranking_score_mlp.", pooled)[:, 0]


# ------------------------------------------------------------------------------------------------
# train-step loss of FABind+ (FABind_plus/fabind/utils/training.py:55-97, utils/permutation_loss.py:4-38,
# main_fabind.py:151-158): permutation-invariant SmoothL1 over precomputed ligand automorphisms + 6 further terms
# ------------------------------------------------------------------------------------------------
def best_isomorphism_index(pos_x, pos_y, num_atoms, isomorphisms):
    """permutation_loss.py:4-33: per ligand, the automorphism (index list) minimising the mean SmoothL1 to the target."""
    idx, pre = [], 0
    with torch.no_grad():
        for n, isos in zip(num_atoms, isomorphisms):
            isos = [torch.as_tensor(i, dtype=torch.long) for i in isos]
            if len(isos) == 1:
                idx.append(isos[0] + pre)
            else:
                x, y = pos_x[pre:pre + n], pos_y[pre:pre + n]
                losses = torch.stack([F.smooth_l1_loss(x[i], y, reduction="none").mean() for i in isos])
                idx.append(isos[int(torch.argmin(losses))] + pre)
            pre += n
    return torch.cat(idx)


def compute_loss(out, coords_true, data, num_atoms, isomorphisms, ligand_radius, w=None):
    """-> (total, dict of the 7 terms).  `coords_true` = data.coords AFTER forward (pocket frame)."""
    w = dict(coord=1.5, pair=1.0, distill=1.0, cls=1.0, center=0.05, radius=0.05, huber_delta=3.0) if w is None else w
    coords, c_batch, y_pred, y_by, logits, pocket_cls, p_mask, _, center, dis_map, _, radius, _ = out
    cls = w["cls"] * F.binary_cross_entropy_with_logits(logits, pocket_cls.float()) * (p_mask.numel() / p_mask.sum())
    cen = w["center"] * F.huber_loss(center, data.coords_center, delta=w["huber_delta"])
    contact = w["pair"] * F.mse_loss(y_pred, dis_map)
    contact_by = w["pair"] * F.mse_loss(y_by, dis_map)
    distill = w["distill"] * F.mse_loss(y_by, y_pred)
    rad = w["radius"] * F.huber_loss(radius[:, 0], ligand_radius.to(radius.dtype), delta=w["huber_delta"])
    new_idx = best_isomorphism_index(coords, coords_true, num_atoms, isomorphisms)
    coord = w["coord"] * F.smooth_l1_loss(coords[new_idx], coords_true, reduction="none").mean()
    total = coord + contact + contact_by + distill + cls + rad + cen
    return total, dict(coord=coord, contact=contact, contact_by_pred=contact_by, distill=distill, pocket_cls=cls,
                       pocket_radius=rad, pocket_center=cen)
