"""FABindPlus / get_model on the HIP engine (reference FABind_plus/fabind/models/model.py:13-675).

Same constructor, parameter names and return tuples as the reference: forward -> 13-tuple (7-tuple with
`args.confidence_training`), inference -> 2-tuple (3-tuple with the ranking score); the per-sample python loops
(model.py:84-97, 176-197, 225-330) are vectorised index arithmetic.  forward is differentiable (autograd through the HIP
stack); inference covers eval and the dropout-sampling mode with the DBSCAN centre choice."""
import numpy as np
import torch
import torch.nn as nn

from ... import kernels as K
from ... import ops
from ...models.model import IaBNet_mean_and_pocket_prediction_cls_coords_dependent as _V1
from ...models.model import _offsets
from ...utils.utils import gumbel_softmax_no_random  # noqa: F401
from .. import engine as pengine
from .att_model import EfficientMCAttModel
from .model_utils import MLP, MLP4Confidence


def dbscan_labels(pts, valid, eps, min_samples):
    """Batched DBSCAN with scikit-learn's labelling (sklearn.cluster.DBSCAN, euclidean): pts [B,S,3], valid [B,S] ->
    labels [B,S] int64 (-1 = noise / padding).  sklearn's rule, restated: a point is core when >= min_samples points
    (itself included) lie within eps; clusters are the connected components of core points, numbered by their smallest
    core index; a non-core point within eps of core points joins the lowest-numbered such cluster (clusters are expanded
    one after the other, the first to reach a border point keeps it).  Distances in float64 like sklearn's tree queries.
    Components by min-label propagation with pointer jumping on the [S,S] adjacency (S = a few hundred residues)."""
    B, S, _ = pts.shape
    dev = pts.device
    x = pts.double()
    d2 = (x[:, :, None, :] - x[:, None, :, :]).pow(2).sum(-1)
    adj = (d2 <= eps * eps) & valid[:, :, None] & valid[:, None, :]
    core = (adj.sum(-1) >= min_samples) & valid
    cadj = adj & core[:, :, None] & core[:, None, :]
    BIG = S
    idx = torch.arange(S, device=dev)
    lab = torch.where(core, idx[None, :].expand(B, S), torch.full((B, S), BIG, device=dev))
    while True:
        for _ in range(4):
            nb = torch.where(cadj, lab[:, None, :].expand(B, S, S), torch.full((), BIG, device=dev)).min(-1)[0]
            new = torch.minimum(lab, nb)
            new = torch.where(core, torch.gather(torch.cat([new, torch.full((B, 1), BIG, device=dev)], 1), 1, new), new)   # jump
            changed = (new != lab).any()
            lab = new
        if not bool(changed):
            break
    # rank the component roots (root = smallest core index of the component) -> cluster numbers 0..k-1
    is_root = core & (lab == idx[None, :])
    rank = torch.cumsum(is_root.long(), 1) - 1
    clus = torch.where(core, torch.gather(rank, 1, lab.clamp(max=S - 1)), torch.full((B, S), -1, device=dev))
    # border points: the lowest-numbered cluster among the core points within eps
    cand = torch.where(adj & core[:, None, :], clus[:, None, :].expand(B, S, S), torch.full((), BIG, device=dev)).min(-1)[0]
    border = torch.where(cand < BIG, cand, torch.full((B, S), -1, device=dev))
    return torch.where(core, clus, torch.where(valid, border, torch.full((B, S), -1, device=dev)))


class FABindPlus(nn.Module):
    def __init__(self, args, embedding_channels=128, pocket_pred_embedding_channels=128):
        super().__init__()
        if args.use_for_radius_pred != "ligand" or args.force_fix_radius:
            raise NotImplementedError("only --use-for-radius-pred ligand (production) is built")
        self.args = args
        self.coordinate_scale = args.coordinate_scale
        self.normalize_coord = lambda x: x / self.coordinate_scale
        self.unnormalize_coord = lambda x: x * self.coordinate_scale
        self.glb_c = nn.Parameter(torch.ones(1, embedding_channels))
        self.glb_p = nn.Parameter(torch.ones(1, embedding_channels))
        self.protein_linear_whole_protein = nn.Linear(1280, embedding_channels)
        self.compound_linear_whole_protein = nn.Linear(56, embedding_channels)
        self.embedding_shrink = nn.Linear(embedding_channels, pocket_pred_embedding_channels)
        self.embedding_enlarge = nn.Linear(pocket_pred_embedding_channels, embedding_channels)
        n_channel = 1
        self.pocket_pred_model = EfficientMCAttModel(
            args, pocket_pred_embedding_channels, pocket_pred_embedding_channels, n_channel, n_edge_feats=0,
            n_layers=args.pocket_pred_layers, n_iter=args.pocket_pred_n_iter, inter_cutoff=args.inter_cutoff,
            intra_cutoff=args.intra_cutoff, normalize_coord=self.normalize_coord, unnormalize_coord=self.unnormalize_coord)
        n = args.mlp_hidden_scale
        self.pocket_radius_head = MLP(args, embedding_channels=embedding_channels, n=n, out_channels=1)
        self.protein_to_pocket = MLP(args, embedding_channels=embedding_channels, n=n, out_channels=1)
        self.complex_model = EfficientMCAttModel(
            args, embedding_channels, embedding_channels, n_channel, n_edge_feats=0, n_layers=args.mean_layers,
            n_iter=args.n_iter, inter_cutoff=args.inter_cutoff, intra_cutoff=args.intra_cutoff,
            normalize_coord=self.normalize_coord, unnormalize_coord=self.unnormalize_coord)
        self.distmap_mlp = MLP(args, embedding_channels=embedding_channels, n=n, out_channels=1)
        for lin in (self.protein_linear_whole_protein, self.compound_linear_whole_protein, self.embedding_shrink,
                    self.embedding_enlarge):
            torch.nn.init.xavier_uniform_(lin.weight, gain=0.001)
        # sampling mode (inference_sampling_fabind.py:148-187): ranking head on the summed complex embedding and, with
        # --use-clustering, a host-side DBSCAN choice of the pocket centre (model.py:50-61, 147-167)
        self.confidence_training = getattr(args, 'confidence_training', False)
        if self.confidence_training:
            if args.stack_mlp:
                self.ranking_mlp_pre = MLP4Confidence(args, embedding_channels=embedding_channels,
                                                      n=args.confidence_mlp_hidden_scale, out_channels=embedding_channels)
            self.ranking_score_mlp = MLP4Confidence(args, embedding_channels=embedding_channels,
                                                    n=args.confidence_mlp_hidden_scale, out_channels=1)
        if args.use_clustering:
            from sklearn.cluster import DBSCAN
            self.dbscan_module = DBSCAN(eps=args.dbscan_eps, min_samples=args.dbscan_min_samples)

    def _pd(self):
        return float(self.args.dropout) if self.training else 0.0

    def _cluster_centers(self, head, center):
        """DBSCAN over the residues predicted to be pocket (reference: host-side scikit-learn per complex, model.py:147-167).
        The clustering of ALL complexes runs on the device (`dbscan_labels`: same labels as sklearn, see its docstring);
        python's `random` is consumed exactly as the reference does (one randint + one random per complex), which needs
        one small read-back (the number of clusters per complex)."""
        import random
        prob = head['logits'].sigmoid()
        sel = prob > 0.5
        # too few predicted pocket residues (rare with trained weights, every complex with random ones): the 50 most probable instead
        # (model.py:152-154) -- for all such complexes at once; a per-complex loop made a synchronising indexed assignment per complex
        # and pose batch (1,280 queue drains per sampling step of the bench)
        few = sel.sum(1) < 50
        top = torch.argsort(prob, dim=1)[:, -50:]
        sel = torch.where(few[:, None], torch.zeros_like(sel).scatter_(1, top, True), sel)
        B, dev = center.shape[0], center.device
        n_sel = sel.sum(1)
        S = int(n_sel.max())
        order = torch.argsort((~sel).to(torch.int8), dim=1, stable=True)[:, :S]       # selected residues first, index order kept
        valid = torch.arange(S, device=dev)[None, :] < n_sel[:, None]
        pts = torch.gather(head['xyz'].detach().float(), 1, order[:, :, None].expand(-1, -1, 3))
        labels = dbscan_labels(pts, valid, float(self.args.dbscan_eps), int(self.args.dbscan_min_samples))
        n_clusters = (labels.max(1)[0] + 1).tolist()
        cid, use = [], []
        for i in range(B):
            cid.append(random.randint(0, n_clusters[i] - 1))                    # ValueError without any cluster, like the reference
            use.append(random.random() < self.args.choose_cluster_prob)
        cid_d = torch.tensor(cid, device=dev)
        member = (labels == cid_d[:, None]) & valid
        picked = (member[:, :, None] * pts).sum(1) / member.sum(1, keepdim=True).clamp(min=1)
        return torch.where(torch.tensor(use, device=dev)[:, None], picked.to(center.dtype), center)

    def _confidence(self, Ho, batch, B):
        pooled = torch.zeros(B, Ho.shape[1], dtype=torch.float32, device=Ho.device).index_add_(0, batch, Ho.float())
        cd = float(self.args.confidence_dropout) if (self.ranking_score_mlp.training) else 0.0
        if self.args.stack_mlp:
            pooled = pengine.mlp_module(self.ranking_mlp_pre, pooled, pdrop=cd).relu()
        return pengine.mlp_module(self.ranking_score_mlp, pooled, pdrop=cd).squeeze(-1)

    _lin = _V1._lin
    _assemble = _V1._assemble
    _assemble_index = staticmethod(_V1._assemble_index)
    _soft_center = _V1._soft_center
    _pair_lists = staticmethod(_V1._pair_lists)

    # ---- whole-protein pocket model, classifier, radius head (model.py:72-146) -------------------------------------
    def _pocket_head(self, data):
        w = data['complex_whole_protein']
        c_emb = self._lin(self.compound_linear_whole_protein, data['compound'].node_feats)
        p_emb = self._lin(self.protein_linear_whole_protein, data['protein_whole'].node_feats)
        h0, cf, pf = self._assemble(w.segment, w.is_global, c_emb, p_emb)
        h0 = self._lin(self.embedding_shrink, h0)
        Xw = self.normalize_coord(w.node_coords.unsqueeze(-2)).float()
        Xl = self.normalize_coord(w.node_coords_LAS.unsqueeze(-2)).float()
        # (the padded [B, Lmax] bookkeeping reads two sizes back: before the pocket model is queued)
        cb, pb = data['compound'].batch, data['protein_whole'].batch
        B = int(pb[-1].item()) + 1
        cnt = torch.bincount(pb, minlength=B)
        loc = torch.arange(pb.shape[0], device=pb.device) - _offsets(cnt)[pb]
        Lmax = int(cnt.max().item())
        mask = torch.zeros(B, Lmax, dtype=torch.bool, device=pb.device)
        mask[pb, loc] = True
        xyz = torch.zeros(B, Lmax, 3, dtype=data.node_xyz_whole.dtype, device=pb.device)
        xyz[pb, loc] = data.node_xyz_whole
        _, hw, _ = self.pocket_pred_model(
            Xw, h0, batch_id=w.batch, segment_id=w.segment, mask=w.mask, is_global=w.is_global,
            compound_edge_index=data['complex_whole_protein', 'c2c', 'complex_whole_protein'].edge_index,
            LAS_edge_index=data['complex_whole_protein', 'LAS', 'complex_whole_protein'].edge_index,
            batched_complex_coord_LAS=Xl, LAS_mask=None, pair="none")
        hw = self._lin(self.embedding_enlarge, hw)
        c_out, p_out = ops.take_unique_rows(hw, cf), ops.take_unique_rows(hw, pf)
        csum = torch.zeros(B, c_out.shape[1], dtype=torch.float32, device=hw.device).index_add_(0, cb, c_out.float())
        radius = pengine.mlp_module(self.pocket_radius_head, csum, pdrop=self._pd()).relu()                      # [B,1]
        logits_flat = pengine.mlp_module(self.protein_to_pocket, p_out, pdrop=self._pd()).squeeze(-1)            # [sum L]
        logits = torch.zeros(B, Lmax, dtype=logits_flat.dtype, device=pb.device).index_put((pb, loc), logits_flat)
        return dict(B=B, c_out=c_out, p_out=p_out, logits=logits, mask=mask, xyz=xyz, pb=pb, loc=loc, cnt=cnt, radius=radius)

    # ---- radius crop around the predicted centre, pocket-centred frame (model.py:212-330) --------------------------
    def _stage2(self, data, head, center, shift_coords):
        g = self._stage2_nograd(data, head, center, shift_coords)
        g['H'], _, _ = self._assemble(g['segment'], g['is_global'], head['c_out'], ops.take_unique_rows(head['p_out'], torch.nonzero(g.pop('keep')).squeeze(1)))
        return g

    @torch.no_grad()
    def _stage2_nograd(self, data, head, center, shift_coords):
        a, dev = self.args, center.device
        pb, B, cb = head['pb'], head['B'], data['compound'].batch
        r = head['radius'][:, 0].detach()
        r = r * a.pocket_radius_buffer if a.pocket_radius_buffer <= 2.0 else r + a.pocket_radius_buffer
        r = r.clamp(min=a.min_pocket_radius)
        d = (data.node_xyz_whole.float() - center.detach()[pb]).norm(dim=-1)
        keep = d < r[pb]
        kcnt = torch.bincount(pb[keep], minlength=B)
        bad = kcnt < 5
        less5 = int(bad.sum().item())
        if less5:
            keep = keep | (bad[pb] & (head['loc'] < 100))
            kcnt = torch.bincount(pb[keep], minlength=B)
        ncnt = torch.bincount(cb, minlength=B)
        n = ncnt + kcnt + 2
        off = _offsets(n)
        N = int(n.sum().item())
        pocket_batch = pb[keep]
        lig_pos = off[cb] + 1 + (torch.arange(cb.shape[0], device=dev) - _offsets(ncnt)[cb])
        prot_pos = off[pocket_batch] + ncnt[pocket_batch] + 2 + \
            (torch.arange(pocket_batch.shape[0], device=dev) - _offsets(kcnt)[pocket_batch])
        segment = torch.zeros(N, dtype=torch.bool, device=dev)
        segment[prot_pos] = True
        segment[off + ncnt + 1] = True
        is_global = torch.zeros(N, dtype=torch.bool, device=dev)
        is_global[off] = True
        is_global[off + ncnt + 1] = True
        mask = ~segment | is_global
        batch = torch.repeat_interleave(torch.arange(B, device=dev), n)
        pocket_xyz = data.node_xyz_whole[keep].float()
        bias = ops.sum_sorted_segments(pocket_xyz, kcnt) / kcnt[:, None]   # pocket centre (fixed-order sums, like the two below)
        pocket_xyz = pocket_xyz - bias[pocket_batch]
        if shift_coords:                                            # the caller's loss reads data.coords after forward
            data.coords = data.coords - bias[cb].to(data.coords.dtype)
        li = data['compound'].node_coords.float()
        mean_l = ops.sum_sorted_segments(li, ncnt) / ncnt[:, None]
        mean_p = ops.sum_sorted_segments(pocket_xyz, kcnt) / kcnt[:, None]
        X = torch.zeros(N, 3, device=dev)
        X[lig_pos] = li - mean_l[cb] + mean_p[cb]
        X[prot_pos] = pocket_xyz
        XL = torch.zeros(N, 3, device=dev)
        XL[lig_pos] = data['compound'].rdkit_coords.float()
        el, ll = data['compound_atom_edge_list'], data['LAS_edge_list']
        c2c = (el.x + off[el.batch][:, None]).t().contiguous().long()
        las = (ll.x + off[ll.batch][:, None]).t().contiguous().long()
        pi, ci = self._pair_lists(pocket_batch, cb, kcnt, ncnt)
        dis_map = (pocket_xyz[pi] - (li - bias[cb])[ci]).norm(dim=-1).clamp(max=a.dis_map_thres)
        cx = data['complex']
        cx.node_coords, cx.node_coords_LAS, cx.segment, cx.mask, cx.is_global = X, XL, segment, mask, is_global
        data['complex', 'c2c', 'complex'].edge_index = c2c
        data['complex', 'LAS', 'complex'].edge_index = las
        return dict(keep=keep, X=X, XL=XL, segment=segment, mask=mask, is_global=is_global, batch=batch, c2c=c2c, LAS=las,
                    pocket_xyz=pocket_xyz, pocket_batch=pocket_batch, dis_map=dis_map, less5=less5, pairs=(pi, ci), bias=bias)

    def _stage1(self, data, head):
        """Teacher-forced pocket, moved into the pocket frame IN PLACE like the reference (model.py:170-201)."""
        with torch.no_grad():
            g = self._stage1_nograd(data, head)
        cx = data['complex']
        g['H'], _, _ = self._assemble(cx.segment, cx.is_global, head['c_out'], ops.take_unique_rows(head['p_out'], torch.nonzero(data['pocket'].keepNode).squeeze(1)))
        return g

    def _stage1_nograd(self, data, head):
        cx, cb = data['complex'], data['compound'].batch
        B, dev = head['B'], cb.device
        seg1 = cx.segment.bool() if not cx.segment.is_floating_point() else cx.segment > 0.5
        lig = ~seg1 & ~cx.is_global
        poc = seg1 & ~cx.is_global
        X = cx.node_coords.clone()
        ncnt = torch.bincount(cb, minlength=B)
        mean_l = torch.zeros(B, 3, device=dev, dtype=X.dtype).index_add_(0, cb, X[lig]) / ncnt[:, None]
        X[lig] = X[lig] - mean_l[cx.batch[lig]]
        X[poc] = X[poc] - data.pocket_residue_center[cx.batch[poc]].to(X.dtype)
        cx.node_coords = X
        data.coords = data.coords - data.pocket_residue_center[cb].to(data.coords.dtype)
        return dict(X=X, XL=cx.node_coords_LAS, segment=cx.segment, mask=cx.mask, is_global=cx.is_global, batch=cx.batch,
                    c2c=data['complex', 'c2c', 'complex'].edge_index, LAS=data['complex', 'LAS', 'complex'].edge_index,
                    pocket_xyz=data.node_xyz, pocket_batch=data['pocket'].batch, dis_map=data.dis_map, less5=0,
                    bias=torch.zeros(B, 3, device=dev), B=B)

    def _complex(self, g, pair):
        Xn = self.normalize_coord(g['X'].unsqueeze(-2)).float()
        Xl = self.normalize_coord(g['XL'].unsqueeze(-2)).float()
        # (the ligand rows are read back BEFORE the complex model is queued: boolean-mask indexing of its output would read the count
        #  back behind the whole forward pass)
        seg1 = g['segment'].bool() if not g['segment'].is_floating_point() else g['segment'] > 0.5
        cflag, pflag = ~seg1 & ~g['is_global'], seg1 & ~g['is_global']
        cidx = torch.nonzero(cflag).squeeze(1)
        Xo, Ho, Z = self.complex_model(Xn, g['H'], batch_id=g['batch'], segment_id=g['segment'], mask=g['mask'],
                                       is_global=g['is_global'], compound_edge_index=g['c2c'], LAS_edge_index=g['LAS'],
                                       batched_complex_coord_LAS=Xl, LAS_mask=None, pair=pair)
        return Ho, Z, cflag, pflag, ops.take_unique_rows(Xo, cidx).squeeze(-2)

    def _dist_heads(self, data, g, Z, coords_n):
        """distmap_mlp on the threaded pair embedding without its global rows (model.py:379-388) + coordinate distances."""
        z, pairs, _ = Z
        sel = (pairs.i >= 1) & (pairs.j >= 1)                                   # z[:, 1:, 1:] in (b, i, j) order
        # z[sel] without the read-back of boolean indexing (it would wait for the whole forward pass): the number of selected pairs is
        # known on the host; their positions come from a running count, the unselected ones are parked in one spare slot
        n_sel = pairs.n_inner
        slot = torch.where(sel, torch.cumsum(sel, 0) - 1, torch.full_like(pairs.i, n_sel))
        idx = torch.empty(n_sel + 1, dtype=torch.int64, device=z.device).scatter_(0, slot, torch.arange(sel.shape[0], device=z.device))[:n_sel]
        zz = ops.take_unique_rows(z, idx)
        m = self.distmap_mlp
        wd = ops.mm_dtype()
        y = pengine.ln_rows(zz, m.layernorm.weight.float(), m.layernorm.bias.float(), ops.act_dtype())
        pd = self._pd()
        if pd > 0.0 and ops.needs_grad(y) and not (pengine.EPI_DROP_GRAD and pengine.ROWDOT_DROP_GRAD):
            t = pengine._drop(ops.linear(y, m.linear1.weight.to(wd).contiguous(), m.linear1.bias, act_epi=K.ACT_RELU), pd)
            part = (t * m.linear2.weight[0].float()).sum(1, keepdim=True)
        else:       # (under autograd with dropout: the row-dot node with the dropout inside the GEMM epilogue, ops._LinearRowdotDrop)
            part = ops.linear_rowdot(y, m.linear1.weight.to(wd).contiguous(), m.linear1.bias,
                                     m.linear2.weight[0].float().contiguous(), act_epi=K.ACT_RELU, p_drop=pd)
        thres = self.args.dis_map_thres
        y_pred = (part.sum(1) + m.linear2.bias).sigmoid() * thres
        cb = data['compound'].batch
        pairs_ = g.get('pairs')
        if pairs_ is None:
            B = g['B'] if 'B' in g else int(cb[-1].item()) + 1
            pairs_ = self._pair_lists(g['pocket_batch'], cb, torch.bincount(g['pocket_batch'], minlength=B), torch.bincount(cb, minlength=B))
        pi, ci = pairs_
        xp = self.normalize_coord(g['pocket_xyz']).float()
        y_by = self.unnormalize_coord((xp[pi] - ops.take_rows_few(coords_n, ci)).norm(dim=-1)).clamp(0, thres)
        return y_pred, y_by

    # ---- reference API ---------------------------------------------------------------------------------------------
    def forward(self, data, stage=2, train=False):
        """model.py:63-401.  train=True (or stage 1): teacher forcing with the native pocket; module.train(): Gumbel noise
        on the centre, dropout, random n_iter.  Differentiable (autograd through the HIP stack)."""
        cb = data['compound'].batch
        head = self._pocket_head(data)
        center = self._soft_center(head['logits'], head['mask'], head['xyz'], noise=self.pocket_pred_model.training)
        if self.args.use_clustering:                  # model.py:147-167: the DBSCAN centre replaces the soft one (a constant)
            with torch.no_grad():
                center = self._cluster_centers(head, center)
        pocket_cls = torch.zeros_like(head['mask'], dtype=data.pocket_idx.dtype)
        pocket_cls[head['pb'], head['loc']] = data.pocket_idx
        if train or stage == 1:
            g = self._stage1(data, head)
        else:
            c2 = center
            if self.args.train_pred_pocket_noise and train:
                c2 = center + self.args.train_pred_pocket_noise * (2 * torch.rand_like(center) - 1)
            g = self._stage2(data, head, c2, shift_coords=True)
        if self.confidence_training:                  # train_confidence.py: the 7-tuple with the ranking score (model.py:393-399)
            Ho, _, _, _, coords_n = self._complex(g, "none")
            return (self.unnormalize_coord(coords_n), cb, head['logits'] * head['mask'], head['mask'], g['less5'],
                    self._confidence(Ho, g['batch'], head['B']), g['bias'])
        if g.get('pairs') is None:        # (the pair lists read a size back: built before the complex model is queued)
            B_ = g['B'] if 'B' in g else int(data['compound'].batch[-1].item()) + 1
            cb_ = data['compound'].batch
            g['pairs'] = self._pair_lists(g['pocket_batch'], cb_, torch.bincount(g['pocket_batch'], minlength=B_), torch.bincount(cb_, minlength=B_))
        Ho, Z, cflag, pflag, coords_n = self._complex(g, "ragged")
        y_pred, y_by = self._dist_heads(data, g, Z, coords_n)
        return (self.unnormalize_coord(coords_n), cb, y_pred, y_by, head['logits'] * head['mask'], pocket_cls, head['mask'],
                head['xyz'], center, g['dis_map'], g['less5'], head['radius'], g['bias'])

    @torch.no_grad()
    def inference(self, data):
        """model.py:403-670.  In train mode (`--infer-dropout` sampling, inference_sampling_fabind.py:148-152) every dropout
        of the stack and heads is active; with `confidence_training` the 3-tuple carries the ranking score."""
        head = self._pocket_head(data)
        center = self._soft_center(head['logits'], head['mask'], head['xyz'], noise=False)
        if self.args.use_clustering:
            center = self._cluster_centers(head, center)
        g = self._stage2(data, head, center, shift_coords=False)
        data['complex'].batch = g['batch']
        Ho, _, _, _, coords_n = self._complex(g, "none")
        cb = data['compound'].batch
        coords = self.unnormalize_coord(coords_n) + g['bias'][cb]            # back in the whole-protein frame
        if self.confidence_training:
            return coords, cb, self._confidence(Ho, g['batch'], head['B'])
        return coords, cb


def best_isomorphism_index(pos_x, pos_y, num_atoms, isomorphisms):
    """Per ligand, the precomputed automorphism (index list) with the smallest mean SmoothL1 to the target
    (reference utils/permutation_loss.py:4-33).  No gradient.
    All ligands at once: the automorphism lists are flattened on the host (numpy) and uploaded in ONE asynchronous copy; the per-
    automorphism losses, the per-ligand minimum (first automorphism reaching it, like argmin) and the chosen index lists are segment
    operations on the device -- no host round trip.  (The per-ligand loop made two synchronising calls per ligand -- a pageable upload
    of its lists and int(argmin) -- each behind the whole forward pass: 128 queue drains per training step at B = 64.)"""
    import numpy as np
    import torch.nn.functional as F
    from ...param_pack import upload
    dev = pos_x.device
    ns = [int(n) for n in (num_atoms.tolist() if torch.is_tensor(num_atoms) else num_atoms)]
    src, tgt, iso_id, iso_lig, iso_n, iso_off, atom_lig, atom_p = [], [], [], [], [], [], [], []
    pre = n_iso = n_flat = 0
    for b, (n, isos) in enumerate(zip(ns, isomorphisms)):
        a = np.asarray(isos, dtype=np.int64).reshape(-1, n)
        K_ = a.shape[0]
        src.append(a.reshape(-1) + pre)
        tgt.append(np.tile(np.arange(n, dtype=np.int64) + pre, K_))
        iso_id.append(np.repeat(np.arange(K_, dtype=np.int64) + n_iso, n))
        iso_lig.append(np.full(K_, b, dtype=np.int64))
        iso_n.append(np.full(K_, n, dtype=np.int64))
        iso_off.append(n_flat + np.arange(K_, dtype=np.int64) * n)
        atom_lig.append(np.full(n, b, dtype=np.int64))
        atom_p.append(np.arange(n, dtype=np.int64))
        pre, n_iso, n_flat = pre + n, n_iso + K_, n_flat + K_ * n
    parts = [np.concatenate(v) for v in (src, tgt, iso_id, iso_lig, iso_n, iso_off, atom_lig, atom_p)]
    sizes = [len(v) for v in parts]
    flat = upload(np.concatenate(parts), dev, torch.int64)
    src_d, tgt_d, iso_id_d, iso_lig_d, iso_n_d, iso_off_d, atom_lig_d, atom_p_d = torch.split(flat, sizes)
    B = len(ns)
    with torch.no_grad():
        le = F.smooth_l1_loss(pos_x.detach()[src_d], pos_y.detach()[tgt_d], reduction="none").mean(1)            # [sum K n]
        iso_loss = torch.zeros(n_iso, dtype=le.dtype, device=dev).index_add_(0, iso_id_d, le) / iso_n_d.to(le.dtype)
        lig_min = torch.full((B,), float("inf"), dtype=le.dtype, device=dev).scatter_reduce_(0, iso_lig_d, iso_loss, "amin")
        ar = torch.arange(n_iso, device=dev)
        cand = torch.where(iso_loss <= lig_min[iso_lig_d], ar, torch.full_like(ar, n_iso))
        kstar = torch.full((B,), n_iso, dtype=torch.int64, device=dev).scatter_reduce_(0, iso_lig_d, cand, "amin")   # first minimiser
        return src_d[iso_off_d[kstar[atom_lig_d]] + atom_p_d]


def compute_loss(out, data, args=None):
    """FABind+ train-step loss (reference utils/training.py:55-97 with the criteria of main_fabind.py:151-158) as a function
    of the forward's 13-tuple; reads `data.coords` AFTER forward (pocket frame), `data.coords_center`, `data.ligand_radius`,
    and -- with --permutation-invariant -- `data.num_atoms` / `data.isomorphisms`.  Returns (loss, dict of the 7 terms)."""
    import torch.nn.functional as F
    w = dict(coord=1.5, pair=1.0, distill=1.0, cls=1.0, center=0.05, radius=0.05, delta=3.0, perm=True)
    if args is not None:
        w.update(coord=args.coord_loss_weight, pair=args.pair_distance_loss_weight,
                 distill=args.pair_distance_distill_loss_weight, cls=args.pocket_cls_loss_weight,
                 center=args.pocket_distance_loss_weight, radius=args.pocket_radius_loss_weight,
                 delta=args.pocket_coord_huber_delta, perm=args.permutation_invariant)
    coords, cb, y_pred, y_by, logits, pocket_cls, p_mask, _, center, dis_map, _, radius, _ = out
    cls = w["cls"] * F.binary_cross_entropy_with_logits(logits, pocket_cls.float()) * (p_mask.numel() / p_mask.sum())
    cen = w["center"] * F.huber_loss(center, data.coords_center, delta=w["delta"])
    contact = w["pair"] * F.mse_loss(y_pred, dis_map)
    contact_by = w["pair"] * F.mse_loss(y_by, dis_map)
    distill = w["distill"] * F.mse_loss(y_by, y_pred)
    rad = w["radius"] * F.huber_loss(radius.squeeze(1), data.ligand_radius.to(radius.dtype), delta=w["delta"])
    target = data.coords
    if w["perm"]:
        new_idx = best_isomorphism_index(coords, target, data.num_atoms, data.isomorphisms)
        coord = w["coord"] * F.smooth_l1_loss(coords[new_idx], target, reduction="none").mean()
    else:
        coord = w["coord"] * F.smooth_l1_loss(coords, target)
    total = coord + contact + contact_by + distill + cls + rad + cen
    return total, dict(coord=coord, contact=contact, contact_by_pred=contact_by, distill=distill, pocket_cls=cls,
                       pocket_radius=rad, pocket_center=cen)


def get_model(args, logger):
    logger.log_message("FABind plus")
    return FABindPlus(args, args.hidden_size, args.pocket_pred_hidden_size)
