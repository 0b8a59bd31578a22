"""IaBNet_mean_and_pocket_prediction_cls_coords_dependent / get_model on the HIP engine.

Same constructor, parameter names (394 state_dict keys at the production sizes), forward / inference
signatures and return tuples as the reference (FABind/fabind/models/model.py:26-586).  The reference's
per-sample python loops (model.py:104-115, 195-288, 308-319) are replaced by vectorised index
arithmetic (pure data movement); every contraction runs in libfabind_hip.so."""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Linear

from .. import kernels as K
from .. import ops
from ..utils.utils import get_keepNode_tensor, gumbel_softmax_no_random  # noqa: F401
from .att_model import EfficientMCAttModel


class Transition_diff_out_dim(nn.Module):
    """LayerNorm -> Linear(C, n*C) -> ReLU -> Linear(n*C, out) (reference model.py:11-24)."""

    def __init__(self, embedding_channels=256, out_channels=256, n=4):
        super().__init__()
        self.layernorm = nn.LayerNorm(embedding_channels)
        self.linear1 = Linear(embedding_channels, n * embedding_channels)
        self.linear2 = Linear(n * embedding_channels, out_channels)
        torch.nn.init.xavier_uniform_(self.linear1.weight, gain=0.001)
        torch.nn.init.xavier_uniform_(self.linear2.weight, gain=0.001)

    def forward(self, z):
        shp = z.shape
        wd = ops.mm_dtype()
        t = ops.layernorm(z.reshape(-1, shp[-1]).float().contiguous(), self.layernorm.weight, self.layernorm.bias,
                          self.layernorm.eps)
        t = ops.linear(t, self.linear1.weight.to(wd), self.linear1.bias, act_epi=K.ACT_RELU)
        w2 = self.linear2.weight
        y = ops.linear(t, w2.to(wd), self.linear2.bias)
        return y.reshape(shp[:-1] + (w2.shape[0],))


def _offsets(counts):
    return torch.cumsum(counts, 0) - counts


class _AssembleRows(torch.autograd.Function):
    """cat([glb_c, glb_p, c_emb, p_emb])[idx] -- the per-complex [glb_c | ligand | glb_p | protein] layout -- with an adjoint free of float
    atomics: the ligand / residue rows are read by exactly one output row each (row gathers through cf / pf), the two global rows by one
    output row per complex (masked column sums in a fixed order).  autograd's index_select adjoint is an index_add_: the B readers of
    each global row arrived in any order and d glb_c / d glb_p were the only gradients of a training step that changed from run to run."""

    @staticmethod
    def forward(ctx, glb_c, glb_p, c_emb, p_emb, idx, cf, pf):
        ctx.save_for_backward(idx, cf, pf)
        ctx.dts = (glb_c.dtype, glb_p.dtype, c_emb.dtype, p_emb.dtype)
        return torch.cat([glb_c, glb_p, c_emb, p_emb], 0).index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        idx, cf, pf = ctx.saved_tensors
        need = ctx.needs_input_grad
        zero = torch.zeros((), dtype=g.dtype, device=g.device)
        d0 = torch.where((idx == 0)[:, None], g, zero).sum(0, keepdim=True).to(ctx.dts[0]) if need[0] else None
        d1 = torch.where((idx == 1)[:, None], g, zero).sum(0, keepdim=True).to(ctx.dts[1]) if need[1] else None
        dc = g.index_select(0, cf).to(ctx.dts[2]) if need[2] else None
        dp = g.index_select(0, pf).to(ctx.dts[3]) if need[3] else None
        return d0, d1, dc, dp, None, None, None


class IaBNet_mean_and_pocket_prediction_cls_coords_dependent(nn.Module):
    def __init__(self, args, embedding_channels=128, pocket_pred_embedding_channels=128):
        super().__init__()
        self.layernorm = nn.LayerNorm(embedding_channels)
        self.args = args
        self.coordinate_scale = args.coordinate_scale
        self.normalize_coord = lambda x: x / self.coordinate_scale
        self.unnormalize_coord = lambda x: x * self.coordinate_scale
        self.stage_prob = args.stage_prob
        n_channel = 1
        self.complex_model = EfficientMCAttModel(
            args, embedding_channels, embedding_channels, n_channel, n_edge_feats=0, n_layers=args.mean_layers,
            n_iter=args.n_iter, inter_cutoff=args.inter_cutoff, intra_cutoff=args.intra_cutoff,
            normalize_coord=self.normalize_coord, unnormalize_coord=self.unnormalize_coord)
        self.pocket_pred_model = EfficientMCAttModel(
            args, pocket_pred_embedding_channels, pocket_pred_embedding_channels, n_channel, n_edge_feats=0,
            n_layers=args.pocket_pred_layers, n_iter=args.pocket_pred_n_iter, inter_cutoff=args.inter_cutoff,
            intra_cutoff=args.intra_cutoff, normalize_coord=self.normalize_coord,
            unnormalize_coord=self.unnormalize_coord)
        self.protein_to_pocket = Transition_diff_out_dim(embedding_channels=embedding_channels, n=4, out_channels=1)
        self.glb_c = nn.Parameter(torch.ones(1, embedding_channels))
        self.glb_p = nn.Parameter(torch.ones(1, embedding_channels))
        protein_hidden = 1280 if args.use_esm2_feat else 15
        if args.esm2_concat_raw:
            protein_hidden = 1295
        self.protein_linear_whole_protein = nn.Linear(protein_hidden, embedding_channels)
        self.compound_linear_whole_protein = nn.Linear(56, embedding_channels)
        self.embedding_shrink = nn.Linear(embedding_channels, pocket_pred_embedding_channels)
        self.embedding_enlarge = nn.Linear(pocket_pred_embedding_channels, embedding_channels)
        self.distmap_mlp = nn.Sequential(nn.Linear(embedding_channels, embedding_channels), nn.ReLU(),
                                         nn.Linear(embedding_channels, 1))
        for lin in (self.protein_linear_whole_protein, self.compound_linear_whole_protein, self.embedding_shrink,
                    self.embedding_enlarge, self.distmap_mlp[0], self.distmap_mlp[2]):
            torch.nn.init.xavier_uniform_(lin.weight, gain=0.001)

    # ---- helpers ------------------------------------------------------------------------------
    def _lin(self, lin, x, **kw):
        # bf16 mode, config.set_split_sites(3): the embedding Linears around the two stacks contract in split precision on the fp32 master
        # weight (node-level, K = 1280 / 512 / 128: cheap), like the stacks' own split-precision sites; the backward is the bf16 one
        from ..config import split_sites
        W32 = lin.weight.detach() if (split_sites() >= 3 and lin.weight.dtype == torch.float32 and lin.weight.shape[1] % 32 == 0) else None
        return ops.linear(x.float().contiguous(), lin.weight.to(ops.mm_dtype()), lin.bias, W32=W32, **kw)

    @staticmethod
    def _assemble_index(segment, is_global, n_c, n_p):
        """(gather index into [glb_c | glb_p | c_emb | p_emb], ligand-atom rows, residue rows) of a complex layout: index glue that
        depends on the batch only (two read-backs: `plan_stage1` builds it ahead of the step)."""
        seg1 = segment.bool() if not segment.is_floating_point() else segment > 0.5
        N = seg1.shape[0]
        idx = torch.empty(N, dtype=torch.long, device=seg1.device)
        idx[is_global & ~seg1] = 0
        idx[is_global & seg1] = 1
        # the rows of the ligand atoms / residues as INDEX tensors (two read-backs here, before the model that consumes the result is
        # queued): the callers gather the model's output rows with index_select -- boolean-mask indexing would read the counts back
        # again behind the model's forward pass, and once more in its backward
        cf = torch.nonzero(~seg1 & ~is_global).squeeze(1)
        pf = torch.nonzero(seg1 & ~is_global).squeeze(1)
        idx[cf] = 2 + torch.arange(n_c, device=idx.device)
        idx[pf] = 2 + n_c + torch.arange(n_p, device=idx.device)
        return idx, cf, pf

    def _assemble(self, segment, is_global, c_emb, p_emb, index=None):
        """[glb_c | ligand | glb_p | protein] per complex as one gather (replaces model.py:104-115)."""
        idx, cf, pf = index if index is not None else self._assemble_index(segment, is_global, c_emb.shape[0], p_emb.shape[0])
        return _AssembleRows.apply(self.glb_c, self.glb_p, c_emb, p_emb, idx, cf, pf), cf, pf

    @staticmethod
    def _classifier_index(data):
        """Padded [B, Lmax] bookkeeping of the pocket classifier (model.py:138-144): batch-only, two read-backs."""
        pb = data['protein_whole'].batch
        B = int(pb[-1].item()) + 1
        cnt = torch.bincount(pb, minlength=B)
        loc = torch.arange(pb.shape[0], device=pb.device) - _offsets(cnt)[pb]
        Lmax = int(cnt.max().item())
        mask = torch.zeros(B, Lmax, dtype=torch.bool, device=pb.device)
        mask[pb, loc] = True
        xyz = torch.zeros(B, Lmax, 3, dtype=data.node_xyz_whole.dtype, device=pb.device)
        xyz[pb, loc] = data.node_xyz_whole
        return dict(pb=pb, B=B, cnt=cnt, loc=loc, Lmax=Lmax, mask=mask, xyz=xyz)

    @torch.no_grad()
    def plan_stage1(self, data):
        """Everything a STAGE-1 forward needs that depends on the batch alone and costs a host round trip: the gather indices of both
        complex layouts, the classifier's padded bookkeeping, the (pocket, ligand) pair lists, the output-row indices, and -- through
        `engine.prefetch` -- the batch layouts and the ctx / inter graphs of the input coordinates of BOTH stack models.  A forward
        without a plan reads ~18 sizes back at 18 points of the step, each one a drain of the launch queue (the host cannot run ahead of
        the device, and the device idles while the host catches up: 51 ms per step for 45.5 ms of kernels at the bench shape).
        Call it on the data feeder's stream while the previous step is still running, hand the result to `forward(data, stage=1,
        plan=...)`: the forward then waits for the plan's event on the device and makes no host round trip of its own (stage 1 only:
        stage 2 crops the pocket around a PREDICTED centre).  The tensors of `data` must not be replaced between the two calls."""
        from .. import engine
        p = {}
        w, cx, cb = data['complex_whole_protein'], data['complex'], data['compound'].batch
        n_c, n_pw = data['compound'].node_feats.shape[0], data['protein_whole'].node_feats.shape[0]
        p['keep_idx'] = torch.nonzero(data['pocket'].keepNode).squeeze(1)
        p['asm_w'] = self._assemble_index(w.segment, w.is_global, n_c, n_pw)
        p['cls'] = self._classifier_index(data)
        p['Xw'] = self.normalize_coord(w.node_coords.unsqueeze(-2)).float()
        p['Xl_w'] = self.normalize_coord(w.node_coords_LAS.unsqueeze(-2)).float()
        engine.prefetch(self.pocket_pred_model, p['Xw'], w.batch, w.segment,
                        data['complex_whole_protein', 'c2c', 'complex_whole_protein'].edge_index)
        p['asm_c'] = self._assemble_index(cx.segment, cx.is_global, n_c, p['keep_idx'].shape[0])
        B = p['cls']['B']
        pkb = data['pocket'].batch
        p['pairs'] = self._pair_lists(pkb, cb, torch.bincount(pkb, minlength=B), torch.bincount(cb, minlength=B))
        seg1 = cx.segment.bool() if not cx.segment.is_floating_point() else cx.segment > 0.5
        p['cidx'] = torch.nonzero(~seg1 & ~cx.is_global).squeeze(1)
        p['pidx'] = torch.nonzero(seg1 & ~cx.is_global).squeeze(1)
        p['Xn'] = self.normalize_coord(cx.node_coords.unsqueeze(-2)).float()
        p['Xl'] = self.normalize_coord(cx.node_coords_LAS.unsqueeze(-2)).float()
        engine.prefetch(self.complex_model, p['Xn'], cx.batch, cx.segment, data['complex', 'c2c', 'complex'].edge_index)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(cb.device))
        p['event'] = ev
        return p

    @staticmethod
    def _adopt_plan(plan, device):
        """The current stream waits for the plan; the allocator learns that the plan's tensors (made on another stream) are used here."""
        if plan.get('_adopted'):
            return
        cur = torch.cuda.current_stream(device)
        cur.wait_event(plan['event'])

        def rec(v):
            if torch.is_tensor(v):
                if v.is_cuda:
                    v.record_stream(cur)
            elif isinstance(v, dict):
                for u in v.values():
                    rec(u)
            elif isinstance(v, (tuple, list)):
                for u in v:
                    rec(u)
                if hasattr(v, 'blocks'):
                    rec(v.blocks.tensors())
        rec({k: v for k, v in plan.items() if k != 'event'})
        plan['_adopted'] = True

    def _pocket_head(self, data, plan=None):
        """Whole-protein pocket model + classifier (model.py:98-141)."""
        w = data['complex_whole_protein']
        c_emb = self._lin(self.compound_linear_whole_protein, data['compound'].node_feats)
        p_emb = self._lin(self.protein_linear_whole_protein, data['protein_whole'].node_feats)
        h0, cf, pf = self._assemble(w.segment, w.is_global, c_emb, p_emb, index=plan['asm_w'] if plan else None)
        h0 = self._lin(self.embedding_shrink, h0)
        Xw = plan['Xw'] if plan else self.normalize_coord(w.node_coords.unsqueeze(-2)).float()
        Xl = plan['Xl_w'] if plan else self.normalize_coord(w.node_coords_LAS.unsqueeze(-2)).float()
        # the padded [B, Lmax] bookkeeping of the classifier reads two sizes back: done BEFORE the pocket model is queued, so that
        # nothing waits behind its forward pass (or ahead of the step altogether: plan_stage1)
        ci = plan['cls'] if plan else self._classifier_index(data)
        pb, B, cnt, loc, Lmax, mask, xyz = ci['pb'], ci['B'], ci['cnt'], ci['loc'], ci['Lmax'], ci['mask'], ci['xyz']
        _, hw = self.pocket_pred_model(
            Xw, h0, batch_id=w.batch, segment_id=w.segment, mask=w.mask, is_global=w.is_global,
            compound_edge_index=data['complex_whole_protein', 'c2c', 'complex_whole_protein'].edge_index,
            LAS_edge_index=data['complex_whole_protein', 'LAS', 'complex_whole_protein'].edge_index,
            batched_complex_coord_LAS=Xl, LAS_mask=None)
        hw = self._lin(self.embedding_enlarge, hw)
        c_out, p_out = ops.take_unique_rows(hw, cf), ops.take_unique_rows(hw, pf)
        logits_flat = self.protein_to_pocket(p_out).squeeze(-1)                      # [sum L]
        logits = torch.zeros(B, Lmax, dtype=logits_flat.dtype, device=pb.device).index_put((pb, loc), logits_flat)
        return dict(B=B, c_out=c_out, p_out=p_out, logits=logits, mask=mask, xyz=xyz, pb=pb, loc=loc, cnt=cnt,
                    logits_flat=logits_flat)

    def _soft_center(self, logits, mask, xyz, noise):
        """Gumbel-softmax pocket centre (reference model.py:146-158) as ONE kernel each way (ops.pocket_center, csrc/heads.hip).  Train
        mode draws the Gumbel samples exactly as F.gumbel_softmax does (same generator calls), the kernel adds them."""
        gn = None
        if noise:
            gn = -torch.empty((logits.shape[0], logits.shape[1], 2), dtype=torch.float32, device=logits.device).exponential_().log()
        return ops.pocket_center(logits, mask, xyz, tau=self.args.gs_tau, hard=self.args.gs_hard, noise=gn)

    @torch.no_grad()
    def _stage2_indices(self, data, head, center):
        """Pocket crop + re-batch as index arithmetic (replaces the python loop model.py:195-288)."""
        dev = center.device
        pb, B = head['pb'], head['B']
        cb = data['compound'].batch
        keep = get_keepNode_tensor(data.node_xyz_whole, self.args.pocket_radius, None, center.detach()[pb])
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
        glbc_pos, glbp_pos = off, off + ncnt + 1
        segment = torch.zeros(N, dtype=torch.bool, device=dev)
        segment[prot_pos] = True
        segment[glbp_pos] = True
        is_global = torch.zeros(N, dtype=torch.bool, device=dev)
        is_global[glbc_pos] = True
        is_global[glbp_pos] = True
        mask = ~segment | is_global
        batch = torch.repeat_interleave(torch.arange(B, device=dev), n)
        return dict(keep=keep, less5=less5, n=n, off=off, N=N, pocket_batch=pocket_batch, lig_pos=lig_pos,
                    prot_pos=prot_pos, segment=segment, is_global=is_global, mask=mask, batch=batch, ncnt=ncnt, kcnt=kcnt)

    def _stage2(self, data, head, center):
        ix = self._stage2_indices(data, head, center)
        dev = center.device
        cb = data['compound'].batch
        B = head['B']
        pocket_xyz = data.node_xyz_whole[ix['keep']]
        pemb = ops.take_unique_rows(head['p_out'], torch.nonzero(ix['keep']).squeeze(1))
        H, _, _ = self._assemble(ix['segment'], ix['is_global'], head['c_out'], pemb)
        with torch.no_grad():
            li = data['compound'].node_coords.float()
            mean_l = ops.sum_sorted_segments(li, ix['ncnt']) / ix['ncnt'][:, None]              # (fixed-order sums: stage 2 is bit-reproducible)
            mean_p = ops.sum_sorted_segments(pocket_xyz.float(), ix['kcnt']) / ix['kcnt'][:, None]
            X = torch.zeros(ix['N'], 3, device=dev)
            X[ix['lig_pos']] = li - mean_l[cb] + mean_p[cb]
            X[ix['prot_pos']] = pocket_xyz.float()
            XL = torch.zeros(ix['N'], 3, device=dev)
            if self.args.compound_coords_init_mode in ('redocking', 'redocking_no_rotate'):
                XL[ix['lig_pos']] = li
            else:
                XL[ix['lig_pos']] = data['compound'].rdkit_coords.float()
            el, ll = data['compound_atom_edge_list'], data['LAS_edge_list']
            c2c = (el.x + ix['off'][el.batch][:, None]).t().contiguous().long()
            las = (ll.x + ix['off'][ll.batch][:, None]).t().contiguous().long()
            # distance map against the INITIAL ligand pose (model.py:286-287), protein-major per complex
            pairs = self._pair_lists(ix['pocket_batch'], cb, ix['kcnt'], ix['ncnt'])
            pi, ci = pairs
            dis_map = (pocket_xyz.float()[pi] - li[ci]).norm(dim=-1).clamp(max=10.0)
        # the reference overwrites the batch's `complex` stores (model.py:291-300)
        cx = data['complex']
        cx.node_coords, cx.node_coords_LAS, cx.segment, cx.mask, cx.is_global = X, XL, ix['segment'], ix['mask'], ix['is_global']
        data['complex', 'c2c', 'complex'].edge_index = c2c
        data['complex', 'LAS', 'complex'].edge_index = las
        return dict(H=H, X=X, XL=XL, segment=ix['segment'], mask=ix['mask'], is_global=ix['is_global'], batch=ix['batch'],
                    c2c=c2c, LAS=las, pocket_xyz=pocket_xyz, pocket_batch=ix['pocket_batch'], dis_map=dis_map,
                    less5=ix['less5'], pairs=pairs)

    @staticmethod
    def _pair_lists(pocket_batch, cb, kcnt, ncnt):
        """(pocket index, ligand index) of every valid pair, protein-major inside each complex."""
        dev = pocket_batch.device
        reps = ncnt[pocket_batch]                                   # each pocket residue pairs with Nc_b atoms
        pi = torch.repeat_interleave(torch.arange(pocket_batch.shape[0], device=dev), reps)
        start = _offsets(reps)
        j = torch.arange(pi.shape[0], device=dev) - start[pi]
        ci = _offsets(ncnt)[pocket_batch][pi] + j
        # the same pairs as dense per-complex blocks (ops.PairBlocks): the distance head's adjoints walk them without index glue
        kn = torch.stack([kcnt, ncnt]).tolist()
        return _Pairs(pi, ci, ops.PairBlocks(kn[0], kn[1], dev))

    def _complex_and_heads(self, data, g, plan=None):
        scale = self.coordinate_scale
        cb = data['compound'].batch
        if plan:
            Xn, Xl, cidx, pidx = plan['Xn'], plan['Xl'], plan['cidx'], plan['pidx']
        else:
            Xn = self.normalize_coord(g['X'].unsqueeze(-2)).float()
            Xl = self.normalize_coord(g['XL'].unsqueeze(-2)).float()
            # the row indices of the ligand atoms / pocket residues are read back BEFORE the complex model is queued: boolean-mask indexing
            # of its outputs would read their counts back behind its whole forward pass (three queue drains per step)
            seg1 = g['segment'].bool() if not g['segment'].is_floating_point() else g['segment'] > 0.5
            cidx = torch.nonzero(~seg1 & ~g['is_global']).squeeze(1)
            pidx = torch.nonzero(seg1 & ~g['is_global']).squeeze(1)
        Xo, Ho = self.complex_model(Xn, g['H'], batch_id=g['batch'], segment_id=g['segment'], mask=g['mask'],
                                    is_global=g['is_global'], compound_edge_index=g['c2c'], LAS_edge_index=g['LAS'],
                                    batched_complex_coord_LAS=Xl, LAS_mask=None)
        coords_n = ops.take_unique_rows(Xo, cidx).squeeze(-2)
        return Ho, cidx, pidx, coords_n

    def _dist_heads(self, data, g, Ho, cidx, pidx, coords_n):
        cb = data['compound'].batch
        pairs = g.get('pairs')
        if pairs is None:
            B = int(cb[-1].item()) + 1
            pairs = self._pair_lists(g['pocket_batch'], cb, torch.bincount(g['pocket_batch'], minlength=B), torch.bincount(cb, minlength=B))
        pi, ci = pairs
        ln = ops.layernorm(ops.take_unique_rows(Ho, torch.cat([pidx, cidx])), self.layernorm.weight, self.layernorm.bias, self.layernorm.eps)
        npk = pidx.shape[0]
        blocks = getattr(pairs, 'blocks', None)
        hd = ops.rows_hadamard(ln, pi, npk + ci, a_sorted=True, blocks=blocks, n_a=npk)               # LN(p_i) * LN(c_j)
        wd = ops.mm_dtype()
        part = ops.linear_rowdot(hd, self.distmap_mlp[0].weight.to(wd), self.distmap_mlp[0].bias,
                                 self.distmap_mlp[2].weight[0].contiguous(), act_epi=K.ACT_RELU)
        y_pred = (part.sum(1) + self.distmap_mlp[2].bias).sigmoid() * 10
        xp = self.normalize_coord(g['pocket_xyz']).float()
        if blocks is not None and coords_n.is_cuda:
            # cdist -> valid pairs -> unnormalize -> clamp as ONE kernel each way over the pair blocks (ops.pair_dist, csrc/heads.hip)
            y_by = ops.pair_dist(xp, coords_n, blocks, scale=self.coordinate_scale, lo=0.0, hi=10.0)
        else:
            y_by = self.unnormalize_coord((xp[pi] - ops.take_rows_few(coords_n, ci)).norm(dim=-1)).clamp(0, 10)
        return y_pred, y_by

    # ---- reference API ------------------------------------------------------------------------
    def forward(self, data, stage=1, train=False, plan=None):
        """plan (optional, not in the reference's signature): the result of `plan_stage1(data)` built ahead of the step; used when the
        forward takes the stage-1 path, ignored otherwise."""
        cb = data['compound'].batch
        if plan is not None:
            self._adopt_plan(plan, cb.device)
        if plan is not None:
            keep_idx = plan['keep_idx']
        else:
            keep_idx = torch.nonzero(data['pocket'].keepNode).squeeze(1) if (stage == 1 or self.pocket_pred_model.training) else None
        head = self._pocket_head(data, plan)
        training = self.pocket_pred_model.training
        center = self._soft_center(head['logits'], head['mask'], head['xyz'], noise=training)
        pocket_cls = torch.zeros_like(head['mask'], dtype=data.pocket_idx.dtype)
        pocket_cls[head['pb'], head['loc']] = data.pocket_idx
        if training:
            dist = torch.norm(data.coords_center - center, p=2, dim=-1).mean(dim=-1)
            final_stage = 2 if (dist < self.args.center_dist_threshold and random.random() < self.stage_prob) else 1
        else:
            final_stage = stage
        if final_stage == 2:
            c2 = center
            if self.args.local_eval or (self.args.train_pred_pocket_noise and train):
                c2 = center + self.args.train_pred_pocket_noise * (2 * torch.rand_like(center) - 1)
            g = self._stage2(data, head, c2)
        else:
            cx = data['complex']
            pemb = ops.take_unique_rows(head['p_out'], keep_idx if keep_idx is not None else torch.nonzero(data['pocket'].keepNode).squeeze(1))
            H, _, _ = self._assemble(cx.segment, cx.is_global, head['c_out'], pemb, index=plan['asm_c'] if plan else None)

            g = dict(H=H, X=cx.node_coords, XL=cx.node_coords_LAS, segment=cx.segment, mask=cx.mask,
                     is_global=cx.is_global, batch=cx.batch, c2c=data['complex', 'c2c', 'complex'].edge_index,
                     LAS=data['complex', 'LAS', 'complex'].edge_index, pocket_xyz=data.node_xyz,
                     pocket_batch=data['pocket'].batch, dis_map=data.dis_map, less5=0)
        stage1_plan = plan if (plan is not None and final_stage != 2) else None
        if stage1_plan is not None:
            g['pairs'] = plan['pairs']
        if g.get('pairs') is None:        # (the pair lists read a size back: built before the complex model is queued)
            B = head['B']
            g['pairs'] = self._pair_lists(g['pocket_batch'], cb, torch.bincount(g['pocket_batch'], minlength=B), torch.bincount(cb, minlength=B))
        Ho, cidx, pidx, coords_n = self._complex_and_heads(data, g, stage1_plan)
        y_pred, y_by = self._dist_heads(data, g, Ho, cidx, pidx, coords_n)
        return (self.unnormalize_coord(coords_n), cb, y_pred, y_by, head['logits'] * head['mask'], pocket_cls,
                head['mask'], head['xyz'], center, g['dis_map'], g['less5'])

    def inference(self, data):
        head = self._pocket_head(data)
        with torch.no_grad():
            lg, m = head['logits'].detach(), head['mask']
            hard = (lg.sigmoid().round() == 1) & m
            nh = hard.sum(1)
            hard_c = (hard.unsqueeze(-1) * head['xyz']).sum(1) / nh.clamp(min=1).unsqueeze(-1)
            pt = lg.sigmoid().unsqueeze(-1)
            y = gumbel_softmax_no_random(torch.log(torch.cat([1. - pt, pt], -1)), tau=self.args.gs_tau,
                                         hard=self.args.gs_hard)
            wgt = (y[..., 1] * m).unsqueeze(-1)
            soft_c = (wgt * head['xyz']).sum(1) / wgt.sum(1)
            center = torch.where((nh > 0).unsqueeze(-1), hard_c, soft_c)
        g = self._stage2(data, head, center)
        data['complex'].batch = g['batch']
        _, _, _, coords_n = self._complex_and_heads(data, g)
        return self.unnormalize_coord(coords_n), data['compound'].batch


class _Pairs(tuple):
    """(pi, ci) -- what `_pair_lists` always returned -- with the block descriptors of the same pairs riding along."""

    def __new__(cls, pi, ci, blocks):
        t = super().__new__(cls, (pi, ci))
        t.blocks = blocks
        return t


def get_model(args, logger, device=None):
    if args.mode == 5:
        logger.log_message("FABind")
        return IaBNet_mean_and_pocket_prediction_cls_coords_dependent(args, args.hidden_size, args.pocket_pred_hidden_size)
    raise NotImplementedError("only args.mode == 5 exists in the reference (model.py:582-586)")


def compute_loss(out, data, args=None):
    """The reference's train-step loss (main_fabind.py:398-417) as a function of the forward's 11-tuple.

    Returns (loss, dict of the six terms: pocket_cls, pocket_center, contact, contact_by_pred, distill, coord)."""
    w = dict(coord=1.0, pair=1.0, distill=1.0, cls=1.0, center=0.05, delta=3.0)
    if args is not None:
        w.update(coord=args.coord_loss_weight, pair=args.pair_distance_loss_weight,
                 distill=args.pair_distance_distill_loss_weight, cls=args.pocket_cls_loss_weight,
                 center=args.pocket_distance_loss_weight, delta=args.pocket_coord_huber_delta)
    coords, cb, y_pred, y_by, logits, pocket_cls, p_mask, _, center, dis_map, _ = out
    if coords.is_cuda:
        # one launch for the six terms and their sum, one for the five gradient seeds (ops.six_term_loss, csrc/heads.hip: SURVEY K20)
        return ops.six_term_loss(coords, y_pred, y_by, logits, center, data.coords, dis_map, pocket_cls, p_mask, data.coords_center, w)
    raise RuntimeError("fabind_amd.compute_loss: the model's outputs are HIP tensors; got %s" % coords.device)
