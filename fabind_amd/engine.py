"""Host orchestration of the FABind layer stack on the HIP kernels (one process per GPU).

Mirrors, call for call, what the reference's `EfficientMCAttModel.forward` -> `MCAttEGNN.forward`
-> `MC_E_GCL` / `MC_Att_L` / `CrossAttentionModule` do (FABind/fabind/models/att_model.py:170-246,
egnn.py:130-144,308-333,392-466, cross_att.py:24-54) but on the MI355X-first data layout:

* nodes stay in the reference's complex-contiguous order, so the "dense" protein/ligand views are
  contiguous slices (no to_dense_batch / un-batch copies);
* graphs are int32 row-sorted CSR rebuilt on device every refinement iteration;
* the first Linear of every edge MLP is split column-wise and evaluated per NODE, then gathered;
* the [B,P,C,H] pair tensor is never materialised in FABind-v1 mode: every layer consumes the same
  z0 = W_o(a_i*b_j)+b_o (att_model.py:198-206; egnn.py:413-418 discards the update), so the
  RowAttention pair biases are one K=H contraction per forward, and `pair_transition` is only
  evaluated at the (ligand, protein) pairs that carry an inter edge (egnn.py:286-304 gathers nothing else).

Only parameter-only preprocessing (weight slicing / composition, O(H^2)) uses torch ops; everything that
touches activations runs in libfabind_hip.so.
"""
import math

import os

import numpy as np
import torch

from . import kernels as K
from . import ops

from .config import fp32_storage, get_precision, set_precision, split_sites  # noqa: F401
from .config import x3_backward_bf16 as config_x3_backward_bf16
from .param_pack import EagerPack, ParamPack


FUSED_EDGE = True     # forward-only bf16: fused gather->GEMM->GEMM->segment-sum edge kernel
DEBUG_CAPTURE = None  # set to a dict to record per-layer outputs of the LAST egnn_forward call (tests)


def _wd():
    return torch.float32 if fp32_storage() else torch.bfloat16


# ------------------------------------------------------------------------------------------------
# layout of a batch of complexes
# ------------------------------------------------------------------------------------------------
def _count(idx, n):
    """Histogram of idx over [0, n) WITHOUT a host round trip: torch.bincount reads the largest index back to size its output, i.e. it
    waits for everything queued -- `ctx_by_col` is first asked for in the middle of the backward pass, `Layout.ranges` in the middle of a
    full-model step (tools/probes/sync_sites.py lists every synchronising call of a step)."""
    return torch.zeros(n, dtype=torch.int64, device=idx.device).index_add_(0, idx.long(), torch.ones_like(idx, dtype=torch.int64))


class Layout:
    """Per-batch index arrays.  One host sync (reading per-complex node counts) per construction; `Layout.of` reuses the layout of
    the SAME index tensors (object identity + version counter), so a batch that is run again -- the bench loop, several stages or
    epochs over a resident batch -- pays the sync and the numpy assembly once.  A new batch is a new tensor object and is rebuilt."""
    _cache = []                # [(weakref(batch_id), version, weakref(segment_id), version, n_pair_out, layout)], most recent first

    @classmethod
    def of(cls, batch_id, segment_id, n_pair_out=8):
        import weakref
        if os.environ.get("FABIND_LAYOUT_CACHE", "1") == "0":       # development knob for same-box A/B runs
            return cls(batch_id, segment_id, n_pair_out)
        for ent in cls._cache:
            if ent[0]() is batch_id and ent[2]() is segment_id and ent[1] == batch_id._version and ent[3] == segment_id._version \
                    and ent[4] == n_pair_out:
                return ent[5]
        lay = cls(batch_id, segment_id, n_pair_out)
        cls._cache.insert(0, (weakref.ref(batch_id), batch_id._version, weakref.ref(segment_id), segment_id._version, n_pair_out, lay))
        del cls._cache[4:]
        return lay

    def __init__(self, batch_id, segment_id, n_pair_out=8):
        dev = batch_id.device
        B = int(batch_id[-1].item()) + 1
        seg = (segment_id > 0.5) if segment_id.is_floating_point() else segment_id.bool()
        bid = batch_id.long()                               # (two read-backs in all: B above, the counts here -- no torch.bincount)
        # nodes are complex-contiguous (the layout below relies on it, like the reference's to_dense_batch): the per-complex counts are
        # differences of the complexes' boundaries in the sorted batch vector (an index_add_ of 98,688 ones into 64 bins is 0.4 ms of
        # contended atomics); whether the vector IS sorted rides in the same read-back
        bound = torch.searchsorted(bid, torch.arange(B + 1, dtype=torch.int64, device=dev))
        cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(seg.to(torch.int64), 0)])
        ok = (bid[1:] >= bid[:-1]).all().to(torch.int64).reshape(1) if bid.numel() > 1 else torch.ones(1, dtype=torch.int64, device=dev)
        cnt = torch.cat([bound[1:] - bound[:-1], cs[bound[1:]] - cs[bound[:-1]], ok]).cpu().numpy()
        if cnt[-1] != 1:
            raise ValueError("fabind_amd: batch_id must be sorted (the nodes of a complex contiguous), as the reference's to_dense_batch requires")
        n, P = cnt[:B].astype(np.int64), cnt[B:2 * B].astype(np.int64)
        C = n - P
        off = np.concatenate([[0], np.cumsum(n)])
        self.B, self.N = B, int(off[-1])
        self.n, self.P, self.C, self.off = n, P, C, off
        self.max_n, self.max_P, self.max_C = int(n.max()), int(P.max()), int(C.max())
        coff = np.concatenate([[0], np.cumsum(C)])
        poff = np.concatenate([[0], np.cumsum(P)])
        pair_off = np.concatenate([[0], np.cumsum(P * C)])
        self.n_pairs = int(pair_off[-1])
        self.sumC, self.sumP = int(coff[-1]), int(poff[-1])
        # every index array is assembled with numpy vector ops into ONE int32 buffer and uploaded once
        bidx = np.repeat(np.arange(B), n)
        loc = np.arange(self.N) - off[:-1][bidx]
        is_c = loc < C[bidx]
        c_index, p_index = np.nonzero(is_c)[0], np.nonzero(~is_c)[0]
        inv = np.empty(self.N, dtype=np.int64)
        inv[c_index] = np.arange(self.sumC)
        inv[p_index] = self.sumC + np.arange(self.sumP)
        lo = lambda v: (v & 0xFFFFFFFF).astype(np.uint32).astype(np.int64)     # low word, reinterpreted as int32 on upload
        hi = lambda v: v >> 32
        one, zero = np.ones(B, dtype=np.int64), np.zeros(B, dtype=np.int64)
        o, cf, pf, po = off[:-1], coff[:-1], poff[:-1], pair_off[:-1]
        NO = n_pair_out
        self.NO = NO
        desc = lambda cols: np.stack(cols, 1).reshape(-1)
        parts = dict(
            node_off=off, c_cnt=C, c_index=c_index, p_index=p_index, inv_perm=inv,
            desc_p=desc([pf, P, cf, C, lo(po), hi(po), C, one]),
            desc_c=desc([cf, C, pf, P, lo(po), hi(po), one, C]),
            # the same blocks with the protein side addressed IN PLACE in the node layout (rows off[b]+C[b] ...): protein-side
            # arrays of the cross attention stay [N, .] and never get gathered / scattered (ligand rows are don't-care)
            desc_pf=desc([o + C, P, cf, C, lo(po), hi(po), C, one]),
            desc_cf=desc([cf, C, o + C, P, lo(po), hi(po), one, C]),
            pb_groups=desc([o + C, P, cf * NO, C * NO, lo(po * NO), hi(po * NO), C * NO, zero]),
            # adjoint of the pair bias as two ragged batched contractions per attention block (ops._PairBias.backward)
            pb_tn_groups=desc([lo(po * NO), hi(po * NO), C * NO, C * NO, o + C, P, zero, zero]))   # c_off filled per H in ops
        flat = np.concatenate([np.asarray(v, dtype=np.int64) for v in parts.values()])
        flat = torch.from_numpy((flat & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).to(dev)
        pos = 0
        for name, v in parts.items():
            t = flat[pos:pos + len(v)]
            pos += len(v)
            setattr(self, name, t.view(-1, 8) if name.startswith(("desc", "pb_")) else t)
        for name in ("desc_p", "desc_c", "desc_pf", "desc_cf"):
            getattr(self, name)._fab_covers_all = True      # these block descriptors tile the whole pair list (ops._CrossAttn)
        self.c_index64, self.p_index64 = self.c_index.long(), self.p_index.long()
        self.inv_perm = self.inv_perm.long()
        self.pb_max_n = int((C * NO).max())
        self.coff, self.pair_off_np = coff, pair_off

    def ranges(self, idx_first):
        """Per-complex [start,end) offsets of a complex-contiguous edge list given its first-node ids."""
        if idx_first.is_cuda and idx_first.dtype == torch.int32 and idx_first.is_contiguous():
            out = torch.empty(self.B + 1, dtype=torch.int32, device=idx_first.device)       # one kernel: a binary search per complex
            K.check(K._lib.load().fabind_lower_bound(K.ptr(idx_first), idx_first.shape[0], K.ptr(self.node_off), self.B + 1, K.ptr(out),
                                                     K.stream()), "fabind_lower_bound")
            return out
        which = torch.bucketize(idx_first, self.node_off[1:].to(idx_first.dtype), right=True)
        cnt = _count(which, self.B)
        out = torch.zeros(self.B + 1, dtype=torch.int32, device=idx_first.device)
        out[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        return out


class Graph:
    """ctx + inter CSR of the current coordinates (rebuilt each refinement iteration, no grad)."""
    _warned_fallback = False

    @torch.no_grad()
    def __init__(self, lay, x, bond_row, bond_col, bond_off, cut_intra, cut_inter):
        deg_ctx, deg_int = K.edges_count(x, lay.node_off, lay.c_cnt, lay.B, lay.max_n, bond_row, bond_off, cut_intra,
                                         cut_inter)
        rp_ctx, rp_int = K.exclusive_scan(deg_ctx), K.exclusive_scan(deg_int)
        lo = lay.node_off[:-1].long()
        n_lig_rows = (rp_int[lo + lay.c_cnt.long()] - rp_int[lo]).sum()
        # (the two row counts of the inter-edge attention's deal-by-degree ride in the same read-back)
        E_ctx, E_int, n_half, n_act, n_heavy = torch.stack([rp_ctx[-1], rp_int[-1], n_lig_rows.to(rp_int.dtype), (deg_int > 0).sum().to(rp_int.dtype),
                                                            (deg_int > K.INTER_ATTN_HEAVY).sum().to(rp_int.dtype)]).tolist()   # the one host sync
        if 2 * n_half != E_int:
            # the pair bookkeeping (inter_meta) needs every ligand->protein edge mirrored; only non-finite coordinates
            # can break that -- fail loudly on the host instead of faulting on the device
            raise RuntimeError("fabind_amd: inter-graph is not symmetric (%d ligand-row edges of %d): finite coordinates: %s"
                               % (n_half, E_int, bool(torch.isfinite(x).all())))
        if E_int == 0:
            if get_precision() == "bf16" and not Graph._warned_fallback:
                Graph._warned_fallback = True
                import warnings
                warnings.warn("fabind_amd: no ligand-protein pair within the inter cut-off in this batch -- the reference's one "
                              "fake pair is used (att_model.py:85-86).  In bf16 mode ligand coordinates are accurate to only "
                              "~0.1-0.3 A on such inputs (long edges amplify bf16 operand rounding; DESIGN.md); use "
                              "engine.set_precision('fp32') if that matters.")
            # reference fallback (att_model.py:85-86): one fake symmetric pair from the first candidate,
            # i.e. first ligand atom of complex 0 <-> its first protein residue.
            u, v = 1, int(lay.C[0]) + 1
            deg_int[u] = 1
            deg_int[v] = 1
            rp_int = K.exclusive_scan(deg_int)
            E_int, n_act, n_heavy = 2, 2, 0
            self.col_ctx, self.row_ctx, _, _ = K.edges_fill(x, lay.node_off, lay.c_cnt, lay.B, lay.max_n, bond_row,
                                                            bond_col, bond_off, cut_intra, -1.0, rp_ctx, rp_ctx * 0,
                                                            E_ctx, 0)
            self.col_int = torch.tensor([v, u], dtype=torch.int32, device=x.device)
            self.row_int = torch.tensor([u, v], dtype=torch.int32, device=x.device)
        else:
            self.col_ctx, self.row_ctx, self.col_int, self.row_int = K.edges_fill(
                x, lay.node_off, lay.c_cnt, lay.B, lay.max_n, bond_row, bond_col, bond_off, cut_intra, cut_inter, rp_ctx,
                rp_int, E_ctx, E_int)
        self.rp_ctx, self.rp_int, self.E_ctx, self.E_int = rp_ctx, rp_int, E_ctx, E_int
        if os.environ.get("FABIND_DEBUG_GRAPH", "0") == "1":             # development aid: validate the CSR on the host
            r, c, Nn = self.row_int.long(), self.col_int.long(), lay.N
            k1, k2 = torch.sort(r * Nn + c)[0], torch.sort(c * Nn + r)[0]
            nbad = int((k1 != k2).sum())
            rr = torch.repeat_interleave(torch.arange(Nn, device=r.device), (rp_int[1:] - rp_int[:-1]).long())
            print("[fabind] graph check: E_int %d, col range [%d, %d] of %d, row mismatch %d, mirror mismatches %d, finite %s"
                  % (E_int, int(c.min()), int(c.max()), Nn, int((rr != r).sum()), nbad, bool(torch.isfinite(x).all())), flush=True)
        self.red_off, self.red_idx, self.red_c, self.red_p, self.mirror = K.inter_meta(
            lay.node_off, lay.c_cnt, lay.B, rp_int, self.col_int, self.row_int)
        # the rows of the inter graph sorted by degree, descending (index glue: one stable sort per graph): the inter-edge attention kernels
        # put a row with more than 8 edges on four waves and start the heavy rows first (csrc/inter_attn_rows.hip)
        self.int_deal = (torch.argsort(deg_int, descending=True, stable=True).to(torch.int32), int(n_heavy), int(n_act))
        self._ctx_bycol = None
        self.N = lay.N

    def int_by_col(self):
        """Inter edges grouped by sending node: the graph is symmetric, so node n's incoming edges are the mirrors of
        its own CSR row."""
        return self.rp_int, self.mirror

    def ctx_by_col(self):
        """(colptr, perm): ctx edges grouped by their SENDING node (for column-wise reductions in backward)."""
        if self._ctx_bycol is None:
            colsorted, perm = torch.sort(self.col_ctx.long(), stable=True)
            cnt = _count(colsorted, self.N)
            colptr = torch.zeros(self.N + 1, dtype=torch.int32, device=perm.device)
            colptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
            self._ctx_bycol = (colptr, perm.to(torch.int32).contiguous())
        return self._ctx_bycol


# ------------------------------------------------------------------------------------------------
# parameter preparation (weight slicing / composition; differentiable torch ops on parameters only)
# ------------------------------------------------------------------------------------------------
def _cat(ts, d=0):
    return torch.cat(list(ts), d)


def gcl_params(m, pk=None):
    """Kernel-side parameter pack of one MC_E_GCL module (first edge Linear split per node, see module docstring).  pk: a
    param_pack.ParamPack collecting the whole model's copy requests (entries are handles until pk.resolve), default: built here.
    (Biases go through the pack too: their gradients then reach the pack's adjoint -- the ONE place where the queued weight-gradient
    contractions are flushed, kernels.tn_flush -- instead of each bias's own AccumulateGrad, which would flush per Linear.)"""
    wd = _wd()
    own = pk is None
    pk = pk or EagerPack(m.edge_mlp[0].weight.device)
    W1 = m.edge_mlp[0].weight
    Hin = (W1.shape[1] - 1) // 2
    H = W1.shape[0]
    d = dict(
        W_ab=pk.cat([W1[:, :Hin], W1[:, Hin:2 * Hin]], 0, wd, with_T=True),
        b_ab=pk.cat([m.edge_mlp[0].bias, pk.zeros(H)]),
        w_r=pk.copy(W1[:, 2 * Hin]), W2=pk.copy(m.edge_mlp[2].weight, wd), b2=pk.copy(m.edge_mlp[2].bias),
        Wc=pk.copy(m.coord_mlp[0].weight, wd), bc=pk.copy(m.coord_mlp[0].bias), w3=pk.copy(m.coord_mlp[2].weight[0]),
        Wn1=pk.copy(m.node_mlp[0].weight, wd, with_T=True), bn1=pk.copy(m.node_mlp[0].bias),
        Wn2=pk.copy(m.node_mlp[2].weight, wd, with_T=True), bn2=pk.copy(m.node_mlp[2].bias))
    return pk.resolve(d) if own else d


_PACK_CACHE = {}
_LAST_PARAMS = [None, None]                                # [weakref(model), list(model.parameters())] of the last cached_pack call


def last_params_of(model):
    """list(model.parameters()), reusing the walk cached_pack has just made for this very model object."""
    ref = _LAST_PARAMS[0]
    return _LAST_PARAMS[1] if (ref is not None and ref() is model) else list(model.parameters())


def cached_pack(model, builder):
    """Kernel-side parameter pack of `model`.  Without autograd (inference / sampling: the same weights serve many
    calls) the pack is built once and reused until a parameter is modified or replaced or the precision mode changes;
    under autograd it is rebuilt every call, because it must be a differentiable function of the parameters."""
    params = list(model.parameters())
    import weakref
    _LAST_PARAMS[:] = [weakref.ref(model), params]         # (the builders key their kept request plans on the same list: one module walk)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return builder(model)
    key = (get_precision(), split_sites(), tuple((p.data_ptr(), p._version) for p in params))
    ent = _PACK_CACHE.get(id(model))
    if ent is not None and ent[0] == key and ent[2]() is model:
        return ent[1]
    import weakref
    with torch.no_grad():
        P = builder(model)
    _PACK_CACHE[id(model)] = (key, P, weakref.ref(model, lambda _r, k=id(model): _PACK_CACHE.pop(k, None)))
    return P


def prepare_stack_params(model):
    """Pack the parameters of an EfficientMCAttModel for the kernels (see module docstring)."""
    return cached_pack(model, _build_stack_params)


def cam_node_params(cam, pk=None):
    """Node-path parameters of one CrossAttentionModule (both RowAttentionBlocks' projections, both node transitions, the
    32-wide Hadamard projections) in the kernels' layout.  pk: see gcl_params."""
    wd = _wd()
    own = pk is None
    pk = pk or EagerPack(cam.inter_layer.linear_p.weight.device)
    d = {}
    for tag, blk in (("p", cam.p_attention_block), ("c", cam.c_attention_block)):
        a = blk.mha
        d["Wqg_" + tag] = pk.cat([a.linear_q.weight, a.linear_g.weight], 0, wd, with_T=True)
        d["bqg_" + tag] = pk.cat([pk.zeros(a.linear_g.bias.shape[0]), a.linear_g.bias])
        d["Wkv_" + tag] = pk.cat([a.linear_k.weight, a.linear_v.weight], 0, wd, with_T=True)
        d["Wo_" + tag], d["bo_" + tag] = pk.copy(a.linear_o.weight, wd, with_T=True), pk.copy(a.linear_o.bias)
    for tag, tr in (("p", cam.p_transition), ("c", cam.c_transition)):
        d["Wt1_" + tag], d["bt1_" + tag] = pk.copy(tr.linear_1.weight, wd, with_T=True), pk.copy(tr.linear_1.bias)
        d["Wt2_" + tag], d["bt2_" + tag] = pk.copy(tr.linear_2.weight, wd, with_T=True), pk.copy(tr.linear_2.bias)
    i32 = cam.inter_layer
    zw, zb = pk.zeros(*i32.linear_p.weight.shape), pk.zeros(i32.linear_p.bias.shape[0])
    d["W_ab32"] = pk.cat([i32.linear_p.weight, zw, i32.linear_c.weight, zw], 0, wd, with_T=True)
    d["b_ab32"] = pk.cat([i32.linear_p.bias, zb, i32.linear_c.bias, zb])
    return pk.resolve(d) if own else d


def att_edge_params(m, pk=None):
    """Inter-edge attention parameters of one MC_Att_L (egnn.py:197-252): q | k | v projections with the interleaved kv split
    (k = kv[0::2], v = kv[1::2], egnn.py:205) undone, the radial columns, the coordinate MLP.  pk: see gcl_params; with a shared pk the
    composed entry `wcr` is added by `att_edge_composed` after pk.resolve."""
    wd = _wd()
    own = pk is None
    pk = pk or EagerPack(m.linear_q.weight.device)
    d = {}
    Wkv, bkv = m.linear_kv.weight, m.linear_kv.bias
    d["Wqkv"] = pk.cat([m.linear_q.weight, Wkv[0::2, 1:], Wkv[1::2, 1:]], 0, wd, with_T=True)
    d["bqkv"] = pk.cat([m.linear_q.bias, bkv[0::2], bkv[1::2]])
    d["w_rk"], d["w_rv"] = pk.copy(Wkv[0::2, 0]), pk.copy(Wkv[1::2, 0])
    d["Wc"], d["bc"] = pk.copy(m.coord_mlp[0].weight, wd, with_T=True), pk.copy(m.coord_mlp[0].bias)
    d["w3"] = pk.copy(m.coord_mlp[2].weight[0])
    if own:
        d = pk.resolve(d)
        att_edge_composed(m, d)
    return d


def att_edge_composed(m, d):
    """The product entry of att_edge_params (an ordinary torch op on the resolved pack)."""
    d["wcr"] = (m.coord_mlp[0].weight @ d["w_rv"]).contiguous()


PACK_PLAN = os.environ.get("FABIND_PACK_PLAN", "1") == "1"      # keep a model's pack requests across calls (0: rebuild them per call, A/B)
_PLAN_CACHE = {}           # id(model) -> (key, ParamPack with its requests, request tree, weakref(model))


def _stack_requests(model):
    """The copy-type requests of a stack model's parameter pack (the ParamPack holding them, and the tree of handles / raw parameters)."""
    wd = _wd()
    gnn = model.gnn
    H = gnn.hidden_nf
    L = gnn.n_layers
    il = model.inter_layer
    dev = il.linear_p.weight.device
    pk = EagerPack(dev) if os.environ.get("FABIND_PARAM_PACK", "1") == "0" else ParamPack(dev)
    P = {"H": H, "L": L}
    P["W_ab0"] = pk.cat([il.linear_p.weight, il.linear_c.weight], 0, wd, with_T=True)
    P["b_ab0"] = pk.cat([il.linear_p.bias, il.linear_c.bias])
    rows, rb = [], []
    for i in range(L):
        cam = getattr(gnn, "att_%d" % i).cross_attn_module
        for blk in (cam.p_attention_block, cam.c_attention_block):
            rows += [blk.linear.weight, blk.linear_g.weight]                 # [4, H] each: lin heads 0-3, gate heads 4-7
            rb += [blk.linear.bias, blk.linear_g.bias]
    P["_rows"], P["_rb"] = pk.cat(rows, 0), pk.cat(rb)                       # [2L * 8, H], [2L * 8]
    # the parameters the pair path only ever uses COMPOSED (pair_transition, the 32-wide Hadamard block's linear_out, attn_bias_proj:
    # egnn.py:208,286-304, cross_att.py:51-53), stacked over the layers by the same launch: the compositions below are then a handful of
    # batched products for all layers instead of ~10 small launches per layer each way (~100 of a pocket-sized step's ~800)
    cams = [getattr(gnn, "att_%d" % i).cross_attn_module for i in range(L)]
    P["_W1s"] = pk.cat([c.pair_transition.linear_1.weight for c in cams], 0)             # [L * 2H, H]
    P["_b1s"] = pk.cat([c.pair_transition.linear_1.bias for c in cams])                  # [L * 2H]
    P["_W2s"] = pk.cat([c.pair_transition.linear_2.weight for c in cams], 0)             # [L * H, 2H]
    P["_aps"] = pk.cat([getattr(gnn, "att_%d" % i).attn_bias_proj.weight for i in range(L)], 0)   # [L, H]
    P["_Wis"] = pk.cat([c.inter_layer.linear_out.weight for c in cams], 0)               # [L * H, 32]
    P["_bis"] = pk.cat([c.inter_layer.linear_out.bias for c in cams])                    # [L * H]
    P["W_in"], P["b_in"] = pk.copy(gnn.linear_in.weight, wd, with_T=True), pk.copy(gnn.linear_in.bias)
    P["W_out"], P["b_out"] = pk.copy(gnn.linear_out.weight, wd, with_T=True), pk.copy(gnn.linear_out.bias)
    P["gcl"] = [gcl_params(getattr(gnn, "gcl_%d" % i), pk) for i in range(L)]
    P["out_layer"] = gcl_params(gnn.out_layer, pk)
    att = []
    for i in range(L):
        m = getattr(gnn, "att_%d" % i)
        d = cam_node_params(m.cross_attn_module, pk)
        d.update(att_edge_params(m, pk))
        att.append(d)
    P["att"] = att
    return pk, P


def _build_stack_params(model):
    """Every copy-type entry (slices / concatenations / zero padding / casts of parameters) goes through ONE param_pack.ParamPack --
    one launch, one autograd node -- unless FABIND_PARAM_PACK=0 (torch ops, the round-1 behaviour); the products of parameters
    (composed weights of the pair path) are torch ops on its outputs.
    The REQUESTS (which views of which parameters go where: ~130 requests over ~1,000 parameter views, 5-8 ms of Python per call) are
    a property of the model's parameter objects, not of their values: they are built once per model and precision mode and kept;
    a training step re-runs the kept pack (fresh outputs, one launch each way).  Replaced or moved parameters rebuild them."""
    wd = _wd()
    gnn = model.gnn
    H = gnn.hidden_nf
    L = gnn.n_layers
    il = model.inter_layer
    params = last_params_of(model)
    key = (get_precision(), os.environ.get("FABIND_PARAM_PACK", "1"), tuple((id(p_), p_.data_ptr(), p_.requires_grad) for p_ in params))
    ent = _PLAN_CACHE.get(id(model))
    if ent is not None and ent[0] == key and ent[3]() is model and isinstance(ent[1], ParamPack) and PACK_PLAN:
        pk, tree = ent[1], ent[2]
    else:
        import weakref
        pk, tree = _stack_requests(model)
        if isinstance(pk, ParamPack):
            _PLAN_CACHE[id(model)] = (key, pk, tree, weakref.ref(model, lambda _r, k=id(model): _PLAN_CACHE.pop(k, None)))
    P = pk.resolve(tree)

    # ---- products of parameters
    Wo0, bo0 = il.linear_out.weight, il.linear_out.bias
    rows, rb = P.pop("_rows").view(2 * L, -1, H), P.pop("_rb").view(2 * L, -1)
    P["pb_wcomp"] = (rows @ Wo0).contiguous()                                # z0 = Wo0 (a*b) + bo0 folded in
    P["pb_bconst"] = (rows @ bo0 + rb).contiguous()
    # the 32-wide Hadamard block is zero-padded to 64 columns so that K = H + 64 stays a multiple of 64 (LDS-DMA GEMM path); the padded
    # rows/columns are exact zeros and do not change the result.  Per layer l (all layers at once, batched):
    #   Wcomp1_l = W1_l [Wo0 | Wi_l | 0]   [2H, H+64];   bcomp1_l = W1_l (bo0 + bi_l) + b1_l   [2H];   u_l = W2_l^T a_l   [2H]
    W1s, b1s = P.pop("_W1s").view(L, 2 * H, H), P.pop("_b1s").view(L, 2 * H)
    W2s, aps = P.pop("_W2s").view(L, H, 2 * H), P.pop("_aps").view(L, H)
    Wis, bis = P.pop("_Wis").view(L, H, -1), P.pop("_bis").view(L, H)
    n32 = Wis.shape[2]
    A = (W1s.reshape(L * 2 * H, H) @ Wo0).view(L, 2 * H, H)
    Bm = torch.bmm(W1s, Wis)
    Wcomp1 = torch.cat([A, Bm, torch.zeros((L, 2 * H, 64 - n32), dtype=A.dtype, device=A.device)], 2).to(wd)
    bcomp1 = torch.baddbmm(b1s.unsqueeze(2), W1s, (bo0.unsqueeze(0) + bis).unsqueeze(2)).squeeze(2)
    u = torch.bmm(W2s.transpose(1, 2), aps.unsqueeze(2)).squeeze(2)
    Wc_l, bc_l, u_l = Wcomp1.unbind(0), bcomp1.unbind(0), u.unbind(0)        # (unbind: the adjoint is ONE stack of the layers' gradients)
    for i in range(L):
        m = getattr(gnn, "att_%d" % i)
        d = P["att"][i]
        att_edge_composed(m, d)
        d["Wcomp1"], d["bcomp1"], d["u"] = Wc_l[i], bc_l[i], u_l[i]
    _edge_frags(P)
    _split_site_masters(model, P)
    return P


def _split_site_masters(model, P):
    """fp32 master weights of the split-precision sites of the bf16 mode (config.set_split_sites), next to their bf16 pack entries: the
    input Linear and, per attention layer, the V rows of the q | k | v projection (interleaved kv split undone like `Wqkv`) and the coordinate MLP's first
    Linear.  Detached: they are operands of the FORWARD contraction only, the gradients flow through the bf16 entries."""
    if split_sites() < 1:
        return
    gnn = model.gnn
    f32 = lambda w: w.detach().float().contiguous()
    with torch.no_grad():
        P["W_in32"] = f32(gnn.linear_in.weight)
        for i in range(P["L"]):
            m = getattr(gnn, "att_%d" % i)
            Wkv = m.linear_kv.weight.detach()
            d = P["att"][i]
            # (of q | k | v only the v columns carry the gap: tools/probes/precision_qkv_parts.py -- the row-range form of ops.linear's W32)
            H_ = P["H"]
            d["Wqkv32"] = (Wkv[1::2, 1:].float().contiguous(), 2 * H_, 3 * H_)
            d["Wc32"] = f32(m.coord_mlp[0].weight)
            if split_sites() >= 2:       # + the protein-query block's output projection and both blocks' k | v projections
                cam = m.cross_attn_module
                d["Wo_p32"] = f32(cam.p_attention_block.mha.linear_o.weight)
                for tag, blk in (("p", cam.p_attention_block), ("c", cam.c_attention_block)):
                    d["Wkv_%s32" % tag] = torch.cat([blk.mha.linear_k.weight.detach(), blk.mha.linear_v.weight.detach()], 0).float().contiguous()
            if split_sites() >= 3:       # + both Linears of the protein-side Transition (the hidden layer stays fp32 between them)
                cam = m.cross_attn_module
                d["Wt_p32"] = (f32(cam.p_transition.linear_1.weight), f32(cam.p_transition.linear_2.weight))
                # ... and the ligand side's (2,600 rows at B = 64: free): its Transition and its attention block's output projection
                d["Wt_c32"] = (f32(cam.c_transition.linear_1.weight), f32(cam.c_transition.linear_2.weight))
                d["Wo_c32"] = f32(cam.c_attention_block.mha.linear_o.weight)
        if split_sites() >= 3:           # + both Linears of every node MLP
            for i in range(P["L"] + 1):
                m = getattr(gnn, "gcl_%d" % i) if i < P["L"] else gnn.out_layer
                d = P["gcl"][i] if i < P["L"] else P["out_layer"]
                d["Wn_32"] = (f32(m.node_mlp[0].weight), f32(m.node_mlp[2].weight))
            # the hi | lo fragment packs of every split-precision chain (node MLPs, both Transitions) of the model: ONE launch
            if NODE_CHAIN and NODE_CHAIN_X3 and P["H"] in (128, 256, 512):
                chains, where = [], []
                for i in range(P["L"] + 1):
                    d = P["gcl"][i] if i < P["L"] else P["out_layer"]
                    chains.append(d["Wn_32"] + (0,)); where.append((d, "_ncx_node"))
                for i in range(P["L"]):
                    d = P["att"][i]
                    chains.append(d["Wt_p32"] + (1,)); where.append((d, "_ncx_tp"))
                    chains.append(d["Wt_c32"] + (1,)); where.append((d, "_ncx_tc"))
                for (d, key), packs in zip(where, K.node_chain_x3_pack_many(chains)):
                    d[key] = packs


def _edge_frags(P):
    """Fragment packs of every GCL layer's W2 / Wc (and, when a backward pass will come, their transposes) for the fused edge kernels,
    ONE launch for the whole model (kernels.pack_frag_multi; 30 torch permute-copies per training step before).  Operand layouts of
    the pack's bf16 outputs, not differentiable: the gradients of W2 / Wc come out of the fused backward kernel's own contractions."""
    if get_precision() != "bf16" or P["H"] not in (64, 128, 256, 512):
        return
    layers = list(P["gcl"]) + [P["out_layer"]]
    train = torch.is_grad_enabled() and any(d_["W2"].requires_grad for d_ in layers)
    with torch.no_grad():
        ws = []
        for d_ in layers:
            W2, Wc = d_["W2"].detach(), d_["Wc"].detach()
            ws += [W2, Wc] + ([W2.t(), Wc.t()] if train else [])
        packs = K.pack_frag_multi(ws)
    n = 4 if train else 2
    for i, d_ in enumerate(layers):
        d_["_frags"] = tuple(packs[i * n:(i + 1) * n])


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def _fast(*ts):
    """Forward-only bf16 mode: producers apply the activations, node tensors are fed to the GEMMs as bf16."""
    return get_precision() == "bf16" and not ops.needs_grad(*ts)


def _b16(t):
    """bf16 operand copy of an activation (remembered on the tensor / emitted by its producer's epilogue: ops._mm_in)."""
    return ops._mm_in(t) if t.dtype == torch.float32 else t.to(torch.bfloat16)


def _drop(t, pr):
    """nn.Dropout of the reference modules (train mode only).  Bernoulli mask + scale: element-wise RNG plumbing
    (torch's generator, so seeds behave as in the reference); identity in eval mode and for p = 0."""
    return torch.nn.functional.dropout(t, pr, True) if pr > 0.0 else t


NODE_CHAIN = os.environ.get("FABIND_NODE_CHAIN", "1") == "1"     # forward-only bf16 passes: node MLP / Transition as one kernel (csrc/node_chain.hip)


def _node_chain(p, key, W1, b1, W2, b2, act, kind, x1, x2, residual, want16):
    """Linear -> act -> Linear + residual as one forward kernel with the hidden activation in LDS; None when the shapes do not fit
    (H in {128, 256, 512}; kind 0: W1 [H, 2H]; kind 1: W1 [2H, H]).  The fragment packs are kept in the (cached, no-grad) parameter
    pack p under `key`."""
    H = W2.shape[0]
    if not (NODE_CHAIN and H in (128, 256, 512) and W1.dtype == torch.bfloat16 and b1 is not None and b2 is not None
            and x1.dtype == torch.bfloat16 and x1.shape[1] == H and x1.stride(1) == 1 and x1.stride(0) % 8 == 0
            and residual.dtype == torch.float32 and residual.stride(1) == 1 and residual.stride(0) % 4 == 0
            and W1.shape == ((H, 2 * H) if kind == 0 else (2 * H, H)) and W2.shape == ((H, H) if kind == 0 else (H, 2 * H))):
        return None
    if kind == 0 and not (x2 is not None and x2.dtype == torch.bfloat16 and x2.shape == x1.shape and x2.stride(1) == 1 and x2.stride(0) % 8 == 0):
        return None
    packs = p.get(key)
    if packs is None:
        packs = p[key] = K.node_chain_pack(W1, W2, kind)
    out, out16 = K.node_chain_fwd(x1, x2 if kind == 0 else None, packs, b1.float(), b2.float(), act, kind, residual=residual, want16=want16)
    return ops._attach_b16(out, out16)


def _node_chain_x3(p, key, W32, b1, b2, act, kind, x1, x2, residual, want16):
    """The same chain in split precision (config.set_split_sites(3), no-grad passes): fp32 rows, fp32 master weights W32 = (W1_32, W2_32)
    as hi | lo fragment packs (kept in the cached parameter pack under `key`); None when the shapes do not fit."""
    H = W32[1].shape[0]
    ok = lambda t: t is not None and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == H and t.stride(1) == 1 and t.stride(0) % 4 == 0 \
        and t.data_ptr() % 16 == 0
    if not (NODE_CHAIN and NODE_CHAIN_X3 and H in (128, 256, 512) and b1 is not None and b2 is not None and ok(x1) and ok(residual)
            and (kind == 1 or ok(x2)) and tuple(W32[0].shape) == ((H, 2 * H) if kind == 0 else (2 * H, H))
            and tuple(W32[1].shape) == ((H, H) if kind == 0 else (H, 2 * H))):
        return None
    packs = p.get(key)
    if packs is None:
        with torch.no_grad():
            packs = p[key] = K.node_chain_x3_pack_many([(W32[0].detach(), W32[1].detach(), kind)])[0]      # (one launch)
    out, out16 = K.node_chain_x3_fwd(x1, x2 if kind == 0 else None, packs, b1.float(), b2.float(), act, kind, residual=residual, want16=want16)
    return ops._attach_b16(out, out16)


X3_NOGRAD_CHAIN = os.environ.get("FABIND_X3_NOGRAD_CHAIN", "1") == "1"   # 0: no-grad passes of 'bf16x3' run their MLPs as two gemm_x3 launches (A/B)
NODE_CHAIN_X3 = os.environ.get("FABIND_NODE_CHAIN_X3", "1") == "1"     # 0: the split-precision MLPs of no-grad passes as two gemm_x3 launches (A/B)


def gcl_layer(p, h, x, lay, g, clampv, pdrop=0.0):
    """MC_E_GCL.forward (egnn.py:130-144): edge -> coord -> node, all from the layer's input h, x.

    Producers apply the activations once (gcl_pre -> SiLU, GEMM epilogue -> SiLU); under autograd the same
    kernels additionally emit the activation derivative, so every edge-level GEMM (forward and backward)
    streams plain bf16 operands HBM -> LDS.  pdrop > 0 (train mode): dropout on the messages before both
    consumers and on the node-MLP output before the residual (egnn.py:82,106)."""
    H = h.shape[1]
    ad = ops.act_dtype()
    fast = _fast(h, x, p["W2"])
    # bf16x3 with config.set_x3_backward("exact"): a differentiable pass takes the UNFUSED edge pipeline below -- split contractions in the
    # forward, the input gradients AND the weight gradients, fp32 edge tensors: no bf16 rounding anywhere in the edge path's adjoint (the
    # fused backward kernels contract bf16 operands)
    exact_edge = (get_precision() == "bf16x3" and not config_x3_backward_bf16() and not fast
                  and ops.needs_grad(h, x, p["W2"], p["W_ab"]))
    fused_edge = get_precision() in ("bf16", "bf16x3") and FUSED_EDGE and H in (64, 128, 256, 512) and not exact_edge
    if pdrop == 0.0 or fused_edge:  # (the unfused train-mode path has torch-side dropout consumers that do not know the shared buffer)
        h = ops.shared_grad(h)      # three consumers (first edge Linear, node MLP, residual): one gradient buffer, no autograd adds
    hin = _b16(h) if fast else h
    if fused_edge:
        # the whole edge pipeline in one kernel each way, edge tensors stay in LDS (csrc/fused_edge.hip); under
        # autograd nothing per-edge is saved, the backward kernel recomputes tile by tile.  Train-mode dropout on the
        # messages is a counter-based mask evaluated inside both kernels.
        AB = ops.linear(hin, p["W_ab"], p["b_ab"], out_dtype=ad)          # bf16 rows, or fp32 rows for the split-bf16 kernel
        d, rhohat = ops.edge_geom(x, g.row_ctx, g.col_ctx, g.rp_ctx, lay, g.ctx_by_col)
        agg, s = ops.fused_edge(AB, rhohat, p["w_r"], p["W2"], p["b2"], p["Wc"], p["bc"], p["w3"], H, g, pdrop,
                                frags=p.get("_frags") if get_precision() == "bf16" else None)
        x_new = ops.coord_update(x, d, s, g.rp_ctx, mean=True, clampv=clampv)
        if DEBUG_CAPTURE is not None:              # (tools/probes/model_repeat3.py: the pieces of the coordinate update)
            k_ = "gcl_dbg_%d" % sum(1 for q in DEBUG_CAPTURE if q.startswith("gcl_dbg_") and q.endswith(".s"))
            DEBUG_CAPTURE[k_ + ".xin"], DEBUG_CAPTURE[k_ + ".d"], DEBUG_CAPTURE[k_ + ".rhohat"] = x.detach().clone(), d.detach().clone(), rhohat.detach().clone()
            DEBUG_CAPTURE[k_ + ".agg"], DEBUG_CAPTURE[k_ + ".s"] = agg.detach().clone(), s.detach().clone()
        if (get_precision() == "bf16x3" and X3_NOGRAD_CHAIN and pdrop == 0.0 and p["Wn1"].dtype == torch.float32
                and not ops.needs_grad(h, agg, p["Wn1"], p["Wn2"])):
            # round 6: a no-grad pass of the 'bf16x3' mode takes the same split-precision chain kernel (fp32 weights -> hi | lo packs, cached
            # with the no-grad parameter pack): no fp32 hidden layer in HBM
            hn = _node_chain_x3(p, "_ncx3_node", (p["Wn1"], p["Wn2"]), p["bn1"], p["bn2"], K.ACT_SILU, 0, h, agg, h, True)
            if hn is not None:
                return hn, x_new
        if not fast:
            # node MLP + residual as one autograd node (activation adjoint and residual gradient inside GEMM epilogues); train mode: the
            # dropout ahead of the residual (egnn.py:106) inside the second Linear's epilogue, its mask regenerated by the adjoint
            return ops.mlp2(hin, p["Wn1"], p["bn1"], K.ACT_SILU, p["Wn2"], p["bn2"], residual=h, x2=agg, want16=True, p_drop=pdrop,
                            W32=p.get("Wn_32"), chain=p.get("_ncx_node")), x_new
        if fast and pdrop == 0.0 and p.get("Wn_32") is not None:     # split-precision site (level 3): one kernel, the hidden layer on chip
            hn = _node_chain_x3(p, "_ncx_node", p["Wn_32"], p["bn1"], p["bn2"], K.ACT_SILU, 0, h, agg, h, True)
            if hn is not None:
                return hn, x_new
            return ops.mlp2(h, p["Wn1"], p["bn1"], K.ACT_SILU, p["Wn2"], p["bn2"], residual=h, x2=agg, want16=True, W32=p["Wn_32"]), x_new
        if fast and pdrop == 0.0:
            hn = _node_chain(p, "_nc_node", p["Wn1"], p["bn1"], p["Wn2"], p["bn2"], K.ACT_SILU, 0, hin, ops._mm_in(agg), h, True)
            if hn is not None:
                return hn, x_new
        t = ops.linear(hin, p["Wn1"], p["bn1"], x2=agg, act_epi=K.ACT_SILU, out_dtype=ad)
        return ops.linear(t, p["Wn2"], p["bn2"], residual=h, want16=True, p_drop=pdrop), x_new      # (no-grad: epilogue dropout)
    AB = ops.linear(hin, p["W_ab"], p["b_ab"], out_dtype=ad)                           # [N,2H] node-level
    d, rhohat = ops.edge_geom(x, g.row_ctx, g.col_ctx, g.rp_ctx, lay, g.ctx_by_col)
    S1 = ops.gcl_pre(AB, H, g, rhohat, p["w_r"], act=K.ACT_SILU)                        # [E,H] silu(first edge Linear)
    Mm = ops.linear(S1, p["W2"], p["b2"], act_epi=K.ACT_SILU, out_dtype=ad)             # [E,H] messages m_e
    Mm = _drop(Mm, pdrop)
    s, agg = ops.edge_tail(Mm, p["Wc"], p["bc"], p["w3"], g.rp_ctx, g.row_ctx, h.shape[0])   # coord_mlp row-dot, sum_e m_e
    x_new = ops.coord_update(x, d, s, g.rp_ctx, mean=True, clampv=clampv)
    t = ops.linear(hin, p["Wn1"], p["bn1"], x2=agg, act_epi=K.ACT_SILU, out_dtype=ad)
    if pdrop > 0.0:
        h_new = h + _drop(ops.linear(t, p["Wn2"], p["bn2"]), pdrop)
    else:
        h_new = ops.linear(t, p["Wn2"], p["bn2"], residual=h)
    return h_new, x_new


def cross_attention(p, h, lay, pairbias, layer, pdrop=0.0):
    """CrossAttentionModule node path (cross_att.py:38-49).  The protein side (97 % of the nodes) is processed IN PLACE
    in the node layout -- its ligand rows are don't-care (finite, zero gradient) -- only the ligand side is a compact
    [sum C, H] array; the attention kernels address protein rows through the block descriptors.
    pdrop: RowAttentionBlock.dropout on the attention update before the residual (cross_att.py:128)."""
    fast = _fast(h, p["Wo_p"])
    od = ops.act_dtype()
    c16 = (lambda t: _b16(t)) if fast else (lambda t: t)
    h = ops.shared_grad(h)          # consumers: ligand row gather, q / gate projection, residual of the attention update
    hc = ops.take_rows(h, lay.c_index64)
    scale = 1.0 / math.sqrt(32.0)
    # forward-only bf16 passes recompute the pair bias inside the attention kernels (ops.PairBias.fused); everything else reads the
    # [pairs, 8] bias tensors
    # (when the bias tensors exist anyway -- a training call builds them for its differentiable pass -- the no-grad refinement
    #  iterations read them too: recomputing the contraction per iteration measured 8 % slower at n_iter = 8)
    nograd = fast or (get_precision() == "bf16x3" and not ops.needs_grad(h, p["Wo_p"], getattr(pairbias, "a0b0", None)))
    fused = nograd and isinstance(pairbias, ops.PairBias) and pairbias.can_fuse() and not pairbias.has_tensors()
    # the differentiable pass in bf16 mode: the same recomputation in the forward AND the backward kernels (ops._CrossAttnFused)
    ftrain = (not fast) and pdrop >= 0.0 and isinstance(pairbias, ops.PairBias) and not pairbias.has_tensors() and pairbias.can_fuse_train() \
        and ops.needs_grad(h, p["Wo_p"], pairbias.a0b0)
    if not fused and not ftrain:
        pbt = pairbias.tensors() if isinstance(pairbias, ops.PairBias) else pairbias
        bias_p, bias_c = pbt[2 * layer], pbt[2 * layer + 1]
    qg = ops.linear(c16(h), p["Wqg_p"], p["bqg_p"])                                     # [N, 256]
    kv = ops.linear(hc if p.get("Wkv_p32") is not None else c16(hc), p["Wkv_p"], W32=p.get("Wkv_p32"))     # [sum C, 256]
    og = ops.cross_attn_fused(qg, kv, pairbias, 2 * layer, 0, lay, scale) if fused else \
        ops.cross_attn_fused_train(qg, kv, pairbias, 2 * layer, 0, scale) if ftrain else \
        ops.cross_attn(qg, kv, bias_p, 0, 4, lay.desc_pf, lay.B, lay.max_P, lay.max_C, scale)
    # (train mode: RowAttentionBlock's dropout on the attention update ahead of the residual, cross_att.py:128, inside the epilogue -- the
    #  adjoint regenerates its mask; a split-precision site falls back to its bf16 operands then)
    hp = ops.linear(og if p.get("Wo_p32") is not None else c16(og), p["Wo_p"], p["bo_p"], residual=h, want16=True, W32=p.get("Wo_p32"),
                    p_drop=pdrop)
    hp = ops.shared_grad(hp)        # consumers: k / v projection of the ligand-query block, transition, its residual
    hp16 = c16(hp)
    qg = ops.linear(c16(hc), p["Wqg_c"], p["bqg_c"])
    kv = ops.linear(hp if p.get("Wkv_c32") is not None else hp16, p["Wkv_c"], W32=p.get("Wkv_c32"))       # [N, 256], protein rows used
    og = ops.cross_attn_fused(qg, kv, pairbias, 2 * layer + 1, 1, lay, scale) if fused else \
        ops.cross_attn_fused_train(qg, kv, pairbias, 2 * layer + 1, 1, scale) if ftrain else \
        ops.cross_attn(qg, kv, bias_c, 0, 4, lay.desc_cf, lay.B, lay.max_C, lay.max_P, scale)
    hc = ops.linear(og if p.get("Wo_c32") is not None else c16(og), p["Wo_c"], p["bo_c"], residual=hc, p_drop=pdrop, W32=p.get("Wo_c32"))
    if fast:
        hp2 = None
        if p.get("Wt_p32") is not None:                              # split-precision site (level 3)
            hp2 = _node_chain_x3(p, "_ncx_tp", p["Wt_p32"], p["bt1_p"], p["bt2_p"], K.ACT_RELU, 1, hp, None, hp, True)
            if hp2 is None:
                hp2 = ops.mlp2(hp, p["Wt1_p"], p["bt1_p"], K.ACT_RELU, p["Wt2_p"], p["bt2_p"], residual=hp, want16=True, W32=p["Wt_p32"])
        else:
            hp2 = _node_chain(p, "_nc_tp", p["Wt1_p"], p["bt1_p"], p["Wt2_p"], p["bt2_p"], K.ACT_RELU, 1, hp16, None, hp, True)
        if hp2 is None:
            t = ops.linear(hp16, p["Wt1_p"], p["bt1_p"], act_epi=K.ACT_RELU, out_dtype=od)
            hp2 = ops.linear(t, p["Wt2_p"], p["bt2_p"], residual=hp, want16=True)
        hp = hp2
        hc16 = c16(hc)
        if p.get("Wt_c32") is not None:
            hc2 = _node_chain_x3(p, "_ncx_tc", p["Wt_c32"], p["bt1_c"], p["bt2_c"], K.ACT_RELU, 1, hc, None, hc, False)
            if hc2 is None:
                hc2 = ops.mlp2(hc, p["Wt1_c"], p["bt1_c"], K.ACT_RELU, p["Wt2_c"], p["bt2_c"], residual=hc, W32=p["Wt_c32"])
        else:
            hc2 = _node_chain(p, "_nc_tc", p["Wt1_c"], p["bt1_c"], p["Wt2_c"], p["bt2_c"], K.ACT_RELU, 1, hc16, None, hc, False)
        if hc2 is None:
            t = ops.linear(hc16, p["Wt1_c"], p["bt1_c"], act_epi=K.ACT_RELU, out_dtype=od)
            hc2 = ops.linear(t, p["Wt2_c"], p["bt2_c"], residual=hc)
        hc = hc2
    elif (get_precision() == "bf16x3" and X3_NOGRAD_CHAIN and p["Wt1_p"].dtype == torch.float32
          and not ops.needs_grad(hp, hc, p["Wt1_p"], p["Wt2_p"], p["Wt1_c"], p["Wt2_c"])):
        hp2 = _node_chain_x3(p, "_ncx3_tp", (p["Wt1_p"], p["Wt2_p"]), p["bt1_p"], p["bt2_p"], K.ACT_RELU, 1, hp, None, hp, True)
        hp = hp2 if hp2 is not None else ops.mlp2(hp, p["Wt1_p"], p["bt1_p"], K.ACT_RELU, p["Wt2_p"], p["bt2_p"], residual=hp, want16=True)
        hc2 = _node_chain_x3(p, "_ncx3_tc", (p["Wt1_c"], p["Wt2_c"]), p["bt1_c"], p["bt2_c"], K.ACT_RELU, 1, hc, None, hc, False)
        hc = hc2 if hc2 is not None else ops.mlp2(hc, p["Wt1_c"], p["bt1_c"], K.ACT_RELU, p["Wt2_c"], p["bt2_c"], residual=hc)
    else:   # Transition + residual (cross_att.py:48-49) as one autograd node each
        hp = ops.mlp2(hp, p["Wt1_p"], p["bt1_p"], K.ACT_RELU, p["Wt2_p"], p["bt2_p"], residual=hp, want16=True, W32=p.get("Wt_p32"),
                      chain=p.get("_ncx_tp"))
        hc = ops.mlp2(hc, p["Wt1_c"], p["bt1_c"], K.ACT_RELU, p["Wt2_c"], p["bt2_c"], residual=hc, W32=p.get("Wt_c32"), chain=p.get("_ncx_tc"))
    return ops.put_rows(hp, hc, lay.c_index64)


def att_layer(p, h, x, lay, g, a0b0, pairbias, layer, clampv, pdrop=0.0, pdrop_row=0.0):
    """MC_Att_L.forward (egnn.py:308-333).  pdrop: dropout on the aggregated attention message (egnn.py:236)."""
    H = h.shape[1]
    h = cross_attention(p, h, lay, pairbias, layer, pdrop_row)
    if pdrop == 0.0:
        h = ops.shared_grad(h)      # consumers: 32-wide pair projections, q | k | v projection, the attention layer's residual
    # pair embedding at the inter-edge pairs only -> scalar attention bias (egnn.py:208, 286-304)
    h16 = _b16(h) if _fast(h, x, p["Wqkv"]) else h
    ab32 = ops.linear(h16, p["W_ab32"], p["b_ab32"])                                   # [N,128] (a32|0|b32|0)
    hd = ops.pair_hadamard(a0b0, H, ab32, 64, g.red_p, g.red_c, graph=g)               # [n_red, H+64]
    bias_part = ops.linear_rowdot(hd, p["Wcomp1"], p["bcomp1"], p["u"], act_epi=K.ACT_RELU)
    # (split-precision site, config.set_split_sites: the fp32 rows of h and the fp32 master weight instead of their bf16 copies)
    qkv = ops.linear(h if p.get("Wqkv32") is not None else h16, p["Wqkv"], p["bqkv"], want16=True, W32=p.get("Wqkv32"))   # [N,3H] (+ bf16 copy: V is the next operand)
    d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay, g.int_by_col)
    # cv = Linear(V) [N,H] (egnn.py:225) is evaluated inside inter_attn (its adjoint accumulates into dqkv in place)
    # (h is this function's own intermediate -- the row scatter of cross_attention made it -- and with pdrop == 0 nobody reads it after
    #  the call: a no-grad pass lets the kernel update it in place; DEBUG_CAPTURE keeps clones)
    h_new, x_new, alpha = ops.inter_attn(qkv, None, H, h, x, d, rhohat, g, bias_part, p["w_rk"], p["w_rv"], p["wcr"],
                                         p["w3"], clampv, Wc=p["Wc"], bc=p["bc"], own_h=(pdrop == 0.0), Wc32=p.get("Wc32"))
    if pdrop > 0.0:
        h_new = ops.drop_mix(h, h_new, pdrop)          # h + dropout(h_new - h): one pass each way, the mask regenerated by the adjoint
    return h_new, x_new, alpha


def egnn_forward(P, h, x, lay, g, las, x_las, a0b0, pairbias, scale, step, drop=None):
    """MCAttEGNN.forward (egnn.py:392-466).  drop = dict of dropout probabilities (train mode) or None (eval)."""
    clampv = 10.0 / scale
    dp = drop or {}
    h = ops.linear(h, P["W_in"], P["b_in"], want16=True, W32=P.get("W_in32"), p_drop=dp.get("gnn", 0.0))      # (MCAttEGNN.dropout: in the epilogue)
    cap = DEBUG_CAPTURE
    for i in range(P["L"]):
        h, x = gcl_layer(P["gcl"][i], h, x, lay, g, clampv, dp.get("gcl", 0.0))
        if cap is not None:
            cap["gcl_%d.h" % i], cap["gcl_%d.x" % i] = h.detach().clone(), x.detach().clone()
        h, x, alpha = att_layer(P["att"][i], h, x, lay, g, a0b0, pairbias, i, clampv, dp.get("att", 0.0), dp.get("row", 0.0))
        if cap is not None:
            cap["att_%d.h" % i], cap["att_%d.x" % i], cap["att_%d.alpha" % i] = h.detach().clone(), x.detach().clone(), alpha.detach().clone()
        x = ops.las_step(x, x_las, las, lay, step, 15.0 / scale)
        if cap is not None:
            cap["las_%d.xout" % i] = x.detach().clone()
    h, x = gcl_layer(P["out_layer"], h, x, lay, g, clampv, dp.get("out", 0.0))
    return ops.linear(_drop(h, dp.get("gnn", 0.0)), P["W_out"], P["b_out"]), x


def pair_bias_all(P, a0b0, lay):
    """RowAttention pair biases of every layer from z0: a lazy holder -- [pairs, 8] tensors from one ragged-batched K = H contraction
    per block when a consumer asks for them, or the operands of the kernels that recompute the bias on the matrix cores."""
    H = P["H"]
    return ops.PairBias(a0b0, H, P["pb_wcomp"], P["pb_bconst"], lay)


# ------------------------------------------------------------------------------------------------
# input-pipeline half of a stack call, run ahead of it
# ------------------------------------------------------------------------------------------------
_PREFETCHED = []           # [(weakref(X), X._version, weakref(batch_id), layout, graph, event)], most recent first


def _record_streams(obj, stream_):
    """Tell the caching allocator that every tensor held by obj (allocated on a side stream) is used on stream_."""
    for v in vars(obj).values():
        for t in (v if isinstance(v, (tuple, list)) else (v,)):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream_)


def prefetch(model, X, batch_id, segment_id, compound_edge_index):
    """What a stack call on this batch needs BEFORE its first kernel and can only get through host round trips -- the batch layout
    (per-complex node counts read back, index arrays assembled and uploaded) and the ctx / inter graph of the input coordinates (edge
    counts read back to size the edge arrays; att_model.py:209-214) -- built now, on the CURRENT stream.  Call it from the data
    feeder's stream (`with torch.cuda.stream(feeder_stream): engine.prefetch(...)`) while the previous step is still running: the
    round trips then wait for that stream's few kernels only, not for the backlog of the compute stream, and the stack call on the
    same tensor objects (`X`, `batch_id`: identity + version counter) picks both up with a device-side event wait.  Without it every
    step drains the device at its start (the reads are behind the previous step's backward) and the device idles while the host
    assembles the layout and the parameter pack: 5-7 ms of a 100 ms step at the bench shape."""
    import weakref
    # NOT Layout.of: a layout whose index tensors were produced on the feeder stream must reach a consumer only through
    # _take_prefetched (event wait + record_stream); in Layout._cache a call with a modified X -- which misses the prefetch -- would
    # pick it up with neither
    lay = Layout(batch_id, segment_id)
    ex = model.extract_edges
    brow = compound_edge_index[0].to(torch.int32).contiguous()
    bcol = compound_edge_index[1].to(torch.int32).contiguous()
    g = Graph(lay, X.detach().reshape(-1, 3).float().contiguous(), brow, bcol, lay.ranges(brow), float(ex.intra_cutoff),
              float(ex.inter_cutoff))
    g._prefetched_bonds = (brow, bcol)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(X.device))
    _PREFETCHED[:] = [e for e in _PREFETCHED if e[0]() is not None and e[2]() is not None]      # entries of batches that are gone
    _PREFETCHED.insert(0, (weakref.ref(X), X._version, weakref.ref(batch_id), lay, g, ev))
    del _PREFETCHED[4:]
    return lay, g


def _take_prefetched(X, batch_id):
    """-> (layout, graph) prefetched for exactly these tensor objects (and this version of X), or None; the current stream waits
    for the prefetch's event, and the allocator learns that the prefetched tensors are used here."""
    for i, (wx, ver, wb, lay, g, ev) in enumerate(_PREFETCHED):
        if wx() is X and wb() is batch_id and ver == X._version:
            del _PREFETCHED[i]
            cur = torch.cuda.current_stream(X.device)
            cur.wait_event(ev)
            _record_streams(lay, cur)
            _record_streams(g, cur)
            for t in g._prefetched_bonds:
                t.record_stream(cur)
            return lay, g
    return None


class StackContext:
    """Per-batch state shared by every FABind layer of one EfficientMCAttModel call: packed parameters, the batch
    layout, the constant pair-embedding factors (a0 | b0) with the pair biases of every layer, LAS data and -- after
    `rebuild_graph(x)` -- the ctx / inter CSR of the current coordinates.  `EfficientMCAttModel.context(...)` builds
    one; `FABindLayer.forward(h, x, ctx)` / `MCAttEGNN.forward(h, x, ctx=ctx)` consume it."""

    def __init__(self, model, X, Hin, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index, coord_LAS):
        if not X.is_cuda:
            raise RuntimeError("fabind_amd: the docking stack runs on a HIP device only (no CPU fallback); got "
                               + str(X.device))
        args = model.args
        self.model = model
        self.scale = float(args.coordinate_scale)
        self.step = float(args.geometry_reg_step_size)
        self.clampv = 10.0 / self.scale
        self.P = prepare_stack_params(model)
        pre = _take_prefetched(X, batch_id) if _PREFETCHED else None
        self.prefetched_graph = pre[1] if pre is not None else None       # graph of the INPUT coordinates (engine.prefetch)
        self.lay = lay = pre[0] if pre is not None else Layout.of(batch_id, segment_id)
        self.bond_row = compound_edge_index[0].to(torch.int32).contiguous()
        self.bond_col = compound_edge_index[1].to(torch.int32).contiguous()
        self.bond_off = lay.ranges(self.bond_row)
        las = (LAS_edge_index[0].to(torch.int32).contiguous(), LAS_edge_index[1].to(torch.int32).contiguous())
        self.las = las + (lay.ranges(las[0]),)
        self.x_las = coord_LAS.reshape(-1, 3).float().contiguous()
        self.mask_u8 = mask.to(torch.uint8).contiguous()
        self.Hin = Hin.float().contiguous()
        # consumed by the pair biases once and by pair_hadamard in every layer: one shared gradient buffer (ops.GradSink)
        self.a0b0 = ops.shared_grad(ops.linear(self.Hin, self.P["W_ab0"], self.P["b_ab0"]))
        self.pairbias = pair_bias_all(self.P, self.a0b0, lay)
        self.cut_intra = float(model.extract_edges.intra_cutoff)
        self.cut_inter = float(model.extract_edges.inter_cutoff)
        self.drop = None
        if model.training:   # the reference's nn.Dropout modules are active in every refinement iteration (also under no_grad)
            gnn = model.gnn
            drop = dict(gnn=gnn.dropout.p, gcl=gnn.gcl_0.dropout.p, att=gnn.att_0.dropout.p, out=gnn.out_layer.dropout.p,
                        row=gnn.att_0.cross_attn_module.p_attention_block.dropout.p)
            self.drop = drop if any(v > 0 for v in drop.values()) else None
        self.g = None

    def rebuild_graph(self, x, input_coords=False):
        """ctx + inter edges of the coordinates x [N,3] (normalised), att_model.py:209-214.  input_coords: x is the X this context
        was built from, unchanged -- a graph that engine.prefetch built for it is taken instead of building one."""
        pre, self.prefetched_graph = self.prefetched_graph, None
        if input_coords and pre is not None:
            self.g = pre
            return self.g
        self.g = Graph(self.lay, x.detach().reshape(-1, 3).float().contiguous(), self.bond_row, self.bond_col, self.bond_off,
                       self.cut_intra, self.cut_inter)
        return self.g

    # -- one loop body of MCAttEGNN.forward (egnn.py:402-449): gcl_i -> att_i -> LAS step
    def layer(self, i, h, x):
        dp = self.drop or {}
        h, x = gcl_layer(self.P["gcl"][i], h, x, self.lay, self.g, self.clampv, dp.get("gcl", 0.0))
        h, x, alpha = att_layer(self.P["att"][i], h, x, self.lay, self.g, self.a0b0, self.pairbias, i, self.clampv,
                                dp.get("att", 0.0), dp.get("row", 0.0))
        x = ops.las_step(x, self.x_las, self.las, self.lay, self.step, 15.0 / self.scale)
        return h, x, alpha

    def gnn(self, h, x):
        return egnn_forward(self.P, h, x, self.lay, self.g, self.las, self.x_las, self.a0b0, self.pairbias, self.scale,
                            self.step, self.drop)


def stack_forward(model, X, Hin, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                  coord_LAS, n_iter):
    """EfficientMCAttModel.forward with refine='refine_coord' (att_model.py:198-246).

    Mutates X in place like the reference (X[mask] = Z[mask]) and returns (X, H)."""
    ctx = StackContext(model, X, Hin, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index, coord_LAS)
    x = X.reshape(-1, 3).float().contiguous()
    Hout = None
    for r in range(n_iter):
        last = r == n_iter - 1
        g = ctx.rebuild_graph(x, input_coords=(r == 0))
        with torch.set_grad_enabled(last and torch.is_grad_enabled()):
            Hout, z = ctx.gnn(ctx.Hin, x)
            x = ops.select_rows(x, z, ctx.mask_u8)
    model.last_graph = g
    with torch.no_grad():
        X.copy_(x.reshape(X.shape).to(X.dtype))
    Xout = x.reshape(X.shape)
    return (Xout if Xout.requires_grad else X), Hout


class EdgeListGraph:
    """CSR view of a caller-supplied ctx edge list (MC_E_GCL.forward with the reference's signature): edges sorted by
    receiving node (index glue on the device), exposing what gcl_layer reads from a Graph."""

    def __init__(self, edge_index, n_nodes):
        row, col = edge_index[0].long(), edge_index[1].long()
        srow, order = torch.sort(row, stable=True)
        self.row_ctx = srow.to(torch.int32).contiguous()
        self.col_ctx = col[order].to(torch.int32).contiguous()
        rp = torch.zeros(n_nodes + 1, dtype=torch.int32, device=row.device)
        rp[1:] = torch.cumsum(torch.bincount(srow, minlength=n_nodes), 0).to(torch.int32)
        self.rp_ctx, self.N, self.E_ctx, self._bycol = rp, n_nodes, int(row.shape[0]), None

    def ctx_by_col(self):
        if self._bycol is None:
            cs, perm = torch.sort(self.col_ctx.long(), stable=True)
            cp = torch.zeros(self.N + 1, dtype=torch.int32, device=perm.device)
            cp[1:] = torch.cumsum(torch.bincount(cs, minlength=self.N), 0).to(torch.int32)
            self._bycol = (cp, perm.to(torch.int32).contiguous())
        return self._bycol


class BatchOnlyLayout:
    """node_off / B of a complex-contiguous batch vector: all coord2radial's per-sample norm needs."""

    def __init__(self, batch_id):
        B = int(batch_id[-1].item()) + 1
        off = torch.zeros(B + 1, dtype=torch.int32, device=batch_id.device)
        off[1:] = torch.cumsum(torch.bincount(batch_id, minlength=B), 0).to(torch.int32)
        self.node_off, self.B, self.N = off, B, int(batch_id.shape[0])
