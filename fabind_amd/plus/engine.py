"""Host orchestration of the FABind+ layer stack on the HIP kernels (reference FABind_plus/fabind/models/
att_model.py:165-223, egnn.py:104-118, 277-300, 358-433, cross_att.py:20-47, model_utils.py:10-74).

What differs from FABind v1 (fabind_amd/engine.py) and why the data path differs:

* every MLP is LayerNorm -> Linear -> relu -> Linear (-> relu): the first edge Linear can no longer be split per node
  for free, so the edge input [h_r | h_c | rhohat] is layer-normalised while it is gathered (`fabind_edge_ln_concat`)
  and the two edge Linears run as plain GEMMs over [E, 2H+1 padded to a multiple of 8];
* the pair embedding is THREADED through the layers (z_{l+1} = relu-MLP(LN(z_l + I32(p', c')))), so it is
  materialised -- as a ragged [sum_b P_b C_b, H] list without padding (bf16 in bf16 mode), never as the padded
  [B, Pmax, Cmax, H] tensor except for the value the reference's API returns;
* the RowAttention pair biases of a layer are one N=16 GEMM over that list (4 linear + 4 gate heads for each of the two
  blocks); the attention-bias of the inter-edge attention gathers z at the edge pairs.

The kernels are the generic ones (GEMM, row LayerNorm, cross attention, inter-edge attention, CSR reductions) plus, in
bf16 inference / sampling, one fused pair-update kernel per layer (csrc/pair_fused.hip); under autograd the pair update
runs as separate Hadamard / GEMM / LayerNorm launches."""
import math
import os

import numpy as np
import torch

from .. import kernels as K
from .. import ops
from ..config import fp32_storage, get_precision
from ..engine import Graph, Layout, _cat, cached_pack
from ..param_pack import EagerPack, ParamPack


def _wd():
    return torch.float32 if fp32_storage() else torch.bfloat16


def _pad8(n):
    """Contraction dims are padded to a multiple of 64 (zero weights / zero columns): the LDS-DMA pipelined GEMM needs
    K % 64 == 0, the 2H+1 = 1025-wide edge MLP would otherwise fall back to the register-staged kernel."""
    return (n + 63) // 64 * 64


def _padded(w, rows, cols):
    out = torch.zeros((rows, cols), dtype=w.dtype, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _padvec(v, n):
    out = torch.zeros(n, dtype=v.dtype, device=v.device)
    out[:v.shape[0]] = v
    return out


# ------------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------------
def _mlp(m, pk, k_pad=None, n_pad=None, w2=True):
    """LN-MLP pack: LayerNorm (w, b), linear1 padded to [n_pad, k_pad], linear2 padded to [., n_pad].  pk: the pack collecting the
    copy requests (param_pack.ParamPack: one launch for the whole model; EagerPack: torch ops)."""
    wd = _wd()
    w1 = m.linear1.weight
    k_pad = k_pad or w1.shape[1]
    n_pad = n_pad or w1.shape[0]
    has_ln = hasattr(m, "layernorm")
    return dict(ln_w=m.layernorm.weight if has_ln else None, ln_b=m.layernorm.bias if has_ln else None,
                C=w1.shape[1], k_pad=k_pad,
                W1=pk.pad2d(w1, n_pad, k_pad, wd, with_T=True), b1=pk.cat([m.linear1.bias, pk.zeros(n_pad - w1.shape[0])]),
                # (w2 = False: the caller requests linear2's one row itself -- a parameter element may sit in ONE request of a pack)
                W2=pk.pad2d(m.linear2.weight, m.linear2.weight.shape[0], n_pad, wd, with_T=True) if w2 else None,
                b2=None if m.linear2.bias is None else pk.copy(m.linear2.bias))


def prepare_stack_params(model):
    return cached_pack(model, _build_stack_params)


def _gcl_pack(m, H, pk=None):
    """Kernel-side parameters of one MC_E_GCL (edge / node / coord LN-MLPs; in bf16 inference also the LayerNorm-folded
    forms).  pk: see _mlp (default: torch ops, resolved here)."""
    wd = _wd()
    infer = not torch.is_grad_enabled()          # cached_pack builds the no-autograd pack under no_grad: inference-only extras
    own = pk is None
    pk = pk or EagerPack(m.node_mlp.linear2.weight.device)
    W = lambda t: t.to(wd).contiguous()
    K8 = _pad8(2 * H + 1)
    d = dict(edge=_mlp(m.edge_mlp, pk, K8, K8), node=_mlp(m.node_mlp, pk), coord=_mlp(m.coord_mlp, pk, w2=False))
    d["coord"]["w3"] = pk.copy(m.coord_mlp.linear2.weight[0])
    if wd == torch.bfloat16 and infer and H % 64 == 0:
        # coord_mlp on the messages: LayerNorm folded into the GEMM epilogue (FabindGemmArgs.row_mu / row_rs / col_c)
        cm = m.coord_mlp
        Wc = cm.linear1.weight.float() * cm.layernorm.weight.float()[None, :]
        d["coord"]["foldp"] = dict(W1w=W(Wc), cvec=Wc.to(wd).float().sum(1).contiguous(), eps=float(cm.layernorm.eps),
                                   dvec=(cm.linear1.weight.float() @ cm.layernorm.bias.float() + cm.linear1.bias.float()).contiguous())
    if wd == torch.bfloat16 and infer:
        # LayerNorm folded into per-node projections of the first edge Linear (csrc/norm.hip: edge_lnfold_kernel); under autograd the
        # same operands are built per call by _fold_train below
        em = m.edge_mlp
        W1w = _padded(em.linear1.weight.float() * em.layernorm.weight.float()[None, :], K8, 2 * H + 1)
        d["fold"] = dict(W_ab=W(_cat([W1w[:, :H], W1w[:, H:2 * H]])), w_r=W1w[:, 2 * H].contiguous(),
                         c_r=W1w[:, :H].sum(1).contiguous(), c_c=W1w[:, H:2 * H].sum(1).contiguous(),
                         eps=float(em.layernorm.eps),
                         dvec=_padvec(em.linear1.weight.float() @ em.layernorm.bias.float() + em.linear1.bias.float(), K8))
    return pk.resolve(d) if own else d


def _cam_pack(cam, H, pk=None):
    """Kernel-side parameters of one CrossAttentionModule: both RowAttentionBlocks, the three LN-MLP transitions, the
    32-wide Hadamard pair update.  pk: see _mlp."""
    wd = _wd()
    infer = not torch.is_grad_enabled()
    own = pk is None
    pk = pk or EagerPack(cam.inter_layer.linear_p.weight.device)
    d = {}
    for tag, blk in (("p", cam.p_attention_block), ("c", cam.c_attention_block)):
        a = blk.mha
        d["Wqg_" + tag] = pk.cat([a.linear_q.weight, a.linear_g.weight], 0, wd, with_T=True)
        d["bqg_" + tag] = pk.cat([pk.zeros(a.linear_g.bias.shape[0]), a.linear_g.bias])
        d["Wkv_" + tag] = pk.cat([a.linear_k.weight, a.linear_v.weight], 0, wd, with_T=True)
        d["Wo_" + tag], d["bo_" + tag] = pk.copy(a.linear_o.weight, wd, with_T=True), pk.copy(a.linear_o.bias)
    # pair biases of both blocks from z in one GEMM: columns [p lin 0-3 | p gate 4-7 | c lin 8-11 | c gate 12-15]
    pb, cb = cam.p_attention_block, cam.c_attention_block
    d["W_pb"] = pk.cat([pb.linear.weight, pb.linear_g.weight, cb.linear.weight, cb.linear_g.weight], 0, wd, with_T=True)
    d["b_pb"] = pk.cat([pb.linear.bias, pb.linear_g.bias, cb.linear.bias, cb.linear_g.bias])
    d["tr_p"], d["tr_c"] = _mlp(cam.p_transition, pk), _mlp(cam.c_transition, pk)
    d["tr_z"] = _mlp(cam.pair_transition, pk)
    i32 = cam.inter_layer
    # the 32-wide Hadamard block is zero-padded to 64 (a32 | 0 | b32 | 0) so that the contraction with W_o32 runs on the
    # LDS-DMA GEMM path (K % 64 == 0); the padded columns are exact zeros
    n32, Hn = i32.linear_p.weight.shape
    d["W_ab32"] = pk.cat([i32.linear_p.weight, pk.zeros(n32, Hn), i32.linear_c.weight, pk.zeros(n32, Hn)], 0, wd, with_T=True)  # [128, H]
    d["b_ab32"] = pk.cat([i32.linear_p.bias, pk.zeros(n32), i32.linear_c.bias, pk.zeros(n32)])
    wo = i32.linear_out.weight
    d["W_o32"] = pk.cat([wo, pk.zeros(wo.shape[0], wo.shape[1])], 1, wd, with_T=True)                                        # [H, 64]
    d["b_o32"] = pk.copy(i32.linear_out.bias)
    if own:
        d = pk.resolve(d)
    if wd == torch.bfloat16 and infer and H in (64, 128, 256, 512) and own:
        # fragment-packed operands of the fused pair-update kernel (csrc/pair_fused.hip, inference)
        tz = cam.pair_transition
        d["pair_fused"] = dict(Wop=K.pack_frag(i32.linear_out.weight), bo=i32.linear_out.bias.float().contiguous(),
                               ln_w=tz.layernorm.weight.float().contiguous(), ln_b=tz.layernorm.bias.float().contiguous(),
                               eps=float(tz.layernorm.eps),
                               W1p=K.pack_frag(tz.linear1.weight), b1=tz.linear1.bias.float().contiguous(),
                               W2p=K.pack_frag(tz.linear2.weight), b2=tz.linear2.bias.float().contiguous(),
                               Wbp=K.pack_frag(d["W_pb"]), bb=d["b_pb"].float().contiguous())
    return d


def _att_pack(m, H, pk=None):
    """Kernel-side parameters of one MC_Att_L: its CrossAttentionModule (_cam_pack) and the inter-edge attention.  pk: see _mlp (with
    a shared pack the inference-only extras are left out: shared packs serve the differentiable path)."""
    wd = _wd()
    infer = not torch.is_grad_enabled()
    own = pk is None
    pk = pk or EagerPack(m.linear_kv.weight.device)
    W = lambda t: t.to(wd).contiguous()
    d = _cam_pack(m.cross_attn_module, H, pk if not own else None)
    Wkv, bkv = m.linear_kv.weight, m.linear_kv.bias
    d2 = {}
    d2["Wqkv"] = pk.cat([m.linear_q.weight, Wkv[0::2, 1:], Wkv[1::2, 1:]], 0, wd, with_T=True)
    d2["bqkv"] = pk.cat([m.linear_q.bias, bkv[0::2], bkv[1::2]])
    d2["w_rk"], d2["w_rv"] = pk.copy(Wkv[0::2, 0]), pk.copy(Wkv[1::2, 0])
    d2["W_bias"] = pk.cat([m.attn_bias_proj.weight[0:1], pk.zeros(7, H)], 0, wd, with_T=True)
    d2["b_bias"] = pk.cat([m.attn_bias_proj.bias, pk.zeros(8 - m.attn_bias_proj.bias.shape[0])])
    d2["coord"] = _mlp(m.coord_mlp, pk, w2=False)
    d2["coord"]["w3"] = pk.copy(m.coord_mlp.linear2.weight[0])
    if own:
        d2 = pk.resolve(d2)
    d.update(d2)
    if wd == torch.bfloat16 and infer and H % 8 == 0 and H <= 512 and own:
        # coord_mlp's LayerNorm folded into a per-node projection of V (csrc/norm.hip: inter_coord_fold_kernel)
        cm = m.coord_mlp
        W1w = cm.linear1.weight.float() * cm.layernorm.weight.float()[None, :]
        wc = d["w_rv"] - d["w_rv"].mean()
        d["coord_fold"] = dict(W1w=W(W1w), wc=wc.contiguous(), q_w=float((wc * wc).sum()), u=(W1w @ wc).contiguous(),
                               d=(cm.linear1.weight.float() @ cm.layernorm.bias.float() + cm.linear1.bias.float()).contiguous(),
                               eps=float(cm.layernorm.eps), w3=d["coord"]["w3"])
    d["zeroH"] = _zeros_const(H, Wkv.device)
    return d


_ZEROS = {}


def _zeros_const(n, dev):
    """A shared read-only zero vector (the unused per-node bias operands of the inter-edge attention kernels)."""
    z = _ZEROS.get((n, dev))
    if z is None:
        z = _ZEROS[(n, dev)] = torch.zeros(n, dtype=torch.float32, device=dev)
    return z


def gcl_params(m):
    """Pack of a stand-alone MC_E_GCL (its reference-signature forward, fabind_amd/plus/dense.py)."""
    return cached_pack(m, lambda mm: _gcl_pack(mm, mm.node_mlp.linear2.weight.shape[0]))


def cam_params(cam):
    """Pack of a stand-alone CrossAttentionModule (reference-signature forward, fabind_amd/plus/dense.py)."""
    return cached_pack(cam, lambda c: _cam_pack(c, c.pair_hidden_dim))


def att_params(m):
    """Pack of a stand-alone MC_Att_L / its CrossAttentionModule (reference-signature forwards, fabind_amd/plus/dense.py)."""
    return cached_pack(m, lambda mm: _att_pack(mm, mm.hidden_nf))


_PLAN_CACHE = {}           # id(model) -> (key, ParamPack with its requests, request tree, weakref(model))
PLUS_PARAM_PACK = os.environ.get("FABIND_PLUS_PARAM_PACK", "1") == "1"     # training: one-launch differentiable pack (0: torch ops per entry)


def _stack_requests(model, pk):
    gnn = model.gnn
    H, L = gnn.hidden_nf, gnn.n_layers
    il = model.inter_layer
    wd = _wd()
    P = {"H": H, "L": L}
    P["W_ab0"] = pk.cat([il.linear_p.weight, il.linear_c.weight], 0, wd, with_T=True)
    P["b_ab0"] = pk.cat([il.linear_p.bias, il.linear_c.bias])
    P["W_o0"], P["b_o0"] = pk.copy(il.linear_out.weight, wd, with_T=True), pk.copy(il.linear_out.bias)
    P["W_in"], P["b_in"] = pk.copy(gnn.linear_in.weight, wd, with_T=True), pk.copy(gnn.linear_in.bias)
    P["W_out"], P["b_out"] = pk.copy(gnn.linear_out.weight, wd, with_T=True), pk.copy(gnn.linear_out.bias)
    P["gcl"] = [_gcl_pack(getattr(gnn, "gcl_%d" % i), H, pk) for i in range(L)]
    P["out_layer"] = _gcl_pack(gnn.out_layer, H, pk)
    P["att"] = [_att_pack(getattr(gnn, "att_%d" % i), H, pk) for i in range(L)]
    return P


def _build_stack_params(model):
    """Without autograd (inference / sampling: built once per weight version by cached_pack): torch ops per entry + the inference-only
    folded / fragment-packed forms.  Under autograd: every entry is a copy-type request (slices, concatenations, zero padding, casts,
    transposes) of ONE param_pack.ParamPack -- one launch each way instead of ~700 torch launches per model and step -- whose requests
    are planned once per model and re-run every step (v1: engine._build_stack_params)."""
    gnn = model.gnn
    H, L = gnn.hidden_nf, gnn.n_layers
    if not (torch.is_grad_enabled() and PLUS_PARAM_PACK and os.environ.get("FABIND_PARAM_PACK", "1") != "0"):
        P = {"H": H, "L": L}
        wd = _wd()
        W = lambda t: t.to(wd).contiguous()
        il = model.inter_layer
        P["W_ab0"] = W(_cat([il.linear_p.weight, il.linear_c.weight]))
        P["b_ab0"] = _cat([il.linear_p.bias, il.linear_c.bias]).contiguous()
        P["W_o0"], P["b_o0"] = W(il.linear_out.weight), il.linear_out.bias
        P["W_in"], P["b_in"] = W(gnn.linear_in.weight), gnn.linear_in.bias
        P["W_out"], P["b_out"] = W(gnn.linear_out.weight), gnn.linear_out.bias
        P["gcl"] = [_gcl_pack(getattr(gnn, "gcl_%d" % i), H) for i in range(L)]
        P["out_layer"] = _gcl_pack(gnn.out_layer, H)
        P["att"] = [_att_pack(getattr(gnn, "att_%d" % i), H) for i in range(L)]
        if torch.is_grad_enabled():
            _fold_train(P, model)
        return P
    from .. import engine as _e
    params = _e.last_params_of(model)
    key = (_wd(), tuple((id(p_), p_.data_ptr(), p_.requires_grad) for p_ in params))
    ent = _PLAN_CACHE.get(id(model))
    if ent is not None and ent[0] == key and ent[3]() is model and _e.PACK_PLAN:
        pk, tree = ent[1], ent[2]
    else:
        import weakref
        pk = ParamPack(model.inter_layer.linear_p.weight.device)
        tree = _stack_requests(model, pk)
        _PLAN_CACHE[id(model)] = (key, pk, tree, weakref.ref(model, lambda _r, k=id(model): _PLAN_CACHE.pop(k, None)))
    P = pk.resolve(tree)
    _fold_train(P, model)
    return P


def _fold_train(P, model):
    """The LayerNorm-folded operands of every MC_E_GCL's first edge Linear UNDER AUTOGRAD (round 5: _EdgeLnFold), as differentiable torch
    ops on the modules' own fp32 parameters, all layers at once (the six edge MLPs have one shape: a dozen batched launches per model call
    each way): W_ab = [W1 diag(ln_w)]_{r|c} (bf16, rows zero-padded to the kernel's K8), w_r = its rho column, c_r / c_c = the row sums of
    the two blocks, dvec = W1 ln_b + b1.  linear1 / LayerNorm of the edge MLPs reach the kernels ONLY through this composition in that
    mode (their parameter-pack entries are unused and receive no gradient from the pack's adjoint)."""
    if not (FOLD_EDGE_LN_TRAIN and FOLD_EDGE_LN and _wd() == torch.bfloat16):
        return
    gnn = model.gnn
    H, L = P["H"], P["L"]
    mods = [getattr(gnn, "gcl_%d" % i) for i in range(L)] + [gnn.out_layer]
    C, K8 = 2 * H + 1, _pad8(2 * H + 1)
    if K8 > 1536 or K8 % 16 != 0 or any(m.edge_mlp.linear1.weight.shape != (C, C) for m in mods):
        return
    F = torch.nn.functional
    W1 = torch.stack([m.edge_mlp.linear1.weight for m in mods]).float()                      # [L + 1, C, C]
    lw = torch.stack([m.edge_mlp.layernorm.weight for m in mods]).float()
    lb = torch.stack([m.edge_mlp.layernorm.bias for m in mods]).float()
    b1 = torch.stack([m.edge_mlp.linear1.bias for m in mods]).float()
    W1w = F.pad(W1 * lw[:, None, :], (0, 0, 0, K8 - C))                                        # [L + 1, K8, C]: rows padded with zeros
    W_ab = torch.cat([W1w[:, :, :H], W1w[:, :, H:2 * H]], 1).to(torch.bfloat16).contiguous()   # [L + 1, 2 K8, H]
    w_r = W1w[:, :, 2 * H].contiguous()
    c_r, c_c = W1w[:, :, :H].sum(2), W1w[:, :, H:2 * H].sum(2)
    dvec = F.pad(torch.bmm(W1, lb[:, :, None]).squeeze(2) + b1, (0, K8 - C))
    for i, (d_, wab, wr, cr, cc, dv) in enumerate(zip(list(P["gcl"]) + [P["out_layer"]], W_ab.unbind(0), w_r.unbind(0), c_r.unbind(0),
                                                        c_c.unbind(0), dvec.unbind(0))):
        d_["fold"] = dict(W_ab=wab, w_r=wr.contiguous(), c_r=cr.contiguous(), c_c=cc.contiguous(), dvec=dv.contiguous(),
                          eps=float(mods[i].edge_mlp.layernorm.eps))


# ------------------------------------------------------------------------------------------------
# pair list: every (protein row, ligand row) of every complex, complex-major, protein-major inside a complex --
# the order the block descriptors of the attention kernels use (pair = pair_off[b] + p_local * C_b + c_local)
# ------------------------------------------------------------------------------------------------
class PairList:
    """Built with device index arithmetic (no per-complex host loop, no large upload)."""

    def __init__(self, lay, device):
        from ..param_pack import upload
        i64 = lambda a: upload(np.ascontiguousarray(a, dtype=np.int64), device, torch.int64)         # (asynchronous: no queue drain)
        packed = i64(np.stack([lay.off[:-1], lay.C, lay.pair_off_np[:-1], lay.P * lay.C]))       # one small upload
        self.off, self.C, self.pair_off, npair = packed[0], packed[1], packed[2], packed[3]
        self.n = int(lay.n_pairs)
        self.n_inner = int(((lay.P - 1) * (lay.C - 1)).sum())        # pairs of non-global nodes (i >= 1, j >= 1): a size the host knows
        self.b = torch.repeat_interleave(torch.arange(lay.B, device=device), npair, output_size=self.n)
        loc = torch.arange(self.n, device=device) - self.pair_off[self.b]
        Cb, ob = self.C[self.b], self.off[self.b]
        self.i = torch.div(loc, Cb, rounding_mode="floor")
        self.j = loc - self.i * Cb
        self.p_node = (ob + Cb + self.i).to(torch.int32)
        self.c_node = (ob + self.j).to(torch.int32)

    def index_of(self, p_node, c_node, batch_id):
        """Row of the pair list for (protein node, ligand node) ids (index arithmetic on small integer arrays)."""
        b = batch_id[p_node.long()]
        return self.pair_off[b] + (p_node.long() - self.off[b] - self.C[b]) * self.C[b] + (c_node.long() - self.off[b])

    def dense(self, z, lay):
        """[B, Pmax, Cmax, H] zero-padded copy (the value the reference's EfficientMCAttModel returns)."""
        out = torch.zeros((lay.B, lay.max_P, lay.max_C, z.shape[1]), dtype=torch.float32, device=z.device)
        out[self.b, self.i, self.j] = z.float()
        return out


def _pair_hadamard_call(T, Hh, p_node, c_node, out_dtype):
    """hd[e, :] = T[p_node[e], :Hh] * T[c_node[e], Hh:2Hh]  (InteractionModule's outer product on a pair list)."""
    from .._lib import check, dt_code, load, ptr, stream
    n = p_node.shape[0]
    hd = torch.empty((n, Hh), dtype=out_dtype, device=T.device)
    a0, b0 = T[:, :Hh], T[:, Hh:2 * Hh]
    check(load().fabind_pair_hadamard(ptr(a0), ptr(b0), T.stride(0), Hh, ptr(a0), ptr(b0), T.stride(0), 0, ptr(p_node),
                                      ptr(c_node), n, ptr(hd), dt_code(out_dtype), Hh, stream()), "fabind_pair_hadamard")
    return hd


# ------------------------------------------------------------------------------------------------
# differentiable pieces specific to the FABind+ data path (the rest comes from fabind_amd.ops)
# ------------------------------------------------------------------------------------------------
class _LnRows(torch.autograd.Function):
    """Row LayerNorm (any dtype / leading dimension, zero padding to pad_to) with a HIP adjoint."""

    @staticmethod
    def forward(ctx, x, w, b, out_dtype, pad_to):
        ctx.save_for_backward(x, w)
        return K.layernorm_rows(x, w, b, out_dtype, pad_to)

    @staticmethod
    def backward(ctx, dy):
        from .._lib import check, dt_code, load, ptr, stream
        x, w = ctx.saved_tensors
        R, C = x.shape
        dy = dy if dy.stride(-1) == 1 else dy.contiguous()
        C8 = (C + 7) // 8 * 8                               # rows padded to 8 elements: the adjoint kernel's 16-byte accesses (csrc/norm.hip)
        dx = torch.empty((R, C8), dtype=x.dtype, device=x.device)[:, :C]
        nblk = max(1, min((R + 3) // 4, 2048))              # work-groups of four waves, rows strided over them: 8 per CU
        dwp = torch.empty((nblk, C), dtype=torch.float32, device=x.device)
        dbp = torch.empty((nblk, C), dtype=torch.float32, device=x.device)
        check(load().fabind_layernorm_rows_bwd(ptr(x), dt_code(x.dtype), x.stride(0), ptr(w), ptr(dy), dt_code(dy.dtype),
                                               dy.stride(0), 1e-5, R, C, ptr(dx), dt_code(dx.dtype), C8, ptr(dwp), ptr(dbp),
                                               nblk, stream()), "fabind_layernorm_rows_bwd")
        return dx, K.colsum(dwp), K.colsum(dbp), None, None


def ln_rows(x, w, b, out_dtype, pad_to=None):
    pad_to = pad_to or x.shape[1]
    if ops.needs_grad(x, w, b):
        return _LnRows.apply(x, w, b, out_dtype, pad_to)
    return K.layernorm_rows(x, w, b, out_dtype, pad_to)


class _EdgeConcat(torch.autograd.Function):
    """[h[row] | h[col] | rhohat | 0...] per edge; adjoint = two CSR segment sums (by receiving node, by sending node
    through the by-sender permutation) + a column read."""

    @staticmethod
    def forward(ctx, h, rhohat, row, col, rowptr, bycol, out_dtype, pad_to):
        from .._lib import check, dt_code, load, ptr, stream
        E, H = row.shape[0], h.shape[1]
        y = torch.empty((E, pad_to), dtype=out_dtype, device=h.device)
        check(load().fabind_edge_concat(ptr(h), h.stride(0), H, ptr(row), ptr(col), ptr(rhohat), E, ptr(y), dt_code(out_dtype),
                                        pad_to, pad_to, stream()), "fabind_edge_concat")
        ctx.n, ctx.H, ctx.bycol = h.shape[0], H, bycol
        ctx.save_for_backward(rowptr)
        return y

    @staticmethod
    def backward(ctx, dy):
        rowptr, = ctx.saved_tensors
        H, n = ctx.H, ctx.n
        dy = dy.contiguous()
        colptr, perm = ctx.bycol()
        dh = K.segment_sum(dy[:, :H], rowptr, n)
        dh = dh + K.segment_sum(dy[:, H:2 * H], colptr, n, eidx=perm)
        return dh, dy[:, 2 * H].float(), None, None, None, None, None, None


class _PairHad1(torch.autograd.Function):
    """hd[e] = T[p_node[e], :Hh] * T[c_node[e], Hh:2Hh] with the atomic adjoint kernel of the v1 path."""

    @staticmethod
    def forward(ctx, T, Hh, p_node, c_node, out_dtype, lay=None):
        ctx.Hh, ctx.lay = Hh, lay
        ctx.save_for_backward(T, p_node, c_node)
        return _pair_hadamard_call(T, Hh, p_node, c_node, out_dtype)

    @staticmethod
    def backward(ctx, dhd):
        from .._lib import check, dt_code, load, ptr, stream
        T, p_node, c_node = ctx.saved_tensors
        Hh, lay = ctx.Hh, ctx.lay
        dhd = dhd.contiguous()
        dT = torch.zeros_like(T)
        if (lay is not None and PAIRHAD_GRID and T.dtype == torch.float32 and p_node.shape[0] == lay.n_pairs and Hh % 4 == 0
                and (Hh >= 256 or 256 % Hh == 0) and T.stride(0) % 4 == 0 and dhd.stride(0) % 4 == 0):
            # the list is the batch's full protein x ligand grid: deterministic row walk instead of float atomics (csrc/bwd.hip)
            nb = getattr(lay, "_node_b", None)
            if nb is None:
                nb = lay._node_b = torch.repeat_interleave(torch.arange(lay.B, dtype=torch.int32, device=T.device),
                                                           (lay.node_off[1:] - lay.node_off[:-1]).long(), output_size=lay.N)
            check(load().fabind_pair_hadamard_bwd_grid(ptr(dhd), dt_code(dhd.dtype), dhd.stride(0), ptr(T), T.stride(0), Hh,
                                                       ptr(lay.node_off), ptr(lay.c_cnt), ptr(nb), ptr(lay.desc_p), lay.N, ptr(dT),
                                                       dT.stride(0), stream()), "fabind_pair_hadamard_bwd_grid")
            return dT, None, None, None, None, None
        a0, b0 = T[:, :Hh], T[:, Hh:2 * Hh]
        da, db = dT[:, :Hh], dT[:, Hh:2 * Hh]
        check(load().fabind_pair_hadamard_bwd(ptr(dhd), dt_code(dhd.dtype), dhd.stride(0), ptr(a0), ptr(b0), T.stride(0), Hh,
                                              ptr(a0), ptr(b0), T.stride(0), 0, ptr(p_node), ptr(c_node), p_node.shape[0], ptr(da),
                                              ptr(db), dT.stride(0), ptr(da), ptr(db), dT.stride(0), stream()),
              "fabind_pair_hadamard_bwd")
        return dT, None, None, None, None, None


PAIRHAD_GRID = os.environ.get("FABIND_PAIRHAD_GRID", "1") == "1"     # adjoint of the all-pairs Hadamard as a row walk (no float atomics)


def pair_had(T, Hh, p_node, c_node, out_dtype, lay=None):
    """lay: the batch layout when (p_node, c_node) is its complete pair list (PairList order) -- selects the atomics-free adjoint."""
    if ops.needs_grad(T):
        return _PairHad1.apply(T, Hh, p_node, c_node, out_dtype, lay)
    return _pair_hadamard_call(T, Hh, p_node, c_node, out_dtype)


class _EdgeLnFold(torch.autograd.Function):
    """relu(W1 LN([h_r | h_c | rho]) + b1) (+ dropout) of the edge LN-MLP from per-node projections AB and per-node statistics, under
    autograd (round 5; FABind_plus egnn.py:52-58, model_utils.py:44-46): forward = the inference kernel (csrc/norm.hip: edge_lnfold), a
    gather; backward = edge_lnfold_bwd (one pass over the [E, Kp] gradient: mask, rs, the four per-edge row-dots, the column sums of the
    four vectors) + the two segment sums the concatenation's adjoint ran anyway (receiving side, sending side through the by-sender
    permutation) + segment sums of eight floats per edge for the statistics.  The [E, 2H+1] concatenation, its LayerNorm, the
    E x (2H+1) x (2H+1) contraction, its input-gradient GEMM and its weight-gradient contraction do not exist: the weight gradient is a
    NODE-level contraction (autograd of AB = hcen W_ab^T)."""

    @staticmethod
    def forward(ctx, AB, stat, rhohat, w_r, c_r, c_c, dvec, g, H, Kp, eps, pd, seed):
        t = K.edge_lnfold(AB, Kp, H, g.row_ctx, g.col_ctx, rhohat, stat, eps, w_r, c_r, c_c, dvec, pd, seed)
        ctx.g, ctx.H, ctx.Kp, ctx.eps, ctx.pd = g, H, Kp, eps, pd
        ctx.save_for_backward(AB, stat, rhohat, w_r, c_r, c_c, t)
        return t

    @staticmethod
    def backward(ctx, dt):
        AB, stat, rhohat, w_r, c_r, c_c, t = ctx.saved_tensors
        g, H, Kp = ctx.g, ctx.H, ctx.Kp
        N = AB.shape[0]
        dt = dt.contiguous() if dt.dtype == torch.bfloat16 else dt.to(torch.bfloat16).contiguous()
        du, es, drho, vecs = K.edge_lnfold_bwd(AB, Kp, H, g.row_ctx, g.col_ctx, rhohat, stat, ctx.eps, w_r, c_r, c_c, t, dt, ctx.pd)
        colptr, perm = g.ctx_by_col()
        dAB = torch.empty((N, 2 * Kp), dtype=torch.bfloat16, device=AB.device)
        half = (Kp // 2 + 7) // 8 * 8                          # (the segment-sum kernel takes <= 1024 columns: two column ranges per side)
        for lo, hi in ((0, half), (half, Kp)):
            K.segment_sum(du[:, lo:hi], g.rp_ctx, N, out16=dAB[:, lo:hi])
            K.segment_sum(du[:, lo:hi], colptr, N, eidx=perm, out16=dAB[:, Kp + lo:Kp + hi])
        by_r = K.segment_sum(es, g.rp_ctx, N)                  # [N, 8]: columns 0, 1 = d (m, Q) through the edges a node RECEIVES
        by_c = K.segment_sum(es, colptr, N, eidx=perm)         #         columns 2, 3 = through the edges it SENDS
        dstat = torch.stack([by_r[:, 0] + by_c[:, 2], by_r[:, 1] + by_c[:, 3]], 1)
        return dAB, dstat, drho, vecs[3], vecs[1], vecs[2], vecs[0], None, None, None, None, None, None


class _InterView:
    """The inter graph under the attribute names ops.gcl_pre reads (row_ctx / col_ctx / rp_ctx / ctx_by_col)."""

    def __init__(self, g):
        self.row_ctx, self.col_ctx, self.rp_ctx, self.ctx_by_col = g.row_int, g.col_int, g.rp_int, g.int_by_col


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
MLP2_NODE = os.environ.get("FABIND_PLUS_MLP2_NODE", "1") == "1"      # training: the two Linears of an LN-MLP as one autograd node (ops._MLP2Relu)
Z_SINK = os.environ.get("FABIND_PLUS_Z_SINK", "1") == "1"             # training: shared gradient buffer for the pair embedding of every layer
ROWDOT_DROP_GRAD = os.environ.get("FABIND_PLUS_ROWDOT_DROP", "1") == "1"   # training: coord-MLP row-dot with its dropout inside the GEMM epilogue
EPI_DROP_GRAD = os.environ.get("FABIND_EPI_DROP_GRAD", "1") == "1"   # training: ReLU + dropout inside the GEMM epilogue (no torch mask kernels)


def _drop(t, pr):
    """nn.Dropout of the reference modules (train mode = FABind+ sampling inference): element-wise mask + scale between
    kernels, torch's generator."""
    return torch.nn.functional.dropout(t, pr, True) if pr > 0.0 else t


def ln_mlp(m, x, last_act, residual=None, out_dtype=torch.float32, pdrop=0.0):
    """LN -> linear1 -> relu -> [dropout] -> linear2 (-> relu -> [dropout]) (+ residual).  model_utils.py:10-74.
    No-grad: the dropouts run inside the GEMM epilogues (before the residual); under autograd they are torch masks."""
    ad = ops.act_dtype()
    y = ln_rows(x, m["ln_w"], m["ln_b"], ad, m["k_pad"]) if m["ln_w"] is not None else x
    act2 = K.ACT_RELU if last_act else K.ACT_NONE
    if (MLP2_NODE and (pdrop == 0.0 or EPI_DROP_GRAD) and ops.needs_grad(y, m["W1"], m["W2"], residual)
            and y.dtype == torch.bfloat16 and m["W1"].dtype == torch.bfloat16 and m["b1"] is not None and m["b2"] is not None
            and m["W1"].shape[0] % 8 == 0 and m["W2"].shape[0] % 8 == 0):
        # both Linears as one autograd node (ops._MLP2Relu); relu + dropout ahead of a residual: the node returns the dropped
        # activation, the residual is added here
        if last_act and residual is not None:
            y2 = ops.mlp2_relu(y, m["W1"], m["b1"], m["W2"], m["b2"], True, None, torch.float32, pdrop)
            return (residual + y2).to(out_dtype)
        return ops.mlp2_relu(y, m["W1"], m["b1"], m["W2"], m["b2"], last_act, residual, out_dtype, pdrop)
    if pdrop > 0.0 and ops.needs_grad(y, m["W1"], m["W2"], residual):
        # under autograd the ReLU + dropout pairs run inside the GEMM epilogue too (ops._Linear: the zeros of the saved output are the
        # dropped positions); only a dropout that is followed by a residual keeps the torch mask
        t = ops.linear(y, m["W1"], m["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pdrop if EPI_DROP_GRAD else 0.0)
        t = t if EPI_DROP_GRAD else _drop(t, pdrop)
        if last_act:
            if residual is None and EPI_DROP_GRAD:
                return ops.linear(t, m["W2"], m["b2"], act_epi=act2, out_dtype=out_dtype, p_drop=pdrop)
            y2 = _drop(ops.linear(t, m["W2"], m["b2"], act_epi=act2), pdrop)
            return (y2 if residual is None else residual + y2).to(out_dtype)
        return ops.linear(t, m["W2"], m["b2"], act_epi=act2, residual=residual, out_dtype=out_dtype)
    t = ops.linear(y, m["W1"], m["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pdrop)
    return ops.linear(t, m["W2"], m["b2"], act_epi=act2, residual=residual, out_dtype=out_dtype,
                      p_drop=pdrop if last_act else 0.0)


def lin_drop(x, W, b, residual, pd):
    """residual + dropout(x W^T + b) with the dropout inside the GEMM's fp32 epilogue; under autograd the adjoint regenerates the
    counter-based mask (round 5: ops._Linear) -- a torch mask only for a residual that is not fp32."""
    if pd > 0.0 and ops.needs_grad(x, W, b, residual) and not (EPI_DROP_GRAD and residual.dtype == torch.float32 and W.shape[0] % 4 == 0):
        return residual + _drop(ops.linear(x, W, b), pd)
    return ops.linear(x, W, b, residual=residual, p_drop=pd)


def _coord_scalar(c, v, pd):
    """MLPwoBias coord_mlp on per-edge rows v [E,H] -> per-edge scalar partials [E, nt]: LN -> linear1 -> relu ->
    [dropout] -> bias-free linear2 as the row-dot of the GEMM epilogue."""
    ad = ops.act_dtype()
    fp = c.get("foldp")
    if FOLD_EDGE_LN and fp is not None and v.dtype == torch.bfloat16 and not ops.needs_grad(v, c["W1"], c["w3"]):
        # inference: the GEMM runs on the un-normalised messages, the epilogue applies rs * (acc - mu * cvec) + dvec
        mu, rs = K.row_stats(v, fp["eps"])
        return ops.linear_rowdot(v, fp["W1w"], fp["dvec"], c["w3"], act_epi=K.ACT_RELU, p_drop=pd, fold=(mu, rs, fp["cvec"]))
    yc = ln_rows(v, c["ln_w"], c["ln_b"], ad, c["k_pad"])
    if pd > 0.0 and ops.needs_grad(yc, c["W1"], c["w3"]) and not (EPI_DROP_GRAD and ROWDOT_DROP_GRAD):
        tc = ops.linear(yc, c["W1"], c["b1"], act_epi=K.ACT_RELU, p_drop=pd) if EPI_DROP_GRAD else \
            _drop(ops.linear(yc, c["W1"], c["b1"], act_epi=K.ACT_RELU), pd)
        return (tc * c["w3"]).sum(1, keepdim=True)
    return ops.linear_rowdot(yc, c["W1"], c["b1"], c["w3"], act_epi=K.ACT_RELU, p_drop=pd)


def gcl_layer(p, h, x, lay, g, clampv, pd=0.0):
    """MC_E_GCL.forward of FABind+ (egnn.py:104-118).  pd: dropout inside the three LN-MLPs (train / sampling mode)."""
    ad = ops.act_dtype()
    e = p["edge"]
    H = h.shape[1]
    d, rhohat = ops.edge_geom(x, g.row_ctx, g.col_ctx, g.rp_ctx, lay, g.ctx_by_col)
    grad = ops.needs_grad(h, x, e["W1"])
    if grad and FOLD_EDGE_LN_TRAIN and FOLD_EDGE_LN and "fold" in p and ad == torch.bfloat16 and e["k_pad"] <= 1536 and e["k_pad"] % 16 == 0:
        # training, bf16: the LayerNorm-folded first Linear under autograd (_EdgeLnFold); the node-level pieces are torch ops / ops.linear
        f = p["fold"]
        hf = h.float()
        mean = hf.mean(1)
        hcen = hf - mean[:, None]
        stat = torch.stack([mean, (hcen * hcen).sum(1)], 1).contiguous()                            # [N, 2]
        AB = ops.linear(hcen, f["W_ab"], None, out_dtype=torch.bfloat16)                             # [N, 2 K8]
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if pd > 0.0 else 0
        t = _EdgeLnFold.apply(AB, stat, rhohat, f["w_r"], f["c_r"], f["c_c"], f["dvec"], g, H, e["k_pad"], f["eps"], pd, seed)
        m = ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)            # [E, H] messages
    elif grad:     # the concatenation is materialised so that LayerNorm is a separate differentiable step
        cat = _EdgeConcat.apply(h, rhohat, g.row_ctx, g.col_ctx, g.rp_ctx, g.ctx_by_col, ad, e["k_pad"])
        y = ln_rows(cat[:, :2 * H + 1], e["ln_w"], e["ln_b"], ad, e["k_pad"])
        if MLP2_NODE and EPI_DROP_GRAD and y.dtype == torch.bfloat16 and e["W1"].dtype == torch.bfloat16:
            m = ops.mlp2_relu(y, e["W1"], e["b1"], e["W2"], e["b2"], True, None, ad, pd)
        elif EPI_DROP_GRAD:
            t = ops.linear(y, e["W1"], e["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)
            m = ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)
        else:
            t = _drop(ops.linear(y, e["W1"], e["b1"], act_epi=K.ACT_RELU, out_dtype=ad), pd)
            m = _drop(ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad), pd)
    elif FOLD_EDGE_LN and "fold" in p:
        # inference, bf16: W1 LN([h_r | h_c | rho]) from per-node projections and per-node statistics -- the
        # [E, 2H+1] x [2H+1, 2H+1] contraction becomes one [N, H] x [H, 2(2H+1)] contraction and a gather
        f = p["fold"]
        hf = h.float()
        mean = hf.mean(1)
        hcen = hf - mean[:, None]
        stat = torch.stack([mean, (hcen * hcen).sum(1)], 1).contiguous()                            # [N, 2]
        AB = ops.linear(hcen, f["W_ab"], None, out_dtype=torch.bfloat16)                             # [N, 2 K8]
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if pd > 0.0 else 0
        t = K.edge_lnfold(AB, e["k_pad"], H, g.row_ctx, g.col_ctx, rhohat, stat, f["eps"], f["w_r"], f["c_r"], f["c_c"], f["dvec"],
                          pd, seed)                                                                  # [E, K8]
        m = ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)            # [E, H] messages
    else:
        y = K.edge_ln_concat(h, g.row_ctx, g.col_ctx, rhohat, e["ln_w"], e["ln_b"], ad, e["k_pad"])  # [E, K8]
        t = ops.linear(y, e["W1"], e["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)            # [E, K8]
        m = ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)            # [E, H] messages
    s = _coord_scalar(p["coord"], m, pd)                                                            # [E, nt]
    x_new = ops.coord_update(x, d, s, g.rp_ctx, mean=True, clampv=clampv)
    agg = ops.segment_sum(m, g.rp_ctx, g.row_ctx, h.shape[0])
    h_new = ln_mlp(p["node"], torch.cat([h, agg], 1), True, residual=h, pdrop=pd)
    return h_new, x_new


FUSE_PAIR = os.environ.get("FABIND_PLUS_FUSE_PAIR", "1") == "1"
FOLD_EDGE_LN_TRAIN = os.environ.get("FABIND_PLUS_FOLD_EDGE_LN_TRAIN", "1") == "1"     # 0: under autograd concat -> LayerNorm -> GEMM (rounds 1-4; A/B)
FOLD_EDGE_LN = os.environ.get("FABIND_PLUS_FOLD_EDGE_LN", "1") == "1"


def cross_part(p, h, z, lay, pairs, pd=0.0, bias=None, p_next=None):
    """CrossAttentionModule.forward of FABind+ (cross_att.py:20-47) on the node layout and the ragged pair list ->
    (h, z_updated, bias_next).  bias: this layer's [pairs, 16] row-attention biases when the previous layer's fused pair
    update already produced them; bias_next: the same for layer p_next (None unless the fused kernel ran)."""
    ad = ops.act_dtype()
    scale = 1.0 / math.sqrt(32.0)
    # ---- both RowAttentionBlocks + node transitions, protein side in place in the node layout
    if bias is None:
        bias = ops.linear(z, p["W_pb"], p["b_pb"])                                                   # [pairs, 16] fp32
    hc = ops.take_rows(h, lay.c_index64)
    qg = ops.linear(h, p["Wqg_p"], p["bqg_p"])
    kv = ops.linear(hc, p["Wkv_p"])
    og = ops.cross_attn(qg, kv, bias, 0, 4, lay.desc_pf, lay.B, lay.max_P, lay.max_C, scale)
    hp = lin_drop(og, p["Wo_p"], p["bo_p"], h, pd)
    qg = ops.linear(hc, p["Wqg_c"], p["bqg_c"])
    kv = ops.linear(hp, p["Wkv_c"])
    og = ops.cross_attn(qg, kv, bias, 8, 12, lay.desc_cf, lay.B, lay.max_C, lay.max_P, scale)
    hc = lin_drop(og, p["Wo_c"], p["bo_c"], hc, pd)
    hp = ln_mlp(p["tr_p"], hp, True, residual=hp, pdrop=pd)
    hc = ln_mlp(p["tr_c"], hc, True, residual=hc, pdrop=pd)
    h = ops.put_rows(hp, hc, lay.c_index64, inplace=False)     # hp is saved by its ReLU epilogue under autograd
    # ---- pair update: z <- relu-MLP(LN(z + W_o32 (a32 * b32) + b_o32))   (no residual around pair_transition)
    ab32 = ops.linear(h, p["W_ab32"], p["b_ab32"])                                                   # [N, 128]
    bias_next = None
    pf = p.get("pair_fused")
    if FUSE_PAIR and pf is not None and z.dtype == torch.bfloat16 and not ops.needs_grad(z, ab32, p["W_o32"]):
        # one kernel per 64-pair tile: Hadamard + linear_out + residual + LN + both transition Linears (+ next biases)
        nf = p_next.get("pair_fused") if p_next is not None else None
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if pd > 0.0 else 0
        z, bias_next = K.pair_update_fused(ab32, 64, pairs.p_node, pairs.c_node, z.contiguous(), pf["Wop"], pf["bo"], pf["ln_w"],
                                           pf["ln_b"], pf["eps"], pf["W1p"], pf["b1"], pf["W2p"], pf["b2"],
                                           nf["Wbp"] if nf else None, nf["bb"] if nf else None, pd, seed)
    else:
        hd32 = pair_had(ab32, 64, pairs.p_node, pairs.c_node, ad, lay)                               # [pairs, 64]
        z1 = ops.linear(hd32, p["W_o32"], p["b_o32"], residual=z, out_dtype=ad)      # (a bf16 z is added as it is: FabindGemmArgs.r_dtype)
        z = ln_mlp(p["tr_z"], z1, True, out_dtype=ad, pdrop=pd)
        if Z_SINK:      # consumers: the inter-edge row gather, the next layer's pair-bias Linear and its residual: one gradient buffer
            z = ops.shared_grad(z)
    return h, z, bias_next


def att_layer(p, h, x, z, lay, g, pairs, batch_id, clampv, pd=0.0, bias=None, p_next=None):
    """MC_Att_L.forward of FABind+ (egnn.py:277-300) -> (h, x, alpha, z_updated, bias_next); z is the ragged pair list.
    pd: dropout of every nn.Dropout on the path (all constructed with args.dropout) in train / sampling mode."""
    H = h.shape[1]
    h, z, bias_next = cross_part(p, h, z, lay, pairs, pd, bias, p_next)
    # ---- inter-edge attention (att_model / node_model identical to v1; coord_mlp is an LN-MLP on v_e, handed to the
    #      fused kernel as a per-edge scalar)
    ridx = pairs.index_of(g.red_p, g.red_c, batch_id)
    zr = ops.take_rows(z, ridx)                                                                      # [n_red, H]
    bias_part = ops.linear(zr, p["W_bias"], p["b_bias"])                                             # [n_red, 8]: col 0
    qkv = ops.linear(h, p["Wqkv"], p["bqkv"])                                                        # [N, 3H]
    d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay, g.int_by_col)
    zero_cv = torch.zeros((h.shape[0], H), dtype=torch.float32, device=h.device)
    cf = p.get("coord_fold")
    if FOLD_EDGE_LN and cf is not None and not ops.needs_grad(qkv, x, p["w_rv"]):
        V = qkv[:, 2 * H:].float()
        Vc = V - V.mean(1, keepdim=True)
        stat = torch.stack([(Vc * Vc).sum(1), Vc @ cf["wc"]], 1).contiguous()                        # [N, 2]
        Pn = ops.linear(Vc, cf["W1w"], None, out_dtype=torch.bfloat16)                               # [N, H]
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if pd > 0.0 else 0
        s_ext = K.inter_coord_fold(Pn, H, g.col_int, rhohat, stat, cf["q_w"], cf["eps"], cf["u"], cf["d"], cf["w3"], pd, seed)
    else:
        AB = torch.cat([zero_cv, qkv[:, 2 * H:].float()], 1)                                         # v_e = V[col] + rho w_rv
        v_e = ops.gcl_pre(AB, H, _InterView(g), rhohat, p["w_rv"])
        s_ext = _coord_scalar(p["coord"], v_e, pd).sum(1)                                            # [E_int]
    h_new, x_new, alpha = ops.inter_attn(qkv, zero_cv, H, h, x, d, rhohat, g, bias_part, p["w_rk"], p["w_rv"], p["zeroH"],
                                         p["zeroH"], clampv, s_ext=s_ext.contiguous())
    if pd > 0.0:
        h_new = ops.drop_mix(h, h_new, pd)                                                           # egnn.py:207 dropout(agg): h + drop(h_new - h)
    return h_new, x_new, alpha, z, bias_next


def egnn_forward(P, h, x, z0, lay, g, las, x_las, pairs, batch_id, scale, step, capture=None, pd=0.0):
    clampv = 10.0 / scale
    h = ops.linear(h, P["W_in"], P["b_in"], p_drop=pd)      # (dropout in the fp32 epilogue; under autograd the adjoint regenerates the mask)
    z, bias = (ops.shared_grad(z0) if Z_SINK else z0), None
    for i in range(P["L"]):
        h, x = gcl_layer(P["gcl"][i], h, x, lay, g, clampv, pd)
        if capture is not None:
            capture["gcl_%d.h" % i], capture["gcl_%d.x" % i] = h.detach().clone(), x.detach().clone()
        h, x, alpha, z, bias = att_layer(P["att"][i], h, x, z, lay, g, pairs, batch_id, clampv, pd, bias,
                                         P["att"][i + 1] if i + 1 < P["L"] else None)
        if capture is not None:
            capture["att_%d.h" % i], capture["att_%d.x" % i], capture["att_%d.alpha" % i] = \
                h.detach().clone(), x.detach().clone(), alpha.detach().clone()
        x = ops.las_step(x, x_las, las, lay, step, 15.0 / scale)
    h, x = gcl_layer(P["out_layer"], h, x, lay, g, clampv, pd)
    return ops.linear(_drop(h, pd), P["W_out"], P["b_out"]), x, z


DEBUG_CAPTURE = None


def mlp_module(m, x, last_act=False, out_dtype=torch.float32, pdrop=0.0):
    """Apply one of the reference's MLP modules (MLP / MLPwithLastAct / MLPwoBias / MLP4Confidence parameter containers)
    to rows x."""
    def build(mm):
        pk = EagerPack(mm.linear1.weight.device)
        return pk.resolve(_mlp(mm, pk))
    pack = cached_pack(m, build)
    return ln_mlp(pack, x.contiguous(), last_act, out_dtype=out_dtype, pdrop=pdrop)


def stack_forward(model, X, Hin, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index, coord_LAS,
                  n_iter, pair="dense"):
    """EfficientMCAttModel.forward of FABind+ (refine_coord): mutates X in place, returns (X, H, pair_embed).
    pair = "dense": the reference's [B, Pmax, Cmax, H] tensor; "ragged": (z [pairs, H], PairList, Layout) without the
    padded copy (what the model's own heads consume); "none": None."""
    if not X.is_cuda:
        raise RuntimeError("fabind_amd: the FABind+ stack runs on a HIP device only (no CPU fallback); got " + str(X.device))
    args = model.args
    scale = float(args.coordinate_scale)
    P = prepare_stack_params(model)
    H = P["H"]
    lay = Layout.of(batch_id, segment_id)
    pairs = PairList(lay, X.device)
    bond_row = compound_edge_index[0].to(torch.int32).contiguous()
    bond_col = compound_edge_index[1].to(torch.int32).contiguous()
    bond_off = lay.ranges(bond_row)
    las = (LAS_edge_index[0].to(torch.int32).contiguous(), LAS_edge_index[1].to(torch.int32).contiguous())
    las = las + (lay.ranges(las[0]),)
    x = X.reshape(-1, 3).float().contiguous()
    x_las = coord_LAS.reshape(-1, 3).float().contiguous()
    mask_u8 = mask.to(torch.uint8).contiguous()
    Hin = Hin.float().contiguous()
    ad = ops.act_dtype()
    a0b0 = ops.linear(Hin, P["W_ab0"], P["b_ab0"])                                                   # [N, 2H]
    hd0 = pair_had(a0b0, H, pairs.p_node, pairs.c_node, ad, lay)
    z0 = ops.linear(hd0, P["W_o0"], P["b_o0"], out_dtype=ad)                                         # [pairs, H]
    cut_intra, cut_inter = float(model.extract_edges.intra_cutoff), float(model.extract_edges.inter_cutoff)
    Hout = z = None
    pd = float(args.dropout) if model.training else 0.0     # train mode at inference = FABind+ dropout sampling
    for r in range(n_iter):
        last = r == n_iter - 1
        g = Graph(lay, x.detach(), bond_row, bond_col, bond_off, cut_intra, cut_inter)
        with torch.set_grad_enabled(last and torch.is_grad_enabled()):      # att_model.py:199-221: only the last pass has grad
            Hout, xz, z = egnn_forward(P, Hin, x, z0, lay, g, las, x_las, pairs, batch_id, scale,
                                       float(args.geometry_reg_step_size), DEBUG_CAPTURE if last else None, pd)
            x = ops.select_rows(x, xz, mask_u8)
    model.last_graph = g
    with torch.no_grad():
        X.copy_(x.reshape(X.shape).to(X.dtype))
    Xout = x.reshape(X.shape)
    Xout = Xout if Xout.requires_grad else X
    if pair == "dense":
        return Xout, Hout, pairs.dense(z, lay)
    return Xout, Hout, ((z, pairs, lay) if pair == "ragged" else None)
