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

Round 1: inference path (no autograd), the kernels are the generic ones (GEMM, row LayerNorm, cross attention, inter-edge
attention, CSR reductions); the pair MLP is three GEMM launches per layer, not yet one fused kernel."""
import math

import numpy as np
import torch

from .. import kernels as K
from .. import ops
from ..config import get_precision
from ..engine import Graph, Layout, _cat


def _wd():
    return torch.float32 if get_precision() == "fp32" else torch.bfloat16


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
def _mlp(m, W, k_pad=None, n_pad=None):
    """LN-MLP pack: LayerNorm (w, b), linear1 padded to [n_pad, k_pad], linear2 padded to [., n_pad]."""
    w1, w2 = m.linear1.weight, m.linear2.weight
    k_pad = k_pad or w1.shape[1]
    n_pad = n_pad or w1.shape[0]
    has_ln = hasattr(m, "layernorm")
    return dict(ln_w=m.layernorm.weight.float().contiguous() if has_ln else None,
                ln_b=m.layernorm.bias.float().contiguous() if has_ln else None,
                C=w1.shape[1], k_pad=k_pad,
                W1=W(_padded(w1, n_pad, k_pad)), b1=_padvec(m.linear1.bias, n_pad).float().contiguous(),
                W2=W(_padded(w2, w2.shape[0], n_pad)),
                b2=None if m.linear2.bias is None else m.linear2.bias.float().contiguous())


def prepare_stack_params(model):
    wd = _wd()
    W = lambda t: t.to(wd).contiguous()
    gnn = model.gnn
    H, L = gnn.hidden_nf, gnn.n_layers
    il = model.inter_layer
    P = {"H": H, "L": L}
    P["W_ab0"] = W(_cat([il.linear_p.weight, il.linear_c.weight]))
    P["b_ab0"] = _cat([il.linear_p.bias, il.linear_c.bias]).contiguous()
    P["W_o0"], P["b_o0"] = W(il.linear_out.weight), il.linear_out.bias
    P["W_in"], P["b_in"] = W(gnn.linear_in.weight), gnn.linear_in.bias
    P["W_out"], P["b_out"] = W(gnn.linear_out.weight), gnn.linear_out.bias
    K8 = _pad8(2 * H + 1)

    def gcl(m):
        d = dict(edge=_mlp(m.edge_mlp, W, K8, K8), node=_mlp(m.node_mlp, W), coord=_mlp(m.coord_mlp, W))
        d["coord"]["w3"] = m.coord_mlp.linear2.weight[0].float().contiguous()
        return d

    P["gcl"] = [gcl(getattr(gnn, "gcl_%d" % i)) for i in range(L)]
    P["out_layer"] = gcl(gnn.out_layer)
    att = []
    for i in range(L):
        m = getattr(gnn, "att_%d" % i)
        cam = m.cross_attn_module
        d = {}
        for tag, blk in (("p", cam.p_attention_block), ("c", cam.c_attention_block)):
            a = blk.mha
            d["Wqg_" + tag] = W(_cat([a.linear_q.weight, a.linear_g.weight]))
            d["bqg_" + tag] = _cat([torch.zeros_like(a.linear_g.bias), a.linear_g.bias]).contiguous()
            d["Wkv_" + tag] = W(_cat([a.linear_k.weight, a.linear_v.weight]))
            d["Wo_" + tag], d["bo_" + tag] = W(a.linear_o.weight), a.linear_o.bias
        # pair biases of both blocks from z in one GEMM: columns [p lin 0-3 | p gate 4-7 | c lin 8-11 | c gate 12-15]
        pb, cb = cam.p_attention_block, cam.c_attention_block
        d["W_pb"] = W(_cat([pb.linear.weight, pb.linear_g.weight, cb.linear.weight, cb.linear_g.weight]))
        d["b_pb"] = _cat([pb.linear.bias, pb.linear_g.bias, cb.linear.bias, cb.linear_g.bias]).contiguous()
        d["tr_p"], d["tr_c"] = _mlp(cam.p_transition, W), _mlp(cam.c_transition, W)
        d["tr_z"] = _mlp(cam.pair_transition, W)
        i32 = cam.inter_layer
        # the 32-wide Hadamard block is zero-padded to 64 (a32 | 0 | b32 | 0) so that the contraction with W_o32 runs on the
        # LDS-DMA GEMM path (K % 64 == 0); the padded columns are exact zeros
        z32w, z32b = torch.zeros_like(i32.linear_p.weight), torch.zeros_like(i32.linear_p.bias)
        d["W_ab32"] = W(_cat([i32.linear_p.weight, z32w, i32.linear_c.weight, z32w]))  # [128, H]
        d["b_ab32"] = _cat([i32.linear_p.bias, z32b, i32.linear_c.bias, z32b]).contiguous()
        d["W_o32"] = W(_cat([i32.linear_out.weight, torch.zeros_like(i32.linear_out.weight)], 1))   # [H, 64]
        d["b_o32"] = i32.linear_out.bias
        Wkv, bkv = m.linear_kv.weight, m.linear_kv.bias
        d["Wqkv"] = W(_cat([m.linear_q.weight, Wkv[0::2, 1:], Wkv[1::2, 1:]]))
        d["bqkv"] = _cat([m.linear_q.bias, bkv[0::2], bkv[1::2]]).contiguous()
        d["w_rk"], d["w_rv"] = Wkv[0::2, 0].float().contiguous(), Wkv[1::2, 0].float().contiguous()
        wb = torch.zeros((8, H), dtype=m.attn_bias_proj.weight.dtype, device=m.attn_bias_proj.weight.device)
        wb[0] = m.attn_bias_proj.weight[0]
        d["W_bias"] = W(wb)
        d["b_bias"] = _padvec(m.attn_bias_proj.bias, 8).float().contiguous()
        d["coord"] = _mlp(m.coord_mlp, W)
        d["coord"]["w3"] = m.coord_mlp.linear2.weight[0].float().contiguous()
        d["zeroH"] = torch.zeros(H, dtype=torch.float32, device=Wkv.device)
        att.append(d)
    P["att"] = att
    return P


# ------------------------------------------------------------------------------------------------
# pair list: every (protein row, ligand row) of every complex, complex-major, protein-major inside a complex --
# the order the block descriptors of the attention kernels use (pair = pair_off[b] + p_local * C_b + c_local)
# ------------------------------------------------------------------------------------------------
class PairList:
    def __init__(self, lay, device):
        pp, cc, bb, ii, jj = [], [], [], [], []
        for b in range(lay.B):
            Pb, Cb, o = int(lay.P[b]), int(lay.C[b]), int(lay.off[b])
            i, j = np.meshgrid(np.arange(Pb), np.arange(Cb), indexing="ij")
            pp.append((o + Cb + i).ravel()); cc.append((o + j).ravel())
            bb.append(np.full(Pb * Cb, b)); ii.append(i.ravel()); jj.append(j.ravel())
        t = lambda a, dt: torch.from_numpy(np.concatenate(a)).to(dt).to(device)
        self.p_node, self.c_node = t(pp, torch.int32), t(cc, torch.int32)
        self.b, self.i, self.j = t(bb, torch.int64), t(ii, torch.int64), t(jj, torch.int64)
        self.n = int(self.p_node.shape[0])
        self.off = torch.from_numpy(lay.off[:-1].astype(np.int64)).to(device)
        self.C = torch.from_numpy(lay.C.astype(np.int64)).to(device)
        self.pair_off = torch.from_numpy(lay.pair_off_np[:-1].astype(np.int64)).to(device)

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
# layers
# ------------------------------------------------------------------------------------------------
def _drop(t, pr):
    """nn.Dropout of the reference modules (train mode = FABind+ sampling inference): element-wise mask + scale between
    kernels, torch's generator."""
    return torch.nn.functional.dropout(t, pr, True) if pr > 0.0 else t


def ln_mlp(m, x, last_act, residual=None, out_dtype=torch.float32, pdrop=0.0):
    """LN -> linear1 -> relu -> [dropout] -> linear2 (-> relu -> [dropout]) (+ residual).  model_utils.py:10-74; the
    dropouts run inside the GEMM epilogues (before the residual)."""
    ad = ops.act_dtype()
    y = K.layernorm_rows(x, m["ln_w"], m["ln_b"], ad, m["k_pad"]) if m["ln_w"] is not None else x
    t = ops.linear(y, m["W1"], m["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pdrop)
    return ops.linear(t, m["W2"], m["b2"], act_epi=K.ACT_RELU if last_act else K.ACT_NONE, residual=residual,
                      out_dtype=out_dtype, p_drop=pdrop if last_act else 0.0)


def gcl_layer(p, h, x, lay, g, clampv, pd=0.0):
    """MC_E_GCL.forward of FABind+ (egnn.py:104-118).  pd: dropout inside the three LN-MLPs (train / sampling mode)."""
    ad = ops.act_dtype()
    e = p["edge"]
    d, rhohat = ops.edge_geom(x, g.row_ctx, g.col_ctx, g.rp_ctx, lay)
    y = K.edge_ln_concat(h, g.row_ctx, g.col_ctx, rhohat, e["ln_w"], e["ln_b"], ad, e["k_pad"])     # [E, K8]
    t = ops.linear(y, e["W1"], e["b1"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)               # [E, K8]
    m = ops.linear(t, e["W2"], e["b2"], act_epi=K.ACT_RELU, out_dtype=ad, p_drop=pd)               # [E, H] messages
    c = p["coord"]
    yc = K.layernorm_rows(m, c["ln_w"], c["ln_b"], ad, c["k_pad"])
    # coord_mlp: dropout sits between relu(linear1) and the bias-free linear2 = the row-dot of the epilogue
    s = ops.linear_rowdot(yc, c["W1"], c["b1"], c["w3"], act_epi=K.ACT_RELU, p_drop=pd)             # [E, nt]
    x_new = ops.coord_update(x, d, s, g.rp_ctx, mean=True, clampv=clampv)
    agg = K.segment_sum(m, g.rp_ctx, h.shape[0])
    h_new = ln_mlp(p["node"], torch.cat([h, agg], 1), True, residual=h, pdrop=pd)
    return h_new, x_new


def att_layer(p, h, x, z, lay, g, pairs, batch_id, clampv, pd=0.0):
    """MC_Att_L.forward of FABind+ (egnn.py:277-300) -> (h, x, alpha, z_updated); z is the ragged pair list.
    pd: dropout of every nn.Dropout on the path (all constructed with args.dropout) in train / sampling mode."""
    H = h.shape[1]
    ad = ops.act_dtype()
    scale = 1.0 / math.sqrt(32.0)
    # ---- CrossAttentionModule (cross_att.py:20-47), protein side in place in the node layout
    bias = ops.linear(z, p["W_pb"], p["b_pb"])                                                       # [pairs, 16] fp32
    hc = ops.take_rows(h, lay.c_index64)
    qg = ops.linear(h, p["Wqg_p"], p["bqg_p"])
    kv = ops.linear(hc, p["Wkv_p"])
    og = ops.cross_attn(qg, kv, bias, 0, 4, lay.desc_pf, lay.B, lay.max_P, lay.max_C, scale)
    hp = ops.linear(og, p["Wo_p"], p["bo_p"], residual=h, p_drop=pd)
    qg = ops.linear(hc, p["Wqg_c"], p["bqg_c"])
    kv = ops.linear(hp, p["Wkv_c"])
    og = ops.cross_attn(qg, kv, bias, 8, 12, lay.desc_cf, lay.B, lay.max_C, lay.max_P, scale)
    hc = ops.linear(og, p["Wo_c"], p["bo_c"], residual=hc, p_drop=pd)
    hp = ln_mlp(p["tr_p"], hp, True, residual=hp, pdrop=pd)
    hc = ln_mlp(p["tr_c"], hc, True, residual=hc, pdrop=pd)
    h = ops.put_rows(hp, hc, lay.c_index64)
    # ---- pair update: z <- relu-MLP(LN(z + W_o32 (a32 * b32) + b_o32))   (no residual around pair_transition)
    ab32 = ops.linear(h, p["W_ab32"], p["b_ab32"])                                                   # [N, 128]
    hd32 = _pair_hadamard_call(ab32, 64, pairs.p_node, pairs.c_node, ad)                             # [pairs, 64]
    z1 = ops.linear(hd32, p["W_o32"], p["b_o32"], residual=z.float() if z.dtype != torch.float32 else z, out_dtype=ad)
    z = ln_mlp(p["tr_z"], z1, True, out_dtype=ad, pdrop=pd)
    # ---- inter-edge attention (att_model / node_model identical to v1; coord_mlp is an LN-MLP on v_e)
    zr = ops.take_rows(z, pairs.index_of(g.red_p, g.red_c, batch_id))                                # [n_red, H]
    bias_part = ops.linear(zr, p["W_bias"], p["b_bias"])                                             # [n_red, 8]: col 0
    qkv = ops.linear(h, p["Wqkv"], p["bqkv"])                                                        # [N, 3H]
    d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay)
    zero_cv = torch.zeros((h.shape[0], H), dtype=torch.float32, device=h.device)
    h_new, _, alpha = ops.inter_attn(qkv, zero_cv, H, h, x, d, rhohat, g, bias_part, p["w_rk"], p["w_rv"], p["zeroH"],
                                     p["zeroH"], clampv)
    if pd > 0.0:
        h_new = h + _drop(h_new - h, pd)                                                             # egnn.py:207 dropout(agg)
    AB = torch.cat([zero_cv, qkv[:, 2 * H:].float()], 1)                                             # v_e = V[col] + rho w_rv
    v_e = K.gcl_pre(AB, H, g.row_int, g.col_int, rhohat, p["w_rv"], ad)
    c = p["coord"]
    yc = K.layernorm_rows(v_e, c["ln_w"], c["ln_b"], ad, c["k_pad"])
    s = ops.linear_rowdot(yc, c["W1"], c["b1"], c["w3"], act_epi=K.ACT_RELU, p_drop=pd)
    x_new, _ = K.coord_update(x, d, s, g.rp_int, False, clampv, weight=alpha)
    return h_new, x_new, alpha, z


def egnn_forward(P, h, x, z0, lay, g, las, x_las, pairs, batch_id, scale, step, capture=None, pd=0.0):
    clampv = 10.0 / scale
    h = ops.linear(h, P["W_in"], P["b_in"], p_drop=pd)
    z = z0
    for i in range(P["L"]):
        h, x = gcl_layer(P["gcl"][i], h, x, lay, g, clampv, pd)
        if capture is not None:
            capture["gcl_%d.h" % i], capture["gcl_%d.x" % i] = h.clone(), x.clone()
        h, x, alpha, z = att_layer(P["att"][i], h, x, z, lay, g, pairs, batch_id, clampv, pd)
        if capture is not None:
            capture["att_%d.h" % i], capture["att_%d.x" % i], capture["att_%d.alpha" % i] = h.clone(), x.clone(), alpha.clone()
        x = ops.las_step(x, x_las, las, lay, step, 15.0 / scale)
    h, x = gcl_layer(P["out_layer"], h, x, lay, g, clampv, pd)
    return ops.linear(_drop(h, pd), P["W_out"], P["b_out"]), x, z


DEBUG_CAPTURE = None


def mlp_module(m, x, last_act=False, out_dtype=torch.float32, pdrop=0.0):
    """Apply one of the reference's MLP modules (MLP / MLPwithLastAct / MLPwoBias / MLP4Confidence parameter containers)
    to rows x."""
    wd = _wd()
    return ln_mlp(_mlp(m, lambda t: t.to(wd).contiguous()), x.contiguous(), last_act, out_dtype=out_dtype, pdrop=pdrop)


@torch.no_grad()
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
    lay = Layout(batch_id, segment_id)
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
    hd0 = _pair_hadamard_call(a0b0, H, pairs.p_node, pairs.c_node, ad)
    z0 = ops.linear(hd0, P["W_o0"], P["b_o0"], out_dtype=ad)                                         # [pairs, H]
    cut_intra, cut_inter = float(model.extract_edges.intra_cutoff), float(model.extract_edges.inter_cutoff)
    Hout = z = None
    pd = float(args.dropout) if model.training else 0.0     # train mode at inference = FABind+ dropout sampling
    for r in range(n_iter):
        g = Graph(lay, x, bond_row, bond_col, bond_off, cut_intra, cut_inter)
        Hout, xz, z = egnn_forward(P, Hin, x, z0, lay, g, las, x_las, pairs, batch_id, scale,
                                   float(args.geometry_reg_step_size), DEBUG_CAPTURE if r == n_iter - 1 else None, pd)
        x = ops.select_rows(x, xz, mask_u8)
    model.last_graph = g
    X.copy_(x.reshape(X.shape).to(X.dtype))
    if pair == "dense":
        return X, Hout, pairs.dense(z, lay)
    return X, Hout, ((z, pairs, lay) if pair == "ragged" else None)
