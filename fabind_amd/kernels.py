"""Thin Python wrappers over the C ABI (include/fabind_hip.h): allocate outputs with torch, pass raw
device pointers + the current HIP stream.  No math happens here."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_STORED_DERIV, GemmArgs, check, dt_code, ptr, stream

__all__ = ["gemm", "transpose_act", "colsum", "edges_count", "edges_fill", "exclusive_scan", "inter_meta",
           "edge_geom", "gcl_pre", "segment_sum", "coord_update", "cross_attn_fwd", "pair_bmat", "pair_hadamard",
           "inter_attn_fwd", "las_step", "select_rows", "ACT_NONE", "ACT_SILU", "ACT_RELU", "ACT_SIGMOID", "ACT_STORED_DERIV"]

GEMM_BN = 128
GEMM_DEFAULT_CONFIG = 13   # fabind_gemm_set_config default: 256x128x32 tile, 3 stages, 4-waves-per-SIMD bound (csrc/gemm.hip)
PROFILE = None  # dict: label -> list of (start_event, end_event, flops) around every MFMA-kernel launch (bench.py)
PROFILE_ONLY = None


PROFILE_BYTES = {}        # label -> DESIGN HBM bytes of the launches timed under that label (the tensors this design's launch reads once + writes once)
PROFILE_ALGO = {}         # label -> ALGORITHMIC bytes in the sense of SURVEY 8(d): the compulsory traffic of the reference operation the launch replaces
#                           (fused edge kernels: per-node rows + indices only -- the [E, H] tiles this design materialises are NOT in it); a launch that
#                           names no figure of its own is priced at its design bytes


def _profiled(label, flops, fn, nbytes=0.0, algo=None):
    # PROFILE_ONLY: name prefix of the one kernel family to time (bench.py: the dominant family, found in a fully timed warm-up step --
    # two events around each of the ~300 labelled launches of a step cost the host 1.7-2.6 ms of a 100 ms step, 2-3 ms of a 26 ms one)
    if PROFILE is None or (PROFILE_ONLY is not None and not label.startswith(PROFILE_ONLY)):
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.setdefault(label, []).append((e0, e1, flops))
    PROFILE_BYTES[label] = PROFILE_BYTES.get(label, 0.0) + float(nbytes)
    PROFILE_ALGO[label] = PROFILE_ALGO.get(label, 0.0) + float(nbytes if algo is None else algo)
    return r


def _split3():
    from .config import get_precision
    return get_precision() == "bf16x3"


def _ld(t):
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "innermost dimension must be contiguous"
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def gemm(A, W, bias=None, A2=None, act_pro=ACT_NONE, act_epi=ACT_NONE, residual=None, r_index=None,
         out=None, out_dtype=torch.float32, want_out=True, dotvec=None, aux=None, dact=ACT_NONE, alpha=1.0,
         accumulate=False, groups=None, n_groups=0, max_m=0, max_n=0, M=None, N=None, ldc=None, k_splits=1,
         out2=None, groups_ext=False, p_drop=0.0, seed=0, out16=None, flops=None, force_x3=False):
    """C = epi(pro([A|A2]) @ W^T); see FabindGemmArgs.  Returns (C or None, dot_partials or None).  out16: bf16 tensor that
    receives a copy of an fp32 C (plain bias / residual epilogues only).

    `groups` (int32 [G,8] device tensor) selects the ragged-batched mode; then `out` must be given.
    flops: executed multiply-add flops of a ragged launch (bench.py's live roofline accounting; default 2 M N K, which for a
    grouped launch is the padded bounding box, not the work).
    force_x3: fp32 A and fp32 W contracted as split bf16 whatever the precision mode (config.set_split_sites: single sites of the bf16 mode)."""
    lib = _lib.load()
    a = GemmArgs()
    K1 = A.shape[1]
    K = K1 + (A2.shape[1] if A2 is not None else 0)
    M = A.shape[0] if M is None else M
    N = W.shape[0] if N is None else N
    assert groups_ext or W.shape[1] == K, "gemm: K mismatch %s vs %s" % (tuple(W.shape), K)
    if want_out and out is None:
        out = torch.empty((k_splits, M, N) if k_splits > 1 else (M, N), dtype=out_dtype, device=A.device)
        if k_splits > 1:
            ldc = N
    dot_out = None
    if dotvec is not None:
        nt = (N + GEMM_BN - 1) // GEMM_BN
        dot_out = torch.empty((A.shape[0], nt), dtype=torch.float32, device=A.device)
        a.dot_ld = nt
    a.A, a.A2, a.W, a.C = ptr(A), ptr(A2), ptr(W), ptr(out) if want_out else None
    a.bias, a.R, a.r_index = ptr(bias), ptr(residual), ptr(r_index)
    a.dotvec, a.dot_out, a.aux, a.groups = ptr(dotvec), ptr(dot_out), ptr(aux), ptr(groups)
    a.C2 = ptr(out2)
    a.c2_bf16 = 1 if (out2 is not None and want_out and out2.dtype == torch.bfloat16 and out.dtype == torch.float32) else 0
    if out16 is not None:
        a.C16, a.ldc16 = ptr(out16), _ld(out16)
    a.M, a.N, a.K, a.K1 = M, N, K, K1
    a.lda, a.lda2, a.ldw = _ld(A), (_ld(A2) if A2 is not None else 0), _ld(W)
    a.ldc = (ldc if ldc is not None else (_ld(out) if want_out else 0))
    a.ldr = _ld(residual) if residual is not None else 0
    a.r_dtype = dt_code(residual.dtype) if residual is not None else 0
    a.ldaux = _ld(aux) if aux is not None else 0
    a.a_dtype, a.w_dtype = dt_code(A.dtype), dt_code(W.dtype)
    a.c_dtype = dt_code(out.dtype) if want_out else 0
    a.aux_dtype = dt_code(aux.dtype) if aux is not None else 0
    if A2 is not None:
        assert A2.dtype == A.dtype
    a.act_pro, a.act_epi, a.dact_epi = act_pro, act_epi, dact
    a.accumulate = 1 if accumulate else 0
    a.n_groups, a.max_m, a.max_n = n_groups, max_m, max_n
    a.k_splits = k_splits
    a.groups_ext = 1 if groups_ext else 0
    a.alpha = alpha
    a.p_drop, a.drop_seed = float(p_drop), int(seed) & 0xFFFFFFFF
    # 'bf16x3' mode: fp32 x fp32 contractions run as split bf16 (three bf16 MFMAs per product term) instead of the exact fp32 MFMA
    x3 = A.dtype == torch.float32 and W.dtype == torch.float32 and (_split3() or force_x3)
    a.split3 = 1 if x3 else 0
    if PROFILE is None:                                    # (the label costs more host time than the launch: formatted only when events are taken)
        check(lib.fabind_gemm(ctypes.byref(a), stream()), "fabind_gemm")
        return (out if want_out else None), dot_out
    # (the split-bf16 launches are their own family in bench.py's roofline accounting: another kernel, gemm_x3_kernel, with three MFMAs per
    #  product term and fp32 operands)
    label = "%s <%s,%s%s> M=%d N=%d K=%d" % ("fabind_gemm_x3" if x3 else "fabind_gemm", str(A.dtype).replace("torch.", ""),
                                             str(W.dtype).replace("torch.", ""), ",x3" if x3 else "", M, N, K)
    if groups is not None:
        label += " (ragged, %d groups)" % n_groups
    # algorithmic HBM bytes: every operand read once, every result written once (ragged launches: not counted)
    nb = 0.0
    if groups is None:
        mn = float(M) * N
        nb = float(M) * K * A.element_size() + float(N) * K * W.element_size()
        nb += mn * out.element_size() * (2 if accumulate else 1) if want_out else 0.0
        for t_ in (out2, residual, aux):
            if t_ is not None:
                nb += mn * t_.element_size()
        if out16 is not None:
            nb += mn * 2
    _profiled(label, 2.0 * M * N * K if flops is None else float(flops),
              lambda: check(lib.fabind_gemm(ctypes.byref(a), stream()), "fabind_gemm"), nb)
    return (out if want_out else None), dot_out


def transpose_act(x, act=ACT_NONE, out_dtype=None):
    R, C = x.shape
    out = torch.empty((C, R), dtype=out_dtype or x.dtype, device=x.device)
    check(_lib.load().fabind_transpose_act(ptr(x), dt_code(x.dtype), _ld(x), ptr(out), dt_code(out.dtype), R, R, C, act,
                                           stream()), "fabind_transpose_act")
    return out


def colsum(x, out=None, accumulate=False):
    R, C = x.shape
    nchunk = max(1, min(1024, (R + 255) // 256))
    scratch = torch.empty((nchunk, C), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((C,), dtype=torch.float32, device=x.device)
        accumulate = False
    check(_lib.load().fabind_colsum(ptr(x), dt_code(x.dtype), _ld(x), ptr(out), R, C, 1 if accumulate else 0,
                                    ptr(scratch), nchunk, stream()), "fabind_colsum")
    return out


def edges_count(x, node_off, c_cnt, B, max_n, bond_row, bond_off, cut_intra, cut_inter):
    N = x.shape[0]
    deg_ctx = torch.empty(N, dtype=torch.int32, device=x.device)
    deg_int = torch.empty(N, dtype=torch.int32, device=x.device)
    check(_lib.load().fabind_edges_count(ptr(x), ptr(node_off), ptr(c_cnt), B, max_n, ptr(bond_row), ptr(bond_off),
                                         cut_intra, cut_inter, ptr(deg_ctx), ptr(deg_int), stream()), "fabind_edges_count")
    return deg_ctx, deg_int


def exclusive_scan(deg):
    out = torch.empty(deg.shape[0] + 1, dtype=torch.int32, device=deg.device)
    check(_lib.load().fabind_exclusive_scan(ptr(deg), ptr(out), deg.shape[0], stream()), "fabind_exclusive_scan")
    return out


def edges_fill(x, node_off, c_cnt, B, max_n, bond_row, bond_col, bond_off, cut_intra, cut_inter, rowptr_ctx, rowptr_int,
               E_ctx, E_int):
    dev = x.device
    col_ctx = torch.empty(max(E_ctx, 1), dtype=torch.int32, device=dev)
    row_ctx = torch.empty(max(E_ctx, 1), dtype=torch.int32, device=dev)
    col_int = torch.empty(max(E_int, 1), dtype=torch.int32, device=dev)
    row_int = torch.empty(max(E_int, 1), dtype=torch.int32, device=dev)
    check(_lib.load().fabind_edges_fill(ptr(x), ptr(node_off), ptr(c_cnt), B, max_n, ptr(bond_row), ptr(bond_col),
                                        ptr(bond_off), cut_intra, cut_inter, ptr(rowptr_ctx), ptr(rowptr_int),
                                        ptr(col_ctx), ptr(row_ctx), ptr(col_int), ptr(row_int), stream()),
          "fabind_edges_fill")
    return col_ctx[:E_ctx], row_ctx[:E_ctx], col_int[:E_int], row_int[:E_int]


def inter_meta(node_off, c_cnt, B, rowptr_int, col_int, row_int):
    dev = col_int.device
    E = col_int.shape[0]
    red_off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    red_idx = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    n_red = E // 2
    red_c = torch.empty(max(n_red, 1), dtype=torch.int32, device=dev)
    red_p = torch.empty(max(n_red, 1), dtype=torch.int32, device=dev)
    mirror = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    check(_lib.load().fabind_inter_meta(ptr(node_off), ptr(c_cnt), B, ptr(rowptr_int), ptr(col_int), ptr(row_int), E,
                                        ptr(red_off), ptr(red_idx), ptr(red_c), ptr(red_p), ptr(mirror), stream()),
          "fabind_inter_meta")
    return red_off, red_idx[:E], red_c[:n_red], red_p[:n_red], mirror[:E]


def edge_geom(x, row, col, rowptr, node_off, B):
    E, dev = row.shape[0], x.device
    d = torch.empty((max(E, 1), 3), dtype=torch.float32, device=dev)
    rho = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    rhohat = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    norm = torch.empty(B, dtype=torch.float32, device=dev)
    check(_lib.load().fabind_edge_geom(ptr(x), ptr(row), ptr(col), ptr(rowptr), ptr(node_off), B, ptr(d), ptr(rho),
                                       ptr(rhohat), ptr(norm), stream()), "fabind_edge_geom")
    return d[:E], rho[:E], rhohat[:E], norm


def gcl_pre(AB, H, row, col, rhohat, w_r, out_dtype, act=ACT_NONE, want_dact=False):
    E = row.shape[0]
    pre = torch.empty((E, H), dtype=out_dtype, device=AB.device)
    dact = torch.empty((E, H), dtype=out_dtype, device=AB.device) if want_dact else None
    check(_lib.load().fabind_gcl_pre(ptr(AB), dt_code(AB.dtype), _ld(AB), H, ptr(row), ptr(col), ptr(rhohat), ptr(w_r), ptr(pre),
                                     dt_code(out_dtype), E, act, ptr(dact), stream()), "fabind_gcl_pre")
    return (pre, dact) if want_dact else pre


def layernorm_rows(x, w, b, out_dtype, pad_to=None, eps=1e-5):
    """Row LayerNorm -> [R, pad_to] (columns beyond C zero): FABind+ LN-MLP front end."""
    R, C = x.shape
    pad_to = pad_to or C
    y = torch.empty((R, pad_to), dtype=out_dtype, device=x.device)
    check(_lib.load().fabind_layernorm_rows(ptr(x), dt_code(x.dtype), _ld(x), ptr(w), ptr(b), eps, R, C, ptr(y),
                                            dt_code(out_dtype), pad_to, pad_to, stream()), "fabind_layernorm_rows")
    return y


def edge_ln_concat(h, row, col, rhohat, w, b, out_dtype, pad_to, eps=1e-5):
    """LayerNorm([h[row] | h[col] | rhohat]) -> [E, pad_to] without materialising the concatenation."""
    E, H = row.shape[0], h.shape[1]
    y = torch.empty((E, pad_to), dtype=out_dtype, device=h.device)
    check(_lib.load().fabind_edge_ln_concat(ptr(h), _ld(h), H, ptr(row), ptr(col), ptr(rhohat), ptr(w), ptr(b), eps, E, ptr(y),
                                            dt_code(out_dtype), pad_to, pad_to, stream()), "fabind_edge_ln_concat")
    return y


def segment_sum(Z, rowptr, n_rows, act=ACT_NONE, eidx=None, out=None, out16=None):
    """out16 (bf16 [n_rows, >= H], optional): the result as a bf16 operand; given alone, no fp32 result is written."""
    H = Z.shape[1]
    if out is None and out16 is None:
        out = torch.empty((n_rows, H), dtype=torch.float32, device=Z.device)
    check(_lib.load().fabind_segment_sum(ptr(Z), dt_code(Z.dtype), _ld(Z), H, ptr(rowptr), ptr(eidx), n_rows,
                                         Z.shape[0] if eidx is None else eidx.numel(), act, ptr(out), _ld(out) if out is not None else 0,
                                         ptr(out16), _ld(out16) if out16 is not None else 0, stream()), "fabind_segment_sum")
    return out if out is not None else out16


def coord_update(x, d, s_part, rowptr, mean, clampv, weight=None, want_s=False):
    n_rows = x.shape[0]
    x_out = torch.empty_like(x)
    s_out = torch.empty(s_part.shape[0], dtype=torch.float32, device=x.device) if want_s else None
    check(_lib.load().fabind_coord_update(ptr(x), ptr(d), ptr(s_part), s_part.shape[1], ptr(weight), ptr(rowptr), n_rows,
                                          1 if mean else 0, clampv, ptr(x_out), ptr(s_out), stream()), "fabind_coord_update")
    return x_out, s_out


CROSS_ATTN_MFMA = None   # None: MFMA bf16 tiles in bf16 mode, fp32 VALU kernels in fp32 (parity) mode; True / False force one


def cross_attn_use_mfma():
    from .config import get_precision
    return (get_precision() == "bf16") if CROSS_ATTN_MFMA is None else bool(CROSS_ATTN_MFMA)


def cross_attn_fwd(q, k, v, gpre, bias, lin_col, gate_col, desc, B, max_nq, scale, out, want_lse=False, max_nk=0):
    """Gated cross attention with pair bias (include/fabind_hip.h: fabind_cross_attn_fwd / fabind_cross_attn_mfma_fwd)."""
    lse = torch.empty((q.shape[0], 4), dtype=torch.float32, device=q.device) if want_lse else None
    assert k.stride(0) == v.stride(0)
    # few queries / many keys (ligand-query block): split the keys so that the launch fills the chip
    ksplit = max(1, min(32, max_nk // 128)) if (max_nq <= 256 and max_nk >= 512) else 1
    part = torch.empty(q.shape[0] * ksplit * 4 * 34, dtype=torch.float32, device=q.device) if ksplit > 1 else None
    mfma = cross_attn_use_mfma()
    fn = _lib.load().fabind_cross_attn_mfma_fwd if mfma else _lib.load().fabind_cross_attn_fwd
    n_tiles = sum(1 for _ in range(0, max_nq, 16))
    # executed matrix-core work: per 16-query tile and 32-key chunk, 4 heads x (2 score + 2 value) MFMAs of 16x16x32
    flops = 2.0 * 16 * 16 * 32 * 16 * n_tiles * ((max_nk + 31) // 32) * B if mfma else 0.0
    _profiled("cross_attn_%s_fwd B=%d nq<=%d nk<=%d" % ("mfma" if mfma else "fp32", B, max_nq, max_nk), flops,
              lambda: check(fn(ptr(q), _ld(q), ptr(k), ptr(v), _ld(k), ptr(gpre), _ld(gpre), ptr(bias), _ld(bias), lin_col, gate_col,
                               ptr(desc), B, max_nq, scale, ptr(out), _ld(out), ptr(lse), ksplit, ptr(part), q.shape[0], stream()),
                            "fabind_cross_attn_fwd"))
    return out, lse


CROSS_ATTN_FUSED = os.environ.get("FABIND_ATTN_FUSED", "1") == "1"   # forward-only bf16 passes: pair bias recomputed inside the attention kernels
CROSS_ATTN_FUSED_MAX_C = 62


def bo_tiles(lay):
    """(toff [B+1], tile_b [n_tiles], n_tiles) of the packed ligand-side operand of the fused cross attention: complex b owns
    ceil(C_b / 2) 16-column tiles (cached on the layout)."""
    c = getattr(lay, "_bo_tiles", None)
    if c is None:
        import numpy as np
        T = (np.asarray(lay.C) + 1) // 2
        toff = np.concatenate([[0], np.cumsum(T)]).astype(np.int32)
        tile_b = np.repeat(np.arange(lay.B, dtype=np.int32), T)
        dev = lay.node_off.device
        from .param_pack import upload
        c = lay._bo_tiles = (upload(toff, dev, torch.int32), upload(tile_b, dev, torch.int32), int(toff[-1]))
    return c


def pair_bo_pack(b0, wcomp, H, lay):
    """Packed bf16 operands Bo of ALL layers / blocks in one launch (include/fabind_hip.h: fabind_pair_bo_pack).  b0 = fp32 [N, ld] view
    (a0b0[:, H:]), wcomp = fp32 [nblk, 8, H] (or [8, H]) -> list of nblk flat bf16 tensors."""
    toff, tile_b, n_tiles = bo_tiles(lay)
    wc = wcomp.reshape(-1, 8, wcomp.shape[-1]).float().contiguous()
    nblk = wc.shape[0]
    per = max(n_tiles, 1) * (H // 32) * 512
    out = torch.empty((nblk, per), dtype=torch.bfloat16, device=b0.device)
    check(_lib.load().fabind_pair_bo_pack(ptr(b0), _ld(b0), ptr(wc), H, ptr(lay.c_index), ptr(lay.desc_pf), ptr(toff), ptr(tile_b),
                                          n_tiles, ptr(out), nblk, per, stream()), "fabind_pair_bo_pack")
    return [out[k] for k in range(nblk)]


def cross_attn_fused_fwd(q, k, v, gpre, a0_16, bo, bconst8, lay, H, mode, scale, out, lse=None):
    """Gated cross attention with the pair bias recomputed in the kernel (include/fabind_hip.h: fabind_cross_attn_fused_fwd).
    lse (fp32 [query rows, 4], optional): receives the log-sum-exp per (query, head) for the fused backward."""
    toff, _, _ = bo_tiles(lay)
    nsplit = (lay.max_P + 63) // 64
    part = torch.empty(lay.sumC * nsplit * 4 * 34, dtype=torch.float32, device=q.device) if mode == 1 else None
    assert k.stride(0) == v.stride(0) and a0_16.dtype == torch.bfloat16
    n_wg = nsplit * lay.B
    # executed matrix-core work: bias contraction 64 rows x (8 tiles of the busiest wave x 4 waves) x H per work-group + the attention's own
    flops = 2.0 * n_wg * 64 * (((lay.max_C + 1) // 2 + 3) // 4 * 4 * 16) * H + 2.0 * 16 * 16 * 32 * 16 * n_wg * 4 * ((lay.max_C if mode == 0 else 64) + 31) // 32
    _profiled("cross_attn_fused_fwd mode=%d B=%d P<=%d C<=%d (bias contraction K=%d + attention)" % (mode, lay.B, lay.max_P, lay.max_C, H), flops,
              lambda: check(_lib.load().fabind_cross_attn_fused_fwd(ptr(q), _ld(q), ptr(k), ptr(v), _ld(k), ptr(gpre), _ld(gpre), ptr(a0_16),
                                                                    _ld(a0_16), ptr(bo), ptr(toff), ptr(bconst8), ptr(lay.desc_pf), lay.B,
                                                                    lay.max_P, lay.max_C, H, mode, scale, ptr(out), _ld(out), ptr(part),
                                                                    lay.sumC, ptr(lse), stream()), "fabind_cross_attn_fused_fwd"))
    return out


def bot_ksteps(lay):
    """(koff [B+1], kstep_b [n_ksteps], n_ksteps) of the packed operand of the fused attention backward's d a0 contraction: complex b
    owns ceil(C_b / 4) k-steps of 4 ligand-side nodes (cached on the layout)."""
    c = getattr(lay, "_bot_ksteps", None)
    if c is None:
        KS = (np.asarray(lay.C) + 3) // 4
        koff = np.concatenate([[0], np.cumsum(KS)]).astype(np.int32)
        kstep_b = np.repeat(np.arange(lay.B, dtype=np.int32), KS)
        dev = lay.node_off.device
        from .param_pack import upload
        c = lay._bot_ksteps = (upload(koff, dev, torch.int32), upload(kstep_b, dev, torch.int32), int(koff[-1]))
    return c


def pair_bot_pack(b0, wcomp, H, lay):
    """Packed bf16 operands Bo^T of ALL layers / blocks for the fused backward, one launch (include/fabind_hip.h: fabind_pair_bot_pack)."""
    koff, kstep_b, n_ks = bot_ksteps(lay)
    wc = wcomp.reshape(-1, 8, wcomp.shape[-1]).float().contiguous()
    nblk = wc.shape[0]
    per = max(n_ks, 1) * (H // 16) * 512
    out = torch.empty((nblk, per), dtype=torch.bfloat16, device=b0.device)
    check(_lib.load().fabind_pair_bot_pack(ptr(b0), _ld(b0), ptr(wc), H, ptr(lay.c_index), ptr(lay.desc_pf), ptr(koff), ptr(kstep_b), n_ks,
                                           ptr(out), nblk, per, stream()), "fabind_pair_bot_pack")
    return [out[k] for k in range(nblk)]


def cross_attn_fused_bwd(qg, kv, a0_16, bo, boT, bconst8, lay, H, mode, scale, out, lse, dout, dqg, dkv, da0b0, acat, kcol0, colpart, kp):
    """Fused backward of the cross attention (include/fabind_hip.h: fabind_cross_attn_fused_bwd).  dqg / dkv are written (covered rows),
    da0b0[:, :H] is accumulated, acat[:, kcol0 : kcol0 + C * 8] receives the bf16 gradient rows, colpart (a [tiles, >= 8] view: the
    first 8 columns of every row) the per-tile column sums of the per-pair gradients in slot order lin0, gate0, lin1, gate1, ..."""
    lib = _lib.load()
    toff, _, _ = bo_tiles(lay)
    koff, _, _ = bot_ksteps(lay)
    dev = qg.device
    nq = qg.shape[0]
    dO = torch.empty((nq, 128), dtype=torch.float32, device=dev)
    Dv = torch.empty((nq, 4), dtype=torch.float32, device=dev)
    part = torch.empty(max(1, int(lib.fabind_cross_attn_fused_bwd_scratch(lay.B, lay.max_P, lay.max_C, lay.sumC, mode))), dtype=torch.float32, device=dev)
    a = _lib.AttnFusedBwdArgs()
    for name, t in (("qg", qg), ("kv", kv), ("a0", a0_16), ("bo", bo), ("boT", boT), ("toff", toff), ("koff", koff), ("bconst", bconst8),
                    ("desc", lay.desc_pf), ("desc_p", lay.desc_p), ("out", out), ("lse", lse), ("dout", dout), ("dqg", dqg), ("dkv", dkv),
                    ("dO", dO), ("Dv", Dv), ("da0", da0b0), ("acat", acat), ("colpart", colpart), ("part", part)):
        setattr(a, name, ptr(t))
    a.ldq, a.ldkv, a.lda0, a.ldacat, a.kcol0, a.ldcolpart = _ld(qg), _ld(kv), _ld(a0_16), _ld(acat), int(kcol0), colpart.stride(0)
    a.ldda0 = _ld(da0b0) if da0b0 is not None else 0
    a.kp = int(kp)
    a.scale = float(scale)
    nsplit = (lay.max_P + 63) // 64
    n_wg = nsplit * lay.B
    # executed matrix-core work per work-group: bias contraction twice (both passes) + the d a0 contraction + the attention's own tiles
    kt = ((lay.max_C + 1) // 2 + 3) // 4 * 4 * 16
    flops = 2.0 * n_wg * 64 * kt * H * 2 + 2.0 * n_wg * 64 * ((lay.max_C + 3) // 4 * 32) * H
    _profiled("cross_attn_fused_bwd mode=%d B=%d P<=%d C<=%d (2 x bias contraction K=%d + d a0 contraction + attention adjoint)" % (mode, lay.B, lay.max_P, lay.max_C, H),
              flops, lambda: check(lib.fabind_cross_attn_fused_bwd(ctypes.byref(a), lay.B, lay.max_P, lay.max_C, H, mode, lay.sumC, stream()),
                                   "fabind_cross_attn_fused_bwd"))
    return colpart


def pair_bmat(b0, wcomp, c_node, out_dtype):
    NO, H = wcomp.shape
    n_c = c_node.shape[0]
    out = torch.empty((n_c * NO, H), dtype=out_dtype, device=b0.device)
    check(_lib.load().fabind_pair_bmat(ptr(b0), _ld(b0), ptr(wcomp), NO, H, ptr(c_node), n_c, ptr(out),
                                       dt_code(out_dtype), stream()), "fabind_pair_bmat")
    return out


def pair_hadamard(a0, b0, a1, b1, red_p, red_c, out_dtype):
    H, H2 = a0.shape[1], a1.shape[1]
    n_red = red_p.shape[0]
    hd = torch.empty((n_red, H + H2), dtype=out_dtype, device=a0.device)
    assert a0.stride(0) == b0.stride(0) and a1.stride(0) == b1.stride(0)
    check(_lib.load().fabind_pair_hadamard(ptr(a0), ptr(b0), _ld(a0), H, ptr(a1), ptr(b1), _ld(a1), H2, ptr(red_p),
                                           ptr(red_c), n_red, ptr(hd), dt_code(out_dtype), H + H2, stream()),
          "fabind_pair_hadamard")
    return hd


INTER_ATTN_ROWS = os.environ.get("FABIND_INTER_ATTN_ROWS", "1") == "1"    # 0: one wave per row whatever its degree (rounds 1-4; A/B)
INTER_ATTN_HEAVY = 8                                                      # = IA_HEAVY of csrc/inter_attn_rows.hip


def inter_attn_fwd(qkv, cv, H, h, x, d, rhohat, rowptr, col, red_idx, bias_part, w_rk, w_rv, wcr, w3, clampv, s_ext=None, h16=None, deal=None,
                   inplace=False):
    """h16 (bf16 [n_rows, H], optional): receives h_out as a bf16 operand.
    deal = (order int32 [n_rows], n_heavy, n_act) (engine.Graph.int_deal): the rows dealt by degree -- heavy rows on four waves
    (csrc/inter_attn_rows.hip); None: one wave per row.
    inplace (needs `deal`; no-grad passes whose caller owns h): h_out IS h and h16 the caller's existing bf16 copy of h -- only the rows
    WITH inter edges are rewritten (8 % of the rows of the headline batch; the pass-through copy of the other 92 % is 0.5 GB per call)."""
    n_rows, E = h.shape[0], col.shape[0]
    alpha = torch.empty(max(E, 1), dtype=torch.float32, device=h.device)
    cvs = torch.empty(max(E, 1), dtype=torch.float32, device=h.device)
    if deal is not None and INTER_ATTN_ROWS:
        inplace = bool(inplace) and h.is_contiguous()
        h_out, x_out = (h, x.clone()) if inplace else (torch.empty_like(h), torch.empty_like(x))
        check(_lib.load().fabind_inter_attn_fwd_rows(ptr(qkv), _ld(qkv), ptr(cv), _ld(cv), H, ptr(h), _ld(h), ptr(x), ptr(d),
                                                     ptr(rhohat), ptr(rowptr), ptr(col), ptr(red_idx), ptr(bias_part),
                                                     bias_part.shape[1], ptr(w_rk), ptr(w_rv), ptr(wcr), ptr(w3), clampv,
                                                     int(deal[2]) if inplace else n_rows,      # (in place: the rows without edges are not touched)
                                                     ptr(h_out), ptr(x_out), ptr(alpha), ptr(cvs), ptr(s_ext), ptr(h16), ptr(deal[0]),
                                                     int(deal[1]), int(deal[2]), stream()), "fabind_inter_attn_fwd_rows")
        return h_out, x_out, alpha[:E], cvs[:E]
    h_out, x_out = torch.empty_like(h), torch.empty_like(x)
    check(_lib.load().fabind_inter_attn_fwd(ptr(qkv), _ld(qkv), ptr(cv), _ld(cv), H, ptr(h), _ld(h), ptr(x), ptr(d),
                                            ptr(rhohat), ptr(rowptr), ptr(col), ptr(red_idx), ptr(bias_part),
                                            bias_part.shape[1], ptr(w_rk), ptr(w_rv), ptr(wcr), ptr(w3), clampv, n_rows,
                                            ptr(h_out), ptr(x_out), ptr(alpha), ptr(cvs), ptr(s_ext), ptr(h16), stream()),
          "fabind_inter_attn_fwd")
    return h_out, x_out, alpha[:E], cvs[:E]


def las_step(x, x0, las_i, las_j, las_off, node_off, c_cnt, B, max_n, step, clampv):
    x_out = torch.empty_like(x)
    check(_lib.load().fabind_las_step(ptr(x), ptr(x0), ptr(las_i), ptr(las_j), ptr(las_off), ptr(node_off), ptr(c_cnt), B,
                                      max_n, step, clampv, ptr(x_out), stream()), "fabind_las_step")
    return x_out


def select_rows(x, z, mask_u8):
    out = torch.empty_like(x)
    check(_lib.load().fabind_select_rows(ptr(x), ptr(z), ptr(mask_u8), x.shape[0], x.shape[1], ptr(out), stream()),
          "fabind_select_rows")
    return out


def pack_frag(W):
    """[N,K] weight -> bf16 MFMA-fragment order [K/32][N/16][4][16][8] (one contiguous 1 KiB block per wave load)."""
    N, Kd = W.shape
    return W.to(torch.bfloat16).view(N // 16, 16, Kd // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()


_PACKSEG = np.dtype([("src", np.uint64), ("dst", np.uint64), ("src_sr", np.int64), ("src_sc", np.int64), ("N", np.int32), ("K", np.int32),
                     ("src_dt", np.int32), ("pad_", np.int32)])          # = FabindPackSeg


def pack_frag_multi(weights, lo=None):
    """pack_frag of MANY [N, K] weights (any 2-D views: a transposed view is a stride swap) in one launch (csrc/gcl.hip) -> list of
    bf16 tensors in fragment order [K/32][N/16][4][16][8].  lo: optional list of flags, True = pack the LO plane bf16(w - bf16(w)) of an
    fp32 weight (with the plain pack of the same weight: the hi | lo operand of the split-precision kernels)."""
    from .param_pack import _upload
    n = len(weights)
    if n == 0:
        return []
    dev = weights[0].device
    sizes = [W.shape[0] * W.shape[1] for W in weights]
    buf = torch.empty(sum(sizes), dtype=torch.bfloat16, device=dev)
    tab = np.zeros(n, dtype=_PACKSEG)
    off = np.cumsum([0] + sizes[:-1])
    tab["src"] = [W.data_ptr() for W in weights]
    tab["dst"] = buf.data_ptr() + 2 * off
    tab["src_sr"], tab["src_sc"] = [W.stride(0) for W in weights], [W.stride(1) for W in weights]
    tab["N"], tab["K"] = [W.shape[0] for W in weights], [W.shape[1] for W in weights]
    tab["src_dt"] = [dt_code(W.dtype) for W in weights]
    if lo is not None:
        assert len(lo) == n and all(W.dtype == torch.float32 for W, f_ in zip(weights, lo) if f_)
        tab["src_dt"] = [2 if f_ else int(c_) for c_, f_ in zip(tab["src_dt"], lo)]
    tdev = _upload(tab, dev)
    check(_lib.load().fabind_pack_frag_multi(tdev.data_ptr(), tab.ctypes.data, n, stream()), "fabind_pack_frag_multi")
    outs = []
    for W, o, sz in zip(weights, off, sizes):
        N, Kd = W.shape
        outs.append(buf[int(o):int(o) + sz].view(Kd // 32, N // 16, 4, 16, 8))
    return outs


def node_chain_pack(W1, W2, kind):
    """Fragment packs of the H x H blocks of a chain's two weights (see node_chain_fwd): kind 0: W1 [H, 2H], W2 [H, H];
    kind 1: W1 [2H, H], W2 [H, 2H]."""
    H = W2.shape[0]
    if kind == 0:
        assert W1.shape == (H, 2 * H) and W2.shape == (H, H)
        return pack_frag(W1[:, :H].contiguous()), pack_frag(W1[:, H:].contiguous()), pack_frag(W2), None
    assert W1.shape == (2 * H, H) and W2.shape == (H, 2 * H)
    return pack_frag(W1[:H].contiguous()), pack_frag(W1[H:].contiguous()), pack_frag(W2[:, :H].contiguous()), pack_frag(W2[:, H:].contiguous())


def node_chain_fwd(X1, X2, packs, b1, b2, act, kind, residual=None, want16=False):
    """out = act([X1 | X2] W1^T + b1) W2^T + b2 (+ residual) with the hidden activation kept in LDS (csrc/node_chain.hip, forward only).
    -> (out fp32 [M, H], bf16 copy or None)."""
    M, H = X1.shape
    assert X1.dtype == torch.bfloat16 and (X2 is None or X2.dtype == torch.bfloat16)
    out = torch.empty((M, H), dtype=torch.float32, device=X1.device)
    out16 = torch.empty((M, H), dtype=torch.bfloat16, device=X1.device) if want16 else None
    W1a, W1b, W2a, W2b = packs
    hidden = H if kind == 0 else 2 * H
    _profiled("fabind_node_chain_fwd kind=%d M=%d H=%d" % (kind, M, H), 2.0 * M * (2 * H * hidden),
              lambda: check(_lib.load().fabind_node_chain_fwd(
                  ptr(X1), _ld(X1), ptr(X2), _ld(X2) if X2 is not None else 0, ptr(W1a), ptr(W1b), ptr(b1), ptr(W2a), ptr(W2b), ptr(b2), act,
                  kind, ptr(residual), _ld(residual) if residual is not None else 0, ptr(out), _ld(out), ptr(out16),
                  H if want16 else 0, M, H, stream()), "fabind_node_chain_fwd"))
    return out, out16


def node_chain_x3_blocks(W1, W2, kind):
    """The H x H blocks (strided VIEWS, None for an absent one) of a chain's two fp32 master weights in pack order W1a, W1b, W2a, W2b."""
    H = W2.shape[0]
    if kind == 0:
        assert W1.shape == (H, 2 * H) and W2.shape == (H, H)
        return [W1[:, :H], W1[:, H:], W2, None]
    assert W1.shape == (2 * H, H) and W2.shape == (H, 2 * H)
    return [W1[:H], W1[H:], W2[:, :H], W2[:, H:]]


def node_chain_x3_pack_many(chains):
    """hi | lo fragment packs of MANY chains in ONE launch: chains = [(W1_32, W2_32, kind), ...] -> list of 8-tuples (node_chain_x3_fwd's
    `packs`).  (Under autograd the packs are functions of the current weights: built once per step for the whole model.)"""
    ws, lo, slots = [], [], []
    for ci, (W1, W2, kind) in enumerate(chains):
        for bi, b_ in enumerate(node_chain_x3_blocks(W1, W2, kind)):
            if b_ is None:
                continue
            ws += [b_, b_]
            lo += [False, True]
            slots.append((ci, bi))
    packs = pack_frag_multi(ws, lo)
    out = [[None] * 8 for _ in chains]
    for k, (ci, bi) in enumerate(slots):
        out[ci][2 * bi], out[ci][2 * bi + 1] = packs[2 * k], packs[2 * k + 1]
    return [tuple(o) for o in out]


def node_chain_x3_pack(W1, W2, kind):
    """hi | lo fragment packs of the H x H blocks of a chain's two fp32 master weights (node_chain_x3_fwd): 8 packs in the order
    (W1a hi, lo, W1b hi, lo, W2a hi, lo, W2b hi, lo); kind as in node_chain_pack."""
    H = W2.shape[0]
    if kind == 0:
        assert W1.shape == (H, 2 * H) and W2.shape == (H, H)
        blocks = (W1[:, :H], W1[:, H:], W2, None)
    else:
        assert W1.shape == (2 * H, H) and W2.shape == (H, 2 * H)
        blocks = (W1[:H], W1[H:], W2[:, :H], W2[:, H:])
    out = []
    for b_ in blocks:
        out.extend(pack_frag_split(b_.contiguous()) if b_ is not None else (None, None))
    return tuple(out)


def node_chain_x3_fwd(X1, X2, packs, b1, b2, act, kind, residual=None, want16=False, save=False):
    """out = act([X1 | X2] W1^T + b1) W2^T + b2 (+ residual) in split precision with the hidden layer on chip (csrc/node_chain.hip):
    fp32 X rows, packs from node_chain_x3_pack.  -> (out fp32 [M, H], bf16 copy or None); save (training forward): + (t16, d16): bf16 of
    the activated hidden layer [M, hidden] and (SiLU) of its derivative, what the bf16 backward of ops._MLP2 reads."""
    M, H = X1.shape
    assert X1.dtype == torch.float32 and (X2 is None or X2.dtype == torch.float32)
    out = torch.empty((M, H), dtype=torch.float32, device=X1.device)
    out16 = torch.empty((M, H), dtype=torch.bfloat16, device=X1.device) if want16 else None
    hidden = H if kind == 0 else 2 * H
    t16 = torch.empty((M, hidden), dtype=torch.bfloat16, device=X1.device) if save else None
    d16 = torch.empty((M, hidden), dtype=torch.bfloat16, device=X1.device) if (save and act == ACT_SILU) else None
    _profiled("fabind_node_chain_x3_fwd kind=%d M=%d H=%d" % (kind, M, H), 2.0 * M * (2 * H * hidden),
              lambda: check(_lib.load().fabind_node_chain_x3_fwd(
                  ptr(X1), _ld(X1), ptr(X2), _ld(X2) if X2 is not None else 0, *[ptr(t_) for t_ in packs[:4]], ptr(b1),
                  *[ptr(t_) for t_ in packs[4:]], ptr(b2), act, kind, ptr(residual), _ld(residual) if residual is not None else 0,
                  ptr(out), _ld(out), ptr(out16), H if want16 else 0, ptr(t16), hidden if save else 0, ptr(d16), hidden if d16 is not None else 0,
                  M, H, stream()), "fabind_node_chain_x3_fwd"))
    return (out, out16, t16, d16) if save else (out, out16)


def pack_frag_split(W):
    """fp32 [N,K] weight -> (hi, lo) bf16 fragment packs of the split-bf16 kernels: hi = bf16(W), lo = bf16(W - hi)."""
    hi = W.to(torch.bfloat16)
    lo = (W.float() - hi.float()).to(torch.bfloat16)
    return pack_frag(hi), pack_frag(lo)


def gcl_edge_fused_x3(AB, H, row, col, rhohat, w_r, W2, b2, Wc, bc, w3, n_rows, p_drop=0.0, seed=0, rowptr=None, save=False):
    """The fused forward edge pipeline in split-bf16 arithmetic (csrc/fused_edge_fwd3.hip): AB fp32 [N, 2H], fp32 weights (split into
    hi | lo fragment packs here) -> (agg [N,H] fp32, s [E,1] fp32).  rowptr: see gcl_edge_fused.
    save (training forward): also -> (M [E,H] bf16, d2f, z3f) as a third result: what gcl_edge_fused_bwd(saved=...) reads (see gcl_edge_fused)."""
    E = row.shape[0]
    if save and E > 0:
        nt = (E + 63) // 64
        Msave = torch.empty((E, H), dtype=torch.bfloat16, device=AB.device)
        d2f = torch.empty((nt * 64, H), dtype=torch.bfloat16, device=AB.device)
        z3f = torch.empty((nt * 64, H), dtype=torch.bfloat16, device=AB.device)
    else:
        save = False
    assert AB.dtype == torch.float32 and AB.stride(1) == 1
    alloc = torch.zeros if (rowptr is None or E == 0) else torch.empty
    agg = alloc((n_rows, H), dtype=torch.float32, device=AB.device)
    if alloc is torch.empty:
        zero_empty_rows(rowptr, agg, H)
    s = torch.empty((max(E, 1), 1), dtype=torch.float32, device=AB.device)
    bnd = torch.empty(((E + 63) // 64 * 2 + 2, H), dtype=torch.float32, device=AB.device)
    W2h, W2l = pack_frag_split(W2)
    Wch, Wcl = pack_frag_split(Wc)
    if save:
        _profiled("gcl_edge_fused_x3_kernel<%d> E=%d saving (gather + 2 chained H x H split-bf16 contractions + segment-sum per edge; M, silu'(pre2), pre3 kept)" % (H, E),
                  4.0 * E * H * H,
                  lambda: check(_lib.load().fabind_gcl_edge_fused_x3_train(ptr(AB), _ld(AB), AB.shape[0], H, ptr(row), ptr(col), ptr(rhohat), ptr(w_r),
                                                                           ptr(W2h), ptr(W2l), ptr(b2), ptr(Wch), ptr(Wcl), ptr(bc), ptr(w3), E,
                                                                           ptr(agg), ptr(s), float(p_drop), int(seed) & 0xFFFFFFFF, ptr(bnd),
                                                                           ptr(Msave), ptr(d2f), ptr(z3f), stream()), "fabind_gcl_edge_fused_x3_train"),
                  E * (6.0 * H + 16) + n_rows * (8.0 * H + 4.0 * H),
                  algo=2.0 * n_rows * H * 4 + E * 8.0 + n_rows * 24.0)        # 2 N H s + E 8 + N 24 (s = 4: fp32 rows)
        return agg, s[:E], (Msave, d2f, z3f)
    _profiled("gcl_edge_fused_x3_kernel<%d> E=%d (gather + 2 chained H x H split-bf16 contractions + segment-sum per edge)" % (H, E),
              4.0 * E * H * H,
              lambda: check(_lib.load().fabind_gcl_edge_fused_x3(ptr(AB), _ld(AB), AB.shape[0], H, ptr(row), ptr(col), ptr(rhohat), ptr(w_r),
                                                                 ptr(W2h), ptr(W2l), ptr(b2), ptr(Wch), ptr(Wcl), ptr(bc), ptr(w3), E,
                                                                 ptr(agg), ptr(s), float(p_drop), int(seed) & 0xFFFFFFFF, ptr(bnd),
                                                                 stream()), "fabind_gcl_edge_fused_x3"),
              E * 16.0 + n_rows * (8.0 * H + 4.0 * H), algo=2.0 * n_rows * H * 4 + E * 8.0 + n_rows * 24.0)
    return agg, s[:E]


def zero_empty_rows(rowptr, out, C, out2=None):
    """out[r, :C] (and out2[r, :C]) = 0 for the rows of the CSR `rowptr` that have no entries (csrc/gcl.hip)."""
    n_rows = rowptr.shape[0] - 1
    check(_lib.load().fabind_zero_empty_rows(ptr(rowptr), n_rows, ptr(out), dt_code(out.dtype), _ld(out), C, ptr(out2),
                                             dt_code(out2.dtype) if out2 is not None else 0, _ld(out2) if out2 is not None else 0,
                                             stream()), "fabind_zero_empty_rows")


def gcl_edge_fused(AB16, H, row, col, rhohat, w_r, W2p, b2, Wcp, bc, w3, n_rows, p_drop=0.0, seed=0, want16=False, rowptr=None, save=False):
    """want16: also return the aggregated messages as bf16 (the node MLP's operand), written by the same kernels.
    rowptr (CSR of `row`, optional): only the rows without edges are zeroed instead of the whole output.
    save (training forward): the kernel also leaves (M [E,H] bf16, d2f, z3f) -- the messages and, in its accumulator-fragment order,
    silu'(pre2) and pre3 -- appended to the result as one tuple: what gcl_edge_fused_bwd(saved=...) needs to run without the forward
    recompute (csrc/fused_edge_bwd4.hip)."""
    E = row.shape[0]
    if save and E > 0:
        nt = (E + 63) // 64
        Msave = torch.empty((E, H), dtype=torch.bfloat16, device=AB16.device)
        d2f = torch.empty((nt * 64, H), dtype=torch.bfloat16, device=AB16.device)
        z3f = torch.empty((nt * 64, H), dtype=torch.bfloat16, device=AB16.device)
    else:
        save, Msave, d2f, z3f = False, None, None, None
    alloc = torch.zeros if (rowptr is None or E == 0) else torch.empty
    agg = alloc((n_rows, H), dtype=torch.float32, device=AB16.device)
    agg16 = alloc((n_rows, H), dtype=torch.bfloat16, device=AB16.device) if want16 else None
    if alloc is torch.empty:
        zero_empty_rows(rowptr, agg, H, agg16)
    s = torch.empty((max(E, 1), 1), dtype=torch.float32, device=AB16.device)
    bnd = torch.empty(((E + 63) // 64 * 2 + 2, H), dtype=torch.float32, device=AB16.device)      # boundary runs (deterministic sums)
    if save:
        _profiled("gcl_edge_fused_kernel<%d> E=%d saving (gather + 2 chained H x H contractions + segment-sum per edge; M, silu'(pre2), pre3 kept)" % (H, E),
                  4.0 * E * H * H,
                  lambda: check(_lib.load().fabind_gcl_edge_fused_train(ptr(AB16), _ld(AB16), H, ptr(row), ptr(col), ptr(rhohat), ptr(w_r),
                                                                        ptr(W2p), ptr(b2), ptr(Wcp), ptr(bc), ptr(w3), E, ptr(agg), ptr(s),
                                                                        float(p_drop), int(seed) & 0xFFFFFFFF, ptr(bnd), ptr(agg16),
                                                                        ptr(Msave), ptr(d2f), ptr(z3f), stream()),
                                "fabind_gcl_edge_fused_train"),
                  # per edge: 3 saved bf16 rows + row / col / rhohat / s; per node: the AB row once + agg fp32 (+ bf16)
                  E * (6.0 * H + 16) + n_rows * (4.0 * H + 4.0 * H + (2.0 * H if want16 else 0.0)),
                  algo=2.0 * n_rows * H * 2 + E * 8.0 + n_rows * 24.0)        # 2 N H s + E 8 + N 24 (s = 2: bf16)
    else:
        _profiled("gcl_edge_fused_kernel<%d> E=%d (gather + 2 chained H x H contractions + segment-sum per edge)" % (H, E), 4.0 * E * H * H,
                  lambda: check(_lib.load().fabind_gcl_edge_fused(ptr(AB16), _ld(AB16), H, ptr(row), ptr(col), ptr(rhohat), ptr(w_r),
                                                                  ptr(W2p), ptr(b2), ptr(Wcp), ptr(bc), ptr(w3), E, ptr(agg), ptr(s),
                                                                  float(p_drop), int(seed) & 0xFFFFFFFF, ptr(bnd), ptr(agg16), stream()),
                                "fabind_gcl_edge_fused"),
                  E * 16.0 + n_rows * (4.0 * H + 4.0 * H + (2.0 * H if want16 else 0.0)),
                  algo=2.0 * n_rows * H * 2 + E * 8.0 + n_rows * 24.0)
    out = (agg, s[:E], agg16) if want16 else (agg, s[:E])
    return out + ((Msave, d2f, z3f),) if save else out


def edge_lnfold(AB16, Kp, H, row, col, rho, stat, eps, w_r, c_r, c_c, dvec, p_drop=0.0, seed=0):
    """relu(W1 LN([h_r|h_c|rho]) + b1) from per-node projections (csrc/norm.hip: edge_lnfold_kernel) -> bf16 [E, Kp]."""
    E = row.shape[0]
    out = torch.empty((E, Kp), dtype=torch.bfloat16, device=AB16.device)
    assert AB16.dtype == torch.bfloat16 and stat.dtype == torch.float32 and stat.is_contiguous()
    check(_lib.load().fabind_edge_lnfold(ptr(AB16), _ld(AB16), Kp, H, ptr(row), ptr(col), ptr(rho), ptr(stat), float(eps),
                                         ptr(w_r), ptr(c_r), ptr(c_c), ptr(dvec), E, ptr(out), float(p_drop), int(seed) & 0xFFFFFFFF,
                                         stream()), "fabind_edge_lnfold")
    return out


def edge_lnfold_bwd(AB16, Kp, H, row, col, rho, stat, eps, w_r, c_r, c_c, out, dout, p_drop=0.0):
    """Adjoint of edge_lnfold (csrc/norm.hip: edge_lnfold_bwd_kernel) -> (du bf16 [E, Kp], es fp32 [E, 8] = (d m_r, d Q_r, d m_c, d Q_c, 0...),
    drho [E], vecs fp32 [4, Kp] = column sums d dvec, d c_r, d c_c, d w_r)."""
    E = row.shape[0]
    dev = AB16.device
    if E == 0:                                                   # (the library call is a no-op then: nothing would write the partials)
        return (torch.zeros((0, Kp), dtype=torch.bfloat16, device=dev), torch.zeros((0, 8), dtype=torch.float32, device=dev),
                torch.zeros(0, dtype=torch.float32, device=dev), torch.zeros((4, Kp), dtype=torch.float32, device=dev))
    lib = _lib.load()
    nb = max(1, int(lib.fabind_edge_lnfold_bwd_blocks(E)))
    du = torch.empty((max(E, 1), Kp), dtype=torch.bfloat16, device=dev)
    es = torch.empty((max(E, 1), 8), dtype=torch.float32, device=dev)
    drho = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    part = torch.empty((nb, 4 * Kp), dtype=torch.float32, device=dev)
    assert AB16.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and dout.dtype == torch.bfloat16 and dout.is_contiguous() and out.is_contiguous()
    check(lib.fabind_edge_lnfold_bwd(ptr(AB16), _ld(AB16), Kp, H, ptr(row), ptr(col), ptr(rho), ptr(stat), float(eps), ptr(w_r), ptr(c_r),
                                     ptr(c_c), ptr(out), ptr(dout), E, float(p_drop), ptr(du), ptr(es), ptr(drho), ptr(part), nb,
                                     stream()), "fabind_edge_lnfold_bwd")
    return du[:E], es[:E], drho[:E], colsum(part).view(4, Kp)


def inter_coord_fold(P16, H, col, rho, stat, q_w, eps, u, d, w3, p_drop=0.0, seed=0):
    """Per-edge coordinate scalar of FABind+'s inter-edge layer from per-node projections (csrc/norm.hip) -> fp32 [E]."""
    E = col.shape[0]
    out = torch.empty(max(E, 1), dtype=torch.float32, device=P16.device)
    assert P16.dtype == torch.bfloat16 and stat.dtype == torch.float32 and stat.is_contiguous()
    check(_lib.load().fabind_inter_coord_fold(ptr(P16), _ld(P16), H, ptr(col), ptr(rho), ptr(stat), float(q_w), float(eps),
                                              ptr(u), ptr(d), ptr(w3), E, ptr(out), float(p_drop), int(seed) & 0xFFFFFFFF,
                                              stream()), "fabind_inter_coord_fold")
    return out[:E]


def row_stats(x, eps):
    """(mean, rsqrt(var + eps)) per row of x [R,C] (csrc/norm.hip: row_stats_kernel) -> two fp32 [R]."""
    R, C = x.shape
    mu = torch.empty(max(R, 1), dtype=torch.float32, device=x.device)
    rs = torch.empty(max(R, 1), dtype=torch.float32, device=x.device)
    check(_lib.load().fabind_row_stats(ptr(x), dt_code(x.dtype), _ld(x), float(eps), R, C, ptr(mu), ptr(rs), stream()),
          "fabind_row_stats")
    return mu, rs


def pair_update_fused(T, b_off, p_node, c_node, z, Wop, bo, ln_w, ln_b, eps, W1p, b1, W2p, b2, Wbp=None, bb=None, p_drop=0.0,
                      seed=0):
    """FABind+ pair update for a ragged pair list (csrc/pair_fused.hip): z [pairs,H] bf16 -> (z' bf16, bias' [pairs,16] fp32
    or None).  Weights are pack_frag()-ed bf16; T = fp32 [nodes, ld] holding a32 at columns 0..31 and b32 at b_off..b_off+31."""
    n, H = z.shape
    assert z.dtype == torch.bfloat16 and z.is_contiguous() and T.dtype == torch.float32
    z_out = torch.empty_like(z)
    bias_out = torch.empty((max(n, 1), 16), dtype=torch.float32, device=z.device) if Wbp is not None else None
    a = _lib.PairUpdateArgs()
    for name, t in (("T", T), ("p_node", p_node), ("c_node", c_node), ("z_in", z), ("z_out", z_out), ("Wop", Wop), ("bo", bo),
                    ("ln_w", ln_w), ("ln_b", ln_b), ("W1p", W1p), ("b1", b1), ("W2p", W2p), ("b2", b2), ("Wbp", Wbp), ("bb", bb),
                    ("bias_out", bias_out)):
        setattr(a, name, ptr(t))
    a.ldt, a.b_off, a.n_pairs = _ld(T), int(b_off), n
    a.eps, a.p_drop, a.seed = float(eps), float(p_drop), int(seed) & 0xFFFFFFFF
    _profiled("pair_update_fused_kernel<%d> pairs=%d (Hadamard + LN + 2 chained H x H contractions + 16 biases)" % (H, n),
              2.0 * n * H * (2 * H + 32 + 16),
              lambda: check(_lib.load().fabind_pair_update_fused(ctypes.byref(a), H, stream()), "fabind_pair_update_fused"))
    return z_out, (bias_out[:n] if bias_out is not None else None)


_N_CU = {}
EDGE_BWD_GROUPS = 0     # development knob: persistent work-groups of the fused edge backward (0 = derived from CUs and LDS)
EDGE_BWD_TIMES = None   # set to an int64[12] device tensor to collect per-phase cycle counts (tools/edge_bwd_phases.py)


def gcl_edge_fused_bwd(AB16, H, row, col, rhohat, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm, p_drop=0.0, seed=0,
                       dab_bf16=False, w_dtype=torch.float32, rowptr=None, frags=None, saved=None):
    """Adjoint of gcl_edge_fused (csrc/fused_edge.hip): returns dAB [N,2H] fp32, drh [E], dw_r, dW2, db2, dWc, dbc, dw3.
    The five [E,H] bf16 operands it writes (S1, M, dT, dP2 for the weight gradients, dP1 for the sending-side
    reduction) are scratch that is released on return.
    saved = (M, d2f, z3f) of gcl_edge_fused(save=True): the two-contraction kernel (csrc/fused_edge_bwd4.hip) -- no forward recompute,
    M is the forward's, two work-groups per CU."""
    E, N, dev = row.shape[0], AB16.shape[0], AB16.device
    ng = _N_CU.get(dev)
    if ng is None:
        ng = _N_CU[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    lib = _lib.load()
    variant = lib.fabind_gcl_edge_fused_bwd_variant_for(H)
    if saved is not None and E > 0:
        variant = 6                                          # (chosen by the arguments: FabindEdgeBwdArgs.d2f / z3f)
    bm = 64                                                  # edges per tile of every form that is built (variants 0, 5, 6)
    if variant == 6:
        # one in-place [64][H] LDS tile + tables + partial sums, <= 128 VGPRs (four waves per SIMD): two work-groups per CU at H = 512
        per_cu = max(1, min(8, (160 * 1024) // (2 * bm * H + 16 * H + 32 * bm), 16 // max(1, H // 64)))
    elif variant == 5:
        # H/64 compute waves + one store wave per work-group, <= 168 VGPRs (three waves per SIMD): 12 waves per CU; two [64][H] LDS tiles
        per_cu = max(1, min(8, (160 * 1024) // (4 * bm * H + 16 * H + 32 * bm), 12 // (H // 64 + 1)))
    else:
        # variant 0: persistent work-groups per CU: as many as the two LDS tiles (2 * bm * H * 2 B of 160 KiB) and 2 waves/SIMD allow
        per_cu = max(1, min(8, (160 * 1024) // (4 * bm * H + 1024), 512 // H))
    ng = max(1, min(ng * per_cu, (E + bm - 1) // bm))
    if EDGE_BWD_GROUPS:
        ng = int(EDGE_BWD_GROUPS)       # development knob (tools/probes/edge_bwd_variants.py)
    if ng >= 8:
        ng -= ng % 8                # a multiple of 8 selects the XCD-aware walk (csrc/fused_edge.hip)
    d2scratch = torch.empty((2, ng, bm * H), dtype=torch.bfloat16, device=dev) if variant == 5 else None
    bnd = torch.empty(((E + bm - 1) // bm * 2 + 2, H), dtype=torch.float32, device=dev)       # boundary runs (deterministic sums)
    if variant == 6:
        buf = torch.empty((4, max(E, 1), H), dtype=torch.bfloat16, device=dev)
        S1, Mm, dT, dP2, dP1 = buf[0], saved[0], buf[1], buf[2], buf[3]
    else:
        buf = torch.empty((5, max(E, 1), H), dtype=torch.bfloat16, device=dev)
        S1, Mm, dT, dP2, dP1 = buf[0], buf[1], buf[2], buf[3], buf[4]
    dab_bf16 = bool(dab_bf16) and variant in (5, 6)     # both halves of dAB written as bf16 by their producers (no cast pass)
    # rowptr (CSR of `row`): columns [0, H) are written for nodes with edges (scan + fix-up), columns [H, 2H) for every node
    # (segment_sum) -- only the [0, H) halves of nodes without edges need zeroing
    dAB = (torch.zeros if (rowptr is None or E == 0) else torch.empty)((N, 2 * H), dtype=torch.bfloat16 if dab_bf16 else torch.float32, device=dev)
    if rowptr is not None and E > 0:
        zero_empty_rows(rowptr, dAB, H)
    drh = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
    part = torch.zeros((ng, 4, H), dtype=torch.float32, device=dev)
    dagg = dagg.contiguous()
    ds = ds.contiguous()
    a = _lib.EdgeBwdArgs()
    # frags: (W2p, Wcp, W2Tp, WcTp) packed with the rest of the model's operand packs in one launch (engine._build_stack_params)
    W2p, Wcp, W2Tp, WcTp = frags if (frags is not None and len(frags) == 4) else (pack_frag(W2), pack_frag(Wc), pack_frag(W2.t()), pack_frag(Wc.t()))
    keep = (W2p, Wcp, W2Tp, WcTp)
    for name, t in (("AB", AB16), ("row", row), ("col", col), ("rhohat", rhohat), ("w_r", w_r), ("W2p", W2p), ("Wcp", Wcp),
                    ("W2Tp", W2Tp), ("WcTp", WcTp), ("b2", b2), ("bc", bc), ("w3", w3), ("ds", ds), ("dagg", dagg),
                    ("S1", S1), ("Mm", Mm), ("dT", dT), ("dP2", dP2), ("dP1", dP1), ("drh", drh),
                    ("dABrow", None if dab_bf16 else dAB), ("dAB16", dAB if dab_bf16 else None),
                    ("part", part), ("d2scratch", d2scratch), ("bnd", bnd)):
        setattr(a, name, ptr(t))
    a.ldab, a.lddagg, a.lddab, a.E = _ld(AB16), _ld(dagg), _ld(dAB), E
    a.lddab16 = _ld(dAB)
    a.p_drop, a.seed = float(p_drop), int(seed) & 0xFFFFFFFF
    a.dbg = ptr(EDGE_BWD_TIMES) if (EDGE_BWD_TIMES is not None and variant != 6) else None
    if variant == 6:
        a.d2f, a.z3f = ptr(saved[1]), ptr(saved[2])
    _profiled(("gcl_edge_fused_bwd4_kernel<%d> E=%d (2 chained H x H contractions per edge over the forward's saved tiles)" % (H, E)) if variant == 6 else
              "gcl_edge_fused_bwd%s_kernel<%d> E=%d (recompute + 4 chained H x H contractions per edge)" % ("3" if variant == 5 else "", H, E),
              (4.0 if variant == 6 else 8.0) * E * H * H,
              lambda: check(_lib.load().fabind_gcl_edge_fused_bwd(ctypes.byref(a), H, ng, stream()),
                            "fabind_gcl_edge_fused_bwd"),
              # per edge: the [E,H] bf16 tiles read (saved form: silu'(pre2), pre3) and written (S1, dT, dP2, dP1; recompute form: + M) +
              # row / col / rhohat / ds / drh; per node: the AB row and dagg once, dAB's receiving half
              E * ((12.0 if variant == 6 else 10.0) * H + 20) + N * (4.0 * H + 4.0 * H + (2.0 if dab_bf16 else 4.0) * H),
              # SURVEY 8(d) names the forward's compulsory bytes only; the adjoint's, by the same rule (nothing of size [E, H] counts):
              # per node the AB row (2H bf16), the incoming d agg row (H fp32) and the outgoing d AB row (2H), per edge 20 B of indices / scalars
              algo=N * (2.0 * H * 2 + 4.0 * H + 2.0 * H * (2.0 if dab_bf16 else 4.0)) + E * 20.0)
    del keep
    if E > 0:
        if dab_bf16:
            segment_sum(dP1[:E], colptr, N, eidx=perm, out16=dAB[:, H:])
        else:
            segment_sum(dP1[:E], colptr, N, eidx=perm, out=dAB[:, H:])
        # (the split reduction writes the dtype autograd carries the weights in; small graphs: queued with the step's other weight gradients)
        if tn_can_queue(dP2[:E], S1[:E], W2.data_ptr()) and tn_can_queue(dT[:E], Mm[:E], Wc.data_ptr()):
            dW2, dWc = torch.empty((H, H), dtype=w_dtype, device=dev), torch.empty((H, H), dtype=w_dtype, device=dev)
            gemm_tn_queued(dP2[:E], S1[:E], dW2, key=W2.data_ptr())
            gemm_tn_queued(dT[:E], Mm[:E], dWc, key=Wc.data_ptr())
        else:
            TN_STATS["direct"] += 2
            dW2 = gemm_tn(dP2[:E], S1[:E], out_dtype=w_dtype)
            dWc = gemm_tn(dT[:E], Mm[:E], out_dtype=w_dtype)
    else:
        dW2 = torch.zeros((H, H), dtype=w_dtype, device=dev)
        dWc = torch.zeros((H, H), dtype=w_dtype, device=dev)
    ps = part.sum(0)
    out = (dAB, drh[:E], ps[3], dW2, ps[0], dWc, ps[1], ps[2])
    return out


_ZERO_PAGE = {}


_TN_SPLITS = {}


def _tn_splits(M, N, E, tn):
    """Split count of a TN contraction.  256x256 layout (one 8-wave work-group per CU): work-group ids go round-robin over the 8 XCDs
    with all tiles of an e-range on one XCD, so splits come in multiples of 8 and an XCD runs tiles * splits / 8 work-groups on its
    32 CUs in ceil(. / 32) rounds; the count minimises rounds x (k-steps per work-group) + the partial traffic (a small cost model
    in microseconds: 0.45 us per 32-row k-step, 3 us per round, 8 bytes per partial element at 3 TB/s; tools/probes/tn_layout_ab.sh).
    256x128 layout (two 4-wave work-groups per CU): round 1's rule."""
    key = (M, N, E, tn)
    s = _TN_SPLITS.get(key)
    if s is not None:
        return s
    if tn != 256:
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        s = max(1, min(256, max(E // 2048, min(E // 256, 16)), max(1, 1024 // tiles)))
    elif E < 4096:
        s = max(1, min(16, E // 256))
    else:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        best = None
        for c in range(8, 257, 8):
            if E // c < 256:
                break
            rounds = -(-(tiles * c // 8) // 32)
            cost = rounds * (-(-E // c) / 32.0 * 0.45 + 3.0) + c * M * N * 8 / 3e6
            if best is None or cost < best[0]:
                best = (cost, c)
        s = best[1] if best else 8
    _TN_SPLITS[key] = s
    return s


# ------------------------------------------------------------------------------------------------
# queued weight-gradient contractions (fabind_gemm_tn_multi)
# ------------------------------------------------------------------------------------------------
TN_DEFER = os.environ.get("FABIND_TN_DEFER", "1") == "1"          # 0: every contraction is its own launch pair (round 3's behaviour; A/B)
TN_DEFER_JOB_BYTES = int(os.environ.get("FABIND_TN_DEFER_JOB_MB", "1024")) << 20     # operands larger than this fill the chip alone: launched at once
TN_DEFER_QUEUE_BYTES = int(os.environ.get("FABIND_TN_DEFER_QUEUE_MB", "6144")) << 20   # operand bytes the queue may keep alive
_TNQ = {"jobs": [], "bytes": 0, "armed": False, "outs": set()}
_TNJOB = np.dtype([(n_, np.uint64) for n_ in ("Y", "X", "C_part", "out", "out_tail")] +
                  [(n_, np.int32) for n_ in ("ldy", "ldx", "M", "N", "E", "splits", "e_per", "n_tiles", "with_colsum", "out_dt", "ldo",
                                             "wg0", "n_wg", "blk0", "n_blk", "pad_")])        # = _lib.TnJob / FabindTnJob
TN_STATS = {"queued": 0, "flushes": 0, "direct": 0}


def tn_pending():
    return len(_TNQ["jobs"])


def _tn_multi_splits(E):
    """Split count of a queued contraction: the queue as a whole fills the chip, so a job is only cut to bound its work-groups'
    length (<= ~128 k-steps of 32 rows) -- in multiples of 8, which the XCD-aware work-group order of the kernel wants anyway.
    (Measured at the pocket-sized bench shape, 89 jobs: 4 splits for the 9,088-row jobs halve the reduction's partial traffic, 237 -> 159 us,
    but the contraction launch loses more on its tail, 1,364 -> 1,636 us.)"""
    return max(8, min(64, (E // 4096 + 7) // 8 * 8))


def gemm_tn_queued(Y, X, out, out_tail=None, key=None):
    """Queue out[:, :] (a [M, N] view, row stride out.stride(0)) = Y^T X (and out_tail[M] fp32 = Y^T 1) for the next `tn_flush()`.
    The outputs are NOT valid until then: ops.py registers flush hooks on every weight whose gradient is queued, and a callback at the
    end of the backward pass (`_tn_arm`).  key: identity of the weight the result is a gradient of -- a second gradient for the same
    weight while the first is still queued means autograd is about to ADD the two tensors: the caller must flush first."""
    E, M = Y.shape
    N = X.shape[1]
    nb = (Y.numel() + X.numel()) * 2
    q = _TNQ
    q["jobs"].append((Y, X, out, out_tail, M, N, E))
    q["bytes"] += nb
    if key is not None:
        q["outs"].add(key)
    TN_STATS["queued"] += 1
    if q["bytes"] > TN_DEFER_QUEUE_BYTES:
        tn_flush()


def _tn_end_of_backward():
    _TNQ["armed"] = False
    tn_flush()


def tn_arm():
    """Make sure the queue is flushed when the running backward pass ends (gradients that land in `.grad` without passing a hooked
    tensor).  -> False outside a backward pass: nothing may be queued then."""
    if _TNQ["armed"]:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_tn_end_of_backward)
    except RuntimeError:
        return False
    _TNQ["armed"] = True
    return True


def _tn_flush_hook(_g):
    tn_flush()


_TN_HOOKED = {}            # id(tensor) -> weakref(tensor): tensors that carry the flush hook (NOT an attribute on the tensor: a Parameter that is
#                             pickled whole -- torch.save(model), mp spawn -- would keep the attribute and lose the hook)


def _tn_in_backward():
    try:
        return torch._C._current_graph_task_id() != -1
    except Exception:
        return True          # cannot tell: keep the queue's state


def tn_hook(*weights):
    """Forward-time half of the queue's contract: a tensor hook on every weight (and bias) whose gradient the backward pass may
    queue.  The hook runs when autograd has that tensor's gradient complete and is about to hand it to the tensor's producer (the
    parameter pack's adjoint, a torch op composing weights, AccumulateGrad) -- i.e. before anything READS it -- and flushes the queue.
    (A LEAF handed to an op directly flushes at every such op: AccumulateGrad clones a gradient tensor somebody else still holds --
    the queue does -- so even a first accumulation reads it.  The engines route weights AND biases through the parameter pack, whose
    adjoint is then the one flush point of the pass.)"""
    if not TN_DEFER:
        return
    q = _TNQ
    if (q["armed"] or q["jobs"]) and not _tn_in_backward():
        # forward-time call with the queue still armed / loaded: the last backward pass raised (autograd runs no final callbacks then) or
        # was abandoned.  Its leftover jobs would write into gradients nobody will read and keep their operands alive: drop them.
        q["jobs"], q["bytes"], q["outs"], q["armed"] = [], 0, set(), False
    import weakref
    for W in weights:
        if W is None or not W.requires_grad:
            continue
        k = id(W)
        r = _TN_HOOKED.get(k)
        if r is not None and r() is W:
            continue
        W.register_hook(_tn_flush_hook)
        _TN_HOOKED[k] = weakref.ref(W, lambda _r, k_=k: _TN_HOOKED.pop(k_, None))


def tn_can_queue(Y, X, key=None):
    """Is this contraction one for the queue?  (bf16 operands on the 256 x 256 kernel layout, not large enough to fill the chip alone,
    no gradient for the same weight already waiting.)"""
    if not TN_DEFER or (Y.numel() + X.numel()) * 2 > TN_DEFER_JOB_BYTES or _lib.load().fabind_gemm_tn_tile_n() != 256:
        return False
    if key is not None and key in _TNQ["outs"]:
        tn_flush()
        return False
    return tn_arm()


def tn_flush():
    """Launch everything queued by gemm_tn_queued: one contraction launch over all jobs + one reduction launch."""
    q = _TNQ
    jobs, q["jobs"], q["bytes"], q["outs"] = q["jobs"], [], 0, set()
    if not jobs:
        return
    from .param_pack import _upload
    dev = jobs[0][0].device
    zp = _ZERO_PAGE.get(dev)
    if zp is None:
        zp = _ZERO_PAGE[dev] = torch.zeros(256, dtype=torch.bfloat16, device=dev)
    n = len(jobs)
    # long work-groups first: the tail of the launch is made of the short ones
    jobs.sort(key=lambda j_: -(j_[6] // _tn_multi_splits(j_[6])))
    tab = np.zeros(n, dtype=_TNJOB)
    E = np.array([j_[6] for j_ in jobs], dtype=np.int64)
    M = np.array([j_[4] for j_ in jobs], dtype=np.int64)
    N = np.array([j_[5] for j_ in jobs], dtype=np.int64)
    cs = np.array([j_[3] is not None for j_ in jobs], dtype=np.int64)
    sp = np.maximum(8, np.minimum(64, (E // 4096 + 7) // 8 * 8))
    per = M * N + cs * M
    tab["Y"], tab["X"] = [j_[0].data_ptr() for j_ in jobs], [j_[1].data_ptr() for j_ in jobs]
    tab["out"] = [j_[2].data_ptr() for j_ in jobs]
    tab["out_tail"] = [j_[3].data_ptr() if j_[3] is not None else 0 for j_ in jobs]
    tab["ldy"], tab["ldx"] = [j_[0].stride(0) for j_ in jobs], [j_[1].stride(0) for j_ in jobs]
    tab["ldo"], tab["out_dt"] = [j_[2].stride(0) for j_ in jobs], [dt_code(j_[2].dtype) for j_ in jobs]
    tab["M"], tab["N"], tab["E"], tab["splits"], tab["with_colsum"] = M, N, E, sp, cs
    tab["e_per"] = ((E + sp - 1) // sp + 31) // 32 * 32
    tab["n_tiles"] = ((M + 255) // 256) * ((N + 255) // 256)
    n_wg = tab["n_tiles"].astype(np.int64) * ((sp + 7) // 8 * 8)
    n_blk = (per + 4095) // 4096
    tab["n_wg"], tab["n_blk"] = n_wg, n_blk
    tab["wg0"], tab["blk0"] = np.cumsum(n_wg) - n_wg, np.cumsum(n_blk) - n_blk
    wg, blk = int(n_wg.sum()), int(n_blk.sum())
    psz = sp * per
    part = torch.empty(int(psz.sum()), dtype=torch.float32, device=dev)
    tab["C_part"] = part.data_ptr() + 4 * (np.cumsum(psz) - psz)
    jb = tab.view(np.uint8)
    wg_job = np.repeat(np.arange(n, dtype=np.int32), n_wg)
    blk_job = np.repeat(np.arange(n, dtype=np.int32), n_blk)
    pad = (-jb.shape[0]) % 16
    flat = np.concatenate([jb, np.zeros(pad, dtype=np.uint8), wg_job.view(np.uint8), blk_job.view(np.uint8)])
    tdev = _upload(flat, dev)
    p0 = tdev.data_ptr()
    p_wg = p0 + jb.shape[0] + pad
    launch = lambda: check(_lib.load().fabind_gemm_tn_multi(p0, tab.ctypes.data, n, p_wg, wg, p_wg + 4 * wg_job.shape[0], blk,
                                                            ptr(zp), stream()), "fabind_gemm_tn_multi")
    if PROFILE is None:
        launch()
    else:
        _profiled("fabind_gemm_tn_multi jobs=%d" % n, sum(2.0 * j_[4] * j_[5] * j_[6] for j_ in jobs), launch,
                  sum(2.0 * (j_[4] + j_[5]) * j_[6] for j_ in jobs))
    TN_STATS["flushes"] += 1
    del tdev, part, jobs


def gemm_tn(Y, X, splits=None, out_dtype=torch.float32, with_colsum=False):
    """sum_e Y[e,:]^T X[e,:] -> [M, N] as out_dtype (bf16 operands, LDS transpose reads; no materialised transposes).
    with_colsum: also the column sums of Y, [M] fp32, from the same launches -> (dW, colsum)."""
    E, M = Y.shape
    N = X.shape[1]
    assert X.shape[0] == E and Y.dtype == torch.bfloat16 and X.dtype == torch.bfloat16
    dev = Y.device
    zp = _ZERO_PAGE.get(dev)
    if zp is None:
        zp = _ZERO_PAGE[dev] = torch.zeros(256, dtype=torch.bfloat16, device=dev)
    if splits is None:
        splits = _tn_splits(M, N, E, _lib.load().fabind_gemm_tn_tile_n())
    n = M * N + (M if with_colsum else 0)
    part = torch.empty((splits, n), dtype=torch.float32, device=dev)
    launch = lambda: check(_lib.load().fabind_gemm_tn(ptr(Y), _ld(Y), ptr(X), _ld(X), ptr(part), M, N, E, splits, ptr(zp),
                                                      None, 0, 1 if with_colsum else 0, stream()), "fabind_gemm_tn")
    if PROFILE is None:
        launch()
    else:
        _profiled("fabind_gemm_tn M=%d N=%d E=%d" % (M, N, E), 2.0 * M * N * E, launch, 2.0 * (M + N) * E + 4.0 * splits * M * N)
    if with_colsum and n % 4 == 0 and M % 4 == 0 and (splits > 1 or out_dtype != torch.float32):
        # one reduction launch: the weight part as out_dtype, the column sums behind it as fp32
        dW = torch.empty((M, N), dtype=out_dtype, device=dev)
        db = torch.empty(M, dtype=torch.float32, device=dev)
        check(_lib.load().fabind_split_sum(ptr(part), splits, n, ptr(dW), dt_code(out_dtype), M, ptr(db), stream()), "fabind_split_sum")
        return dW, db
    if splits == 1:
        out = part[0] if out_dtype == torch.float32 else part[0].to(out_dtype)
    elif n % 4 == 0:                                       # the split reduction writes the dtype the consumer carries (no cast kernel)
        out = torch.empty(n, dtype=out_dtype, device=dev)
        check(_lib.load().fabind_split_sum(ptr(part), splits, n, ptr(out), dt_code(out_dtype), 0, None, stream()), "fabind_split_sum")
    else:
        out = colsum(part)
        out = out if out_dtype == torch.float32 else out.to(out_dtype)
    if with_colsum:
        return out[:M * N].view(M, N), out[M * N:].float()
    return out.view(M, N)
