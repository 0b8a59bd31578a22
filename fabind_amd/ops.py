"""Differentiable operators of the FABind hot path, each backed by kernels of libfabind_hip.so.

Forward-only calls (torch.no_grad(), 7 of the 8 refinement iterations) go straight to the kernels;
when a gradient is required the same kernels run inside torch.autograd.Function wrappers whose
backward passes are HIP kernels as well (fabind_amd/csrc/bwd.hip + the GEMM family)."""
import ctypes

import os

import numpy as np
import torch

from . import config as _cfg
from . import kernels as K
from ._lib import GemmArgs, check, dt_code, load, ptr, stream


def act_dtype():
    """Storage type of the large edge-/pair-level intermediates."""
    return torch.float32 if _cfg.fp32_storage() else torch.bfloat16


def mm_dtype():
    return torch.float32 if _cfg.fp32_storage() else torch.bfloat16


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def needs_grad(*ts):
    return _needs_grad(*ts)


def _nchunk(rows):
    return max(1, min(1024, (rows + 255) // 256))


def _ksplit(R):
    """(splits, padded K) for a contraction over R rows (weight gradients: tiny M x N, huge K)."""
    S = max(1, min(128, R // 4096))
    unit = 64 * S
    return S, (R + unit - 1) // unit * unit


def _transposed(x, act=K.ACT_NONE, Rp=None):
    """act(x)^T as an mm-dtype [C, Rp] matrix (zero padded) for K=R contractions."""
    R, C = x.shape
    Rp = Rp or _ksplit(R)[1]
    out = torch.empty((C, Rp), dtype=mm_dtype(), device=x.device)
    if Rp != R:
        out[:, R:].zero_()
    check(load().fabind_transpose_act(ptr(x), dt_code(x.dtype), x.stride(0), ptr(out), dt_code(out.dtype), Rp, R, C, act,
                                      stream()), "fabind_transpose_act")
    return out


FUSE_DB_TN = os.environ.get("FABIND_FUSE_DB_TN", "1") == "1"   # bias gradients ride along with the TN weight-gradient contraction


def _tn_ok(dpre, x, x2):
    bf = torch.bfloat16
    return (dpre.dtype == bf and x.dtype == bf and (x2 is None or x2.dtype == bf)
            and dpre.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and (x2 is None or x2.shape[1] % 8 == 0)
            and dpre.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dpre.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def _tn_queue(dpre, x, x2, out_dtype, want_db, W):
    """The TN contraction(s) of one weight gradient put on the queue of kernels.tn_flush (bf16 operands) -> (dW, db) whose contents
    become valid at the flush; None when the contraction is not one for the queue (large operands, no backward pass running, ...)."""
    if W is None or not K.tn_can_queue(dpre, x if x2 is None or x.numel() >= x2.numel() else x2, W.data_ptr()):
        return None
    M, K1, K2 = dpre.shape[1], x.shape[1], (x2.shape[1] if x2 is not None else 0)
    dW = torch.empty((M, K1 + K2), dtype=out_dtype, device=dpre.device)
    db = torch.empty(M, dtype=torch.float32, device=dpre.device) if want_db else None
    K.gemm_tn_queued(dpre, x, dW[:, :K1] if K2 else dW, db, key=W.data_ptr())
    if K2:
        K.gemm_tn_queued(dpre, x2, dW[:, K1:], None, key=W.data_ptr())
    return dW, db


def _weight_grad(dpre, x, act_pro, x2=None, out_dtype=torch.float32, want_db=False, W=None):
    """dW = dpre^T [act(x) | x2] as out_dtype.  bf16: TN contraction with LDS transpose reads (no transposed copies), its split
    reduction writing out_dtype directly; fp32 parity mode / odd widths: explicit transposes + split-K NT GEMMs over the (padded)
    row dimension.  want_db: -> (dW, db) with db = dpre^T 1 (fp32) -- from the SAME contraction launches when the TN path runs
    (`fabind_gemm_tn(..., with_colsum)`; the split reduction writes dW as out_dtype and db as fp32), a separate column sum otherwise.
    W: the weight this is the gradient of (a tensor `kernels.tn_hook` was called on in forward): contractions that do not fill the chip
    alone are queued and run with the rest of the backward pass's weight gradients as one launch (kernels.tn_flush)."""
    if not want_db:
        return _weight_grad_f32(dpre, x, act_pro, x2, out_dtype, W)
    if FUSE_DB_TN and act_pro == K.ACT_NONE and _x3_tn_ok(dpre, x, x2) and dpre.shape[1] % 4 == 0:
        return _x3_weight_grad(dpre, x, x2, out_dtype, True, W)
    if FUSE_DB_TN and act_pro == K.ACT_NONE and _tn_ok(dpre, x, x2):
        q = _tn_queue(dpre, x, x2, out_dtype, True, W)
        if q is not None:
            return q
        dW, db = K.gemm_tn(dpre, x, out_dtype=out_dtype, with_colsum=True)
        if x2 is not None:
            dW = torch.cat([dW, K.gemm_tn(dpre, x2, out_dtype=out_dtype)], 1)
        return dW, db
    return _weight_grad_f32(dpre, x, act_pro, x2, out_dtype, W), K.colsum(dpre)




def _x3_tn_ok(dpre, x, x2):
    """'bf16x3' mode: the weight-gradient contraction dW = dpre^T x runs on bf16 ROUNDINGS of its two fp32 operands through the
    TN kernel (fp32 accumulation over the rows; no transposed copies).  Every weight-gradient entry is a sum over 1e5 .. 1e6 rows
    with heavy cancellation, so single-bf16 operands cost ~2e-3 of the entry -- the level at which this mode's gradients differ
    from exact fp32 anyway (a ReLU unit whose pre-activation sits within the 2^-17 forward error of zero flips its derivative;
    tests/test_gpu_headline.py prints both) -- against explicit transposes + three-MFMA contractions at 4x the time.  The forward
    pass and the activation-gradient chain stay split-bf16.  config.set_x3_backward('exact') (or FABIND_X3_WGRAD=x3) restores the split contraction."""
    f32 = torch.float32
    return (_cfg.x3_backward_bf16() and _cfg.get_precision() == "bf16x3" and dpre.dtype == f32 and x.dtype == f32 and (x2 is None or x2.dtype == f32)
            and dpre.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and (x2 is None or x2.shape[1] % 8 == 0) and dpre.shape[0] >= 256)


def _b16_copy(x):
    """bf16 copy of an fp32 activation, remembered on the tensor (an epilogue may already have attached one: _attach_b16)."""
    if x.dtype == torch.bfloat16:
        return x
    c = getattr(x, "_fab_b16", None)
    if c is not None and c[0] == x._version and c[1].shape == x.shape:
        return c[1]
    y = x.to(torch.bfloat16)
    if x.dim() == 2 and x.is_contiguous():
        try:
            x._fab_b16 = (x._version, y)
        except Exception:
            pass
    return y


def _x3_weight_grad(dpre, x, x2, out_dtype, want_db, W=None):
    """-> (dW, db or None): the 'bf16x3' weight gradient on bf16 roundings (see _x3_tn_ok); db rides along as the TN kernel's column sums."""
    d16 = dpre.to(torch.bfloat16)
    x16, x216 = _b16_copy(x), (_b16_copy(x2) if x2 is not None else None)
    q = _tn_queue(d16, x16, x216, out_dtype, want_db, W)
    if q is not None:
        return q
    if want_db:
        dW, db = K.gemm_tn(d16, x16, out_dtype=out_dtype, with_colsum=True)
    else:
        dW, db = K.gemm_tn(d16, x16, out_dtype=out_dtype), None
    if x2 is not None:
        dW = torch.cat([dW, K.gemm_tn(d16, x216, out_dtype=out_dtype)], 1)
    return dW, db


def _weight_grad_f32(dpre, x, act_pro, x2, out_dtype, W=None):
    if act_pro == K.ACT_NONE and _x3_tn_ok(dpre, x, x2):
        return _x3_weight_grad(dpre, x, x2, out_dtype, False, W)[0]
    if act_pro == K.ACT_NONE and _tn_ok(dpre, x, x2):
        q = _tn_queue(dpre, x, x2, out_dtype, False, W)
        if q is not None:
            return q[0]
        if x2 is None:
            return K.gemm_tn(dpre, x, out_dtype=out_dtype)
        return torch.cat([K.gemm_tn(dpre, x, out_dtype=out_dtype), K.gemm_tn(dpre, x2, out_dtype=out_dtype)], 1)
    if out_dtype != torch.float32:
        return _weight_grad_f32(dpre, x, act_pro, x2, torch.float32).to(out_dtype)
    S, Rp = _ksplit(dpre.shape[0])
    dpt = _transposed(dpre, Rp=Rp)
    xt = _transposed(x, act_pro, Rp=Rp)
    if x2 is not None:
        xt = torch.cat([xt, _transposed(x2, Rp=Rp)], 0)
    if S == 1:
        return K.gemm(dpt, xt)[0]
    part, _ = K.gemm(dpt, xt, k_splits=S)                                  # [S, N, K] fp32 partials
    return K.colsum(part.reshape(S, -1)).reshape(part.shape[1], part.shape[2])


# ------------------------------------------------------------------------------------------------
# linear:  y = act_epi( [x|x2] W^T + b ) (+ residual)
# Under autograd the kernel also emits act_epi'(pre-activation) (SiLU) so that backward is a plain multiply,
# and fp32 activations are fed to the MFMA kernels as bf16 copies in bf16 mode (also what is saved).
# ------------------------------------------------------------------------------------------------
def _mul_dact(dy, aux, act, out_dtype, scale=1.0):
    out = torch.empty(dy.shape, dtype=out_dtype, device=dy.device)
    a = aux if aux is not None else dy
    check(load().fabind_mul_dact(ptr(dy), dt_code(dy.dtype), ptr(a), dt_code(a.dtype), act, ptr(out), dt_code(out_dtype),
                                 dy.numel(), float(scale), stream()), "fabind_mul_dact")
    return out


FUSE_DB = os.environ.get("FABIND_FUSE_DB", "1") == "1"
DB_IN_TN_ROWS = int(os.environ.get("FABIND_DB_IN_TN_ROWS", "32768"))


def _db_in_tn(dy):
    """Should a Linear's bias gradient ride with its (queued) weight-gradient contraction instead of with the pass over dy?  On small
    batches yes: the pass then is a plain cast / activation adjoint -- one launch instead of the column-partials pair -- and the column
    sums cost the queued contraction nothing visible; on large ones the fused pass is the cheaper place (its partials are free next
    to 300 MB of traffic, the contraction's VALU slots are not)."""
    return K.TN_DEFER and dy.shape[0] <= DB_IN_TN_ROWS and dy.dtype in (torch.float32, torch.bfloat16) and mm_dtype() == torch.bfloat16


def _mul_dact_colsum(dy, aux, act, out_dtype, scale=1.0):
    """(scale * dy * act'(aux), its column sums) in one pass (bias gradient fused into the activation adjoint / cast)."""
    R, C = dy.shape
    out = torch.empty((R, C), dtype=out_dtype, device=dy.device)
    nchunk = max(1, min(4096, (R + 63) // 64))           # >= 2 blocks per SIMD at C = 512: the pass is latency-bound
    scratch = torch.empty((nchunk, C), dtype=torch.float32, device=dy.device)
    db = torch.empty(C, dtype=torch.float32, device=dy.device)
    check(load().fabind_mul_dact_colsum(ptr(dy), dt_code(dy.dtype), ptr(aux), dt_code(aux.dtype) if aux is not None else 0,
                                        act, ptr(out), dt_code(out_dtype), R, C, ptr(db), ptr(scratch), nchunk, float(scale), stream()),
          "fabind_mul_dact_colsum")
    return out, db


def _mul_dropmask_colsum(dy, p_drop, seed, out_dtype):
    """(dy * keep(seed, r, c) / (1 - p), its column sums): the adjoint of a GEMM epilogue's dropout with the mask regenerated from its key."""
    R, C = dy.shape
    out = torch.empty((R, C), dtype=out_dtype, device=dy.device)
    nchunk = max(1, min(4096, (R + 63) // 64))
    scratch = torch.empty((nchunk, C), dtype=torch.float32, device=dy.device)
    db = torch.empty(C, dtype=torch.float32, device=dy.device)
    check(load().fabind_mul_dropmask_colsum(ptr(dy), dt_code(dy.dtype), ptr(out), dt_code(out_dtype), R, C, float(p_drop), int(seed) & 0xFFFFFFFF,
                                            ptr(db), ptr(scratch), nchunk, stream()), "fabind_mul_dropmask_colsum")
    return out, db


def _mm_in(x):
    """Operand as the GEMM wants it: bf16 copy of an fp32 activation in bf16 mode.  The copy is remembered on the tensor
    (keyed on its version counter, so an in-place update invalidates it): a residual-stream tensor feeds two or three
    Linears per layer, and each of them used to cast it again."""
    if x is not None and _cfg.get_precision() == "bf16" and x.dtype == torch.float32:
        c = getattr(x, "_fab_b16", None)
        if c is not None and c[0] == x._version:
            return c[1]
        y = x.to(torch.bfloat16)
        if x.dim() == 2 and x.is_contiguous():
            try:
                x._fab_b16 = (x._version, y)
            except Exception:
                pass
        return y
    return x


def _wt(W):
    """W^T as a contiguous operand: the copy the parameter pack wrote next to W (param_pack.ParamPack.cat(with_T=True)), else a transpose kernel."""
    t = getattr(W, "_fab_T", None)
    return t if (t is not None and t.shape == (W.shape[1], W.shape[0])) else W.t().contiguous()


def _attach_b16(y, y16):
    """Remember y16 as the bf16 operand copy of the fp32 tensor y (what _mm_in would otherwise produce with a cast kernel)."""
    if y16 is not None:
        y._fab_b16 = (y._version, y16)
    return y


def _want16(out_dtype, act_epi, want):
    """Should the epilogue also emit the bf16 copy of an fp32 output?  'bf16': it is the next GEMM's operand; 'bf16x3': it is the
    operand of the weight-gradient TN contractions (ops._x3_tn_ok), saving their cast passes."""
    return bool(want) and _cfg.get_precision() in ("bf16", "bf16x3") and out_dtype == torch.float32 and act_epi == K.ACT_NONE


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, x2, residual, act_epi, out_dtype, holder=None, p_drop=0.0, seed=0, W32=None):
        assert act_epi in (K.ACT_NONE, K.ACT_RELU, K.ACT_SILU)
        # epilogue dropout under autograd.  ReLU outputs: y = relu(pre) keep / (1 - p) is what is saved, its zeros ARE the dropped
        # positions (and the inactive units), so the adjoint is [y > 0] / (1 - p): no mask is stored or recomputed.  No activation, with
        # or without a residual (round 5: the reference's train-mode dropouts AHEAD of `h + ...`, egnn.py:106, cross_att.py:128): fp32
        # output through the fp32 epilogue, the adjoint regenerates the counter-based mask from (seed, row, col) -- _mul_dropmask_colsum
        drop_regen = p_drop > 0.0 and act_epi == K.ACT_NONE
        assert p_drop == 0.0 or (act_epi == K.ACT_RELU and residual is None) or \
            (drop_regen and out_dtype == torch.float32 and x2 is None and W32 is None and W.shape[0] % 4 == 0), \
            "epilogue dropout under autograd: ReLU without a residual, or no activation with an fp32 output"
        thr = int(p_drop * 65536.0 + 0.5)
        ctx.drop_scale = 1.0 / (1.0 - thr / 65536.0)
        ctx.drop_regen = (float(p_drop), int(seed)) if drop_regen else None
        xin, x2in = _mm_in(x), _mm_in(x2)
        D = None
        M, N = x.shape[0], W.shape[0]
        y = torch.empty((M, N), dtype=out_dtype, device=x.device)
        # ReLU followed by a residual (FABind+ MLPwithLastAct): y = relu(pre) + R no longer tells where pre > 0, so
        # the derivative is stored like for SiLU
        relu_res = act_epi == K.ACT_RELU and residual is not None
        if act_epi == K.ACT_SILU or relu_res:
            D = torch.empty((M, N), dtype=out_dtype, device=x.device)
        y16 = None
        if holder is not None:                        # the caller wants the epilogue to emit the bf16 operand copy of y as well
            y16 = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
            holder.append(y16)
        if isinstance(W32, tuple):
            _gemm_split_rows(x, xin, W, W32, b, y, y16)
        elif W32 is not None:    # a split-precision site of the bf16 mode (config.set_split_sites): the forward contracts the fp32 activation with
            #                      the fp32 master weight; what is saved -- and everything the backward does -- are the bf16 operands
            K.gemm(x, W32, bias=b, A2=x2, act_epi=act_epi, residual=residual, out=y, out2=D, out16=y16, p_drop=p_drop, seed=seed, force_x3=True)
        else:
            K.gemm(xin, W, bias=b, A2=x2in, act_epi=act_epi, residual=residual, out=y, out2=D, out16=y16, p_drop=p_drop, seed=seed)
        ctx.act_epi, ctx.x_dtype = (K.ACT_SILU if relu_res else act_epi), x.dtype       # backward: stored derivative
        ctx.x2_dtype = x2.dtype if x2 is not None else None
        ctx.has_b, ctx.has_res, ctx.has_x2 = b is not None, residual is not None, x2 is not None
        ctx.res_dtype = residual.dtype if residual is not None else None
        ctx.sink_x, ctx.sink_res = _sink_of(x), _sink_of(residual)
        ctx.x_shape = x.shape
        ctx.save_for_backward(xin, W, x2in, y if (act_epi == K.ACT_RELU and not relu_res) else None, D)
        ctx.Wt = getattr(W, "_fab_T", None)           # W^T written by the parameter pack next to W, if any
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, x2, y, D = ctx.saved_tensors
        dy = dy.contiguous()
        md = mm_dtype()
        db = None
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        fuse_db = FUSE_DB and want_db and dy.dim() == 2 and dy.shape[1] % 4 == 0   # bias gradient in the same pass over dy
        if fuse_db and ctx.needs_input_grad[1] and _db_in_tn(dy):
            fuse_db = False                                                       # ... or with the queued weight-gradient contraction
        md_op = _mul_dact_colsum if fuse_db else (lambda *a: (_mul_dact(*a), None))      # (dy, aux, act, out dtype[, scale])
        if ctx.drop_regen is not None:
            dpre, db = _mul_dropmask_colsum(dy, ctx.drop_regen[0], ctx.drop_regen[1], md)
            if not want_db:
                db = None
        elif ctx.act_epi == K.ACT_RELU:
            dpre, db = md_op(dy, y, K.ACT_RELU, md, ctx.drop_scale)
        elif ctx.act_epi == K.ACT_SILU:
            dpre, db = md_op(dy, D, K.ACT_STORED_DERIV, md)
        elif dy.dtype != md:
            dpre, db = md_op(dy, None, K.ACT_NONE, md)           # dtype conversion only
        else:
            dpre = dy
        K1, N = x.shape[1], W.shape[0]
        dx = dx2 = dW = None
        # x with a shared gradient buffer (ops.shared_grad): its input gradient is stored / accumulated there by the GEMM epilogue.
        # fp32 tensors (the v1 residual stream) and bf16 ones (the FABind+ pair embedding: next layer's residual deposit + this
        # Linear's gradient + the inter-edge row gather meet in ONE epilogue pass instead of two [pairs, H] adds)
        sink_x = ctx.sink_x if (ctx.x_dtype in (torch.float32, torch.bfloat16) and ctx.needs_input_grad[0]) else None
        if ctx.needs_input_grad[0] or (ctx.has_x2 and ctx.needs_input_grad[3]):
            # few output columns over many rows (FABind+: the 16 pair-bias columns of 6e5 .. 4e6 pairs): a contraction dim that is not
            # a multiple of 64 sends the input-gradient GEMM to the register-staged fallback kernel, which wrote the [M, K] gradient
            # at < 1 TB/s (756 us for 648 MB); zero-padded to 64 it runs on the LDS-DMA kernel with the row-contiguous epilogue
            pad64 = N % 64 != 0 and N <= 64 and dpre.shape[0] >= 65536 and dpre.dtype == torch.bfloat16
            if N % 8 == 0 and not pad64:
                Wt, dmm = (ctx.Wt if ctx.Wt is not None else W.t().contiguous()), dpre      # [K, N] (parameter-only transpose)
            else:                                                       # tiny heads (N = 1): pad the contraction dim to 8
                Np = 64 if pad64 else (N + 7) // 8 * 8
                Wt = torch.zeros((W.shape[1], Np), dtype=W.dtype, device=W.device)
                Wt[:, :N] = W.t()
                dmm = torch.zeros((dpre.shape[0], Np), dtype=dpre.dtype, device=dpre.device)
                dmm[:, :N] = dpre
            if sink_x is not None:
                dx = sink_x.gemm_into(dmm, Wt[:K1], x, ctx.x_dtype)     # (x2, if any, gets its own GEMM over the other K-slice of W)
                if ctx.has_x2 and ctx.needs_input_grad[3]:
                    dx2, _ = K.gemm(dmm, Wt[K1:], out_dtype=torch.float32 if ctx.x2_dtype == torch.float32 else ctx.x2_dtype)
            else:
                dfull, _ = K.gemm(dmm, Wt, out_dtype=torch.float32 if ctx.x_dtype == torch.float32 else ctx.x_dtype)
                if ctx.has_x2:
                    dx, dx2 = dfull[:, :K1], dfull[:, K1:]
                    if ctx.x2_dtype != dx2.dtype:
                        dx2 = dx2.to(ctx.x2_dtype)
                else:
                    dx = dfull
        if ctx.needs_input_grad[1] and want_db and db is None:
            dW, db = _weight_grad(dpre, x, K.ACT_NONE, x2, W.dtype, want_db=True, W=W)
        elif ctx.needs_input_grad[1]:
            dW = _weight_grad(dpre, x, K.ACT_NONE, x2, W.dtype, W=W)
        if want_db and db is None:
            db = K.colsum(dpre)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)      # (a bf16 residual stream takes dy as it is)
        if dres is not None and ctx.sink_res is not None:
            dres = ctx.sink_res.deposit(dres)
        return dx, dW, db, dx2, dres, None, None, None, None, None, None


def _gemm_split_rows(x, xin, W, W32, b, y, y16):
    """y = x W^T + b with the output columns [lo, hi) contracted in split precision (fp32 x, fp32 master rows W32p) and the others on the
    bf16 operands: two or three launches into column ranges of the same output (and of its bf16 copy).  The inter-edge attention's
    q | k | v projection: only the v columns carry the ligand-RMSD gap (tools/probes/precision_qkv_parts.py)."""
    W32p, lo, hi = W32
    N = W.shape[0]
    for a_, b_ in ((0, lo), (hi, N)):
        if b_ > a_:
            K.gemm(xin, W[a_:b_], bias=b[a_:b_] if b is not None else None, out=y[:, a_:b_], out16=y16[:, a_:b_] if y16 is not None else None)
    K.gemm(x, W32p, bias=b[lo:hi] if b is not None else None, out=y[:, lo:hi], out16=y16[:, lo:hi] if y16 is not None else None, force_x3=True)


def _split_site_ok(x, W, W32, x2, act_pro, out_dtype, p_drop, act_epi=K.ACT_NONE, residual=None):
    """May this Linear run as a split-precision site?  (bf16 mode, fp32 activation rows and fp32 master weight of W's shape, plain epilogue;
    the row-range form (W32p, lo, hi): no second operand, activation or residual.)"""
    if W32 is None or _cfg.get_precision() != "bf16" or x.dtype != torch.float32 or act_pro != K.ACT_NONE or out_dtype != torch.float32 \
            or p_drop != 0.0 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        return False
    if isinstance(W32, tuple):
        W32p, lo, hi = W32
        return (W32p.dtype == torch.float32 and W32p.stride(-1) == 1 and tuple(W32p.shape) == (hi - lo, W.shape[1]) and 0 <= lo < hi <= W.shape[0]
                and lo % 128 == 0 and hi % 128 == 0 and x2 is None and act_epi == K.ACT_NONE and residual is None)
    return (W32.dtype == torch.float32 and tuple(W32.shape) == tuple(W.shape) and W32.stride(-1) == 1 and (x2 is None or x2.dtype == torch.float32))


def linear(x, W, b=None, act_pro=K.ACT_NONE, act_epi=K.ACT_NONE, residual=None, x2=None, out_dtype=torch.float32,
           p_drop=0.0, want16=False, W32=None):
    """p_drop > 0 (no-grad paths only): dropout on act(x W^T + b) inside the GEMM epilogue, before the residual; the mask is
    a counter-based hash keyed by a seed drawn from torch's CPU generator.
    W32 (bf16 mode): the fp32 master of W -- the forward contraction then runs in split precision on the fp32 activation x (a
    split-precision site, config.set_split_sites); the backward is the bf16 one either way."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0.0 else 0
    if not _split_site_ok(x, W, W32, x2, act_pro, out_dtype, p_drop, act_epi, residual):
        W32 = None
    if _needs_grad(x, W, b, x2, residual):
        assert act_pro == K.ACT_NONE, "producer-side activations only under autograd"
        K.tn_hook(W, b)
        holder = [] if _want16(out_dtype, act_epi, want16) else None
        y = _Linear.apply(x, W, b, x2, residual, act_epi, out_dtype, holder, p_drop, seed, W32)
        return _attach_b16(y, holder[0] if holder else None)
    y16 = None
    if _want16(out_dtype, act_epi, want16) and act_pro == K.ACT_NONE and (p_drop == 0.0 or (act_epi == K.ACT_NONE and x2 is None and W.shape[0] % 4 == 0)):
        y16 = torch.empty((x.shape[0], W.shape[0]), dtype=torch.bfloat16, device=x.device)      # (with epilogue dropout: the plain fp32 epilogues only)
    if isinstance(W32, tuple):
        y = torch.empty((x.shape[0], W.shape[0]), dtype=out_dtype, device=x.device)
        _gemm_split_rows(x, _mm_in(x), W, W32, b, y, y16)
        return _attach_b16(y, y16)
    if W32 is not None:
        y, _ = K.gemm(x, W32, bias=b, A2=x2, act_epi=act_epi, residual=residual, out_dtype=out_dtype, out16=y16, force_x3=True)
        return _attach_b16(y, y16)
    y, _ = K.gemm(_mm_in(x), W, bias=b, A2=_mm_in(x2), act_pro=act_pro, act_epi=act_epi, residual=residual,
                  out_dtype=out_dtype, p_drop=p_drop, seed=seed, out16=y16)
    return _attach_b16(y, y16)


class _MLP2(torch.autograd.Function):
    """y = act([x | x2] W1^T + b1) W2^T + b2 (+ residual) as ONE autograd node (the node MLP of MC_E_GCL, egnn.py:99-109, and
    the Transition blocks, model_utils.py:162-175).  Compared with two chained `linear` nodes the backward
      * applies the activation adjoint inside the input-gradient GEMM of the second Linear (the epilogue multiplies by act'
        while it flushes its tile: no separate pass over the [M, hidden] gradient),
      * folds the residual branch's gradient into the epilogue of the first Linear's input-gradient GEMM (or deposits it in x's
        shared gradient buffer), so autograd adds nothing."""

    @staticmethod
    def forward(ctx, x, x2, W1, b1, W2, b2, residual, act, holder=None, p_drop=0.0, seed=0, W32=None, chain=None):
        # p_drop > 0 (round 5, train mode): dropout on the second Linear's output AHEAD of the residual (egnn.py:106) inside its fp32
        # epilogue; the adjoint regenerates the mask from (seed, row, col) in the one pass that casts dy
        assert act in (K.ACT_RELU, K.ACT_SILU)
        ctx.drop = (float(p_drop), int(seed)) if p_drop > 0.0 else None
        xin, x2in = _mm_in(x), _mm_in(x2)
        M, N1 = x.shape[0], W1.shape[0]
        ad = act_dtype()
        y16 = None
        split_ok = W32 is not None and ctx.drop is None and _split_site_ok(x, W1, W32[0], x2, K.ACT_NONE, torch.float32, 0.0, act, None)
        if split_ok and chain is not None and _chain_ok(x, x2, residual, W1, W2):
            # round 6: the whole forward of the MLP as ONE split-precision kernel (csrc/node_chain.hip): the fp32 hidden layer never leaves the
            # CU; what the (unchanged, bf16) backward needs -- bf16(hidden) and, for SiLU, bf16 of its derivative -- is written on the way
            y, y16, t, D = K.node_chain_x3_fwd(x, x2, chain, b1.float(), b2.float(), act, 0 if x2 is not None else 1, residual=residual,
                                               want16=holder is not None, save=True)
            if holder is not None:
                holder.append(y16)
            ctx.act, ctx.has_x2, ctx.has_res = act, x2 is not None, residual is not None
            ctx.res_is_x = residual is x
            ctx.sink_x, ctx.sink_res = _sink_of(x), _sink_of(residual)
            ctx.x2_dtype = x2.dtype if x2 is not None else None
            ctx.save_for_backward(xin, x2in, W1, W2, t, D)
            ctx.W1t, ctx.W2t = getattr(W1, "_fab_T", None), getattr(W2, "_fab_T", None)
            return y
        if holder is not None:
            y16 = torch.empty((M, W2.shape[0]), dtype=torch.bfloat16, device=x.device)
            holder.append(y16)
        if split_ok:
            # round 6, config.set_split_sites(3): BOTH contractions of the forward in split precision -- the hidden layer stays fp32 between them
            # (what the bf16 mode's remaining gap at n_iter 8 sat in: profiles/r05_precision_sites.txt).  Saved for the backward, which is
            # unchanged: the bf16 roundings of x, of the hidden activation and of its stored derivative.
            # (the first launch's epilogue writes the fp32 hidden tile, its bf16 copy and the bf16 derivative: FabindGemmArgs.c2_bf16)
            t32 = torch.empty((M, N1), dtype=torch.float32, device=x.device)
            t = torch.empty((M, N1), dtype=ad, device=x.device)
            D = torch.empty((M, N1), dtype=ad, device=x.device) if act == K.ACT_SILU else None
            K.gemm(x, W32[0], bias=b1, A2=x2, act_epi=act, out=t32, out2=D, out16=t, force_x3=True)
            y, _ = K.gemm(t32, W32[1], bias=b2, residual=residual, out_dtype=torch.float32, out16=y16, force_x3=True)
            del t32
        else:
            t = torch.empty((M, N1), dtype=ad, device=x.device)
            D = torch.empty((M, N1), dtype=ad, device=x.device) if act == K.ACT_SILU else None
            K.gemm(xin, W1, bias=b1, A2=x2in, act_epi=act, out=t, out2=D)
            y, _ = K.gemm(t, W2, bias=b2, residual=residual, out_dtype=torch.float32, out16=y16, p_drop=p_drop, seed=seed)
        ctx.act, ctx.has_x2, ctx.has_res = act, x2 is not None, residual is not None
        ctx.res_is_x = residual is x
        ctx.sink_x, ctx.sink_res = _sink_of(x), _sink_of(residual)
        ctx.x2_dtype = x2.dtype if x2 is not None else None
        ctx.save_for_backward(xin, x2in, W1, W2, t, D)
        ctx.W1t, ctx.W2t = getattr(W1, "_fab_T", None), getattr(W2, "_fab_T", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xin, x2in, W1, W2, t, D = ctx.saved_tensors
        dy = dy.contiguous()
        md = mm_dtype()
        ni = ctx.needs_input_grad
        if ctx.drop is not None:
            dy16, db2 = _mul_dropmask_colsum(dy, ctx.drop[0], ctx.drop[1], md)   # d (W2 t + b2) = dy * keep / (1 - p): mask regenerated
        elif dy.dtype != md and ni[4] and _db_in_tn(dy):
            dy16, db2 = _mul_dact(dy, None, K.ACT_NONE, md), None           # cast only: the bias gradient rides with dW2's contraction
        elif dy.dtype != md:
            dy16, db2 = _mul_dact_colsum(dy, None, K.ACT_NONE, md)          # one pass: cast + bias gradient
        else:
            dy16, db2 = dy, None
        if ni[4] and db2 is None:
            dW2, db2 = _weight_grad(dy16, t, K.ACT_NONE, None, W2.dtype, want_db=True, W=W2)
        else:
            dW2 = _weight_grad(dy16, t, K.ACT_NONE, None, W2.dtype, W=W2) if ni[4] else None
            db2 = db2 if db2 is not None else K.colsum(dy16)
        aux, dact = (t, K.ACT_RELU) if ctx.act == K.ACT_RELU else (D, K.ACT_STORED_DERIV)
        dpre, _ = K.gemm(dy16, ctx.W2t if ctx.W2t is not None else W2.t().contiguous(), aux=aux, dact=dact, out_dtype=md)      # (dy W2) * act'(pre)
        if ni[2] and ni[3]:
            dW1, db1 = _weight_grad(dpre, xin, K.ACT_NONE, x2in, W1.dtype, want_db=True, W=W1)
        else:
            db1 = K.colsum(dpre) if ni[3] else None
            dW1 = _weight_grad(dpre, xin, K.ACT_NONE, x2in, W1.dtype, W=W1) if ni[2] else None
        W1t = ctx.W1t if ctx.W1t is not None else W1.t().contiguous()
        K1 = xin.shape[1]
        dres = dy if (ctx.has_res and ni[6]) else None
        dx = dx2 = None
        if ni[0]:
            if ctx.sink_x is not None:
                if dres is not None and ctx.sink_res is ctx.sink_x:
                    ctx.sink_x.deposit(dres)                              # folded into the GEMM below (or into the buffer)
                    dres = None
                dx = ctx.sink_x.gemm_into(dpre, W1t[:K1], xin)
            elif dres is not None and ctx.res_is_x:
                dx, _ = K.gemm(dpre, W1t[:K1], residual=dres, out_dtype=torch.float32)
                dres = None
            else:
                dx, _ = K.gemm(dpre, W1t[:K1], out_dtype=torch.float32)
        if dres is not None and ctx.sink_res is not None:
            dres = ctx.sink_res.deposit(dres)
        if ctx.has_x2 and ni[1]:
            dx2, _ = K.gemm(dpre, W1t[K1:], out_dtype=torch.float32 if ctx.x2_dtype == torch.float32 else ctx.x2_dtype)
        return dx, dx2, dW1, db1, dW2, (db2 if ni[5] else None), dres, None, None, None, None, None, None


MLP2_CHAIN = os.environ.get("FABIND_MLP2_CHAIN", "1") == "1"       # 0: the split-precision MLPs under autograd as two gemm_x3 launches (A/B)


def _chain_ok(x, x2, residual, W1, W2):
    H = W2.shape[0]
    ok = lambda t: t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == H and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    kind0 = x2 is not None
    return (MLP2_CHAIN and H in (128, 256, 512) and ok(x) and (x2 is None or ok(x2)) and (residual is None or ok(residual))
            and tuple(W1.shape) == ((H, 2 * H) if kind0 else (2 * H, H)) and tuple(W2.shape) == ((H, H) if kind0 else (H, 2 * H)))


def mlp2(x, W1, b1, act, W2, b2, residual=None, x2=None, want16=False, p_drop=0.0, W32=None, chain=None):
    """act([x | x2] W1^T + b1) W2^T + b2 (+ residual) -> fp32.  One autograd node under autograd (see _MLP2); without
    autograd two GEMMs with fused epilogues.  p_drop: dropout on the second Linear's output ahead of the residual, in its epilogue.
    W32 = (W1_32, W2_32) (bf16 mode, config.set_split_sites(3)): the fp32 masters -- both forward contractions then run in split precision
    on fp32 operands with an fp32 hidden layer; the backward is the bf16 one either way."""
    if W32 is not None and (W32[0] is None or W32[1] is None or p_drop != 0.0 or _cfg.get_precision() != "bf16"):
        W32 = None
    if _needs_grad(x, x2, W1, b1, W2, b2, residual):
        K.tn_hook(W1, b1, W2, b2)
        holder = [] if _want16(torch.float32, K.ACT_NONE, want16) else None
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0.0 else 0
        y = _MLP2.apply(x, x2, W1, b1, W2, b2, residual, act, holder, p_drop, seed, W32, chain if W32 is not None else None)
        return _attach_b16(y, holder[0] if holder else None)
    if W32 is not None:
        t = linear(x, W1, b1, act_epi=act, x2=x2, out_dtype=torch.float32, W32=W32[0])
        return linear(t, W2, b2, residual=residual, want16=want16, W32=W32[1])
    t = linear(x, W1, b1, act_epi=act, x2=x2, out_dtype=act_dtype())
    return linear(t, W2, b2, residual=residual, want16=want16, p_drop=p_drop)


class _MLP2Relu(torch.autograd.Function):
    """y = act2(drop(relu(x W1^T + b1)) W2^T + b2) [dropout on y when act2 = relu] (+ residual when act2 = none) as ONE autograd node:
    the two Linears of the FABind+ LN-MLPs (model_utils.py:10-74) with the ReLU + dropout pairs inside the GEMM epilogues.  What is
    saved are the two epilogue outputs themselves -- their zeros are the dropped and the inactive positions -- and the backward applies
    the hidden layer's adjoint ([t > 0] / (1 - p)) inside the input-gradient GEMM of the second Linear while it flushes its tile, so the
    [M, hidden] gradient is written once instead of written, re-read and re-written by a separate pass; both bias gradients ride along
    with the weight-gradient contractions (or with the one pass over dy that the output activation needs anyway)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, residual, last_act, out_dtype, p_drop, seed1, seed2):
        assert not (last_act and residual is not None), "relu + dropout on the output: the residual is added by the caller"
        thr = int(p_drop * 65536.0 + 0.5)
        ctx.drop_scale = 1.0 / (1.0 - thr / 65536.0)
        ad = act_dtype()
        M = x.shape[0]
        t = torch.empty((M, W1.shape[0]), dtype=ad, device=x.device)
        K.gemm(x, W1, bias=b1, act_epi=K.ACT_RELU, out=t, p_drop=p_drop, seed=seed1)
        y = torch.empty((M, W2.shape[0]), dtype=out_dtype, device=x.device)
        if last_act:
            K.gemm(t, W2, bias=b2, act_epi=K.ACT_RELU, out=y, p_drop=p_drop, seed=seed2)
        else:
            K.gemm(t, W2, bias=b2, residual=residual, out=y)
        ctx.last_act, ctx.has_res = last_act, residual is not None
        ctx.res_dtype = residual.dtype if residual is not None else None
        ctx.sink_res = _sink_of(residual)
        ctx.save_for_backward(x, W1, W2, t, y if last_act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1, W2, t, y = ctx.saved_tensors
        dy = dy.contiguous()
        md = mm_dtype()
        ni = ctx.needs_input_grad
        s = ctx.drop_scale
        db2 = None
        in_tn = ni[3] and ni[4] and _db_in_tn(dy)
        if ctx.last_act:
            dpre2, db2 = (_mul_dact(dy, y, K.ACT_RELU, md, s), None) if in_tn else _mul_dact_colsum(dy, y, K.ACT_RELU, md, s)
        elif dy.dtype != md:
            dpre2, db2 = (_mul_dact(dy, None, K.ACT_NONE, md), None) if in_tn else _mul_dact_colsum(dy, None, K.ACT_NONE, md)
        else:
            dpre2 = dy
        dW2 = None
        if ni[3] and db2 is None and ni[4]:
            dW2, db2 = _weight_grad(dpre2, t, K.ACT_NONE, None, W2.dtype, want_db=True, W=W2)
        elif ni[3]:
            dW2 = _weight_grad(dpre2, t, K.ACT_NONE, None, W2.dtype, W=W2)
        if ni[4] and db2 is None:
            db2 = K.colsum(dpre2)
        dpre1, _ = K.gemm(dpre2, _wt(W2), aux=t, dact=K.ACT_RELU, alpha=s, out_dtype=md)          # (dpre2 W2) * [t > 0] / (1 - p)
        dW1 = db1 = None
        if ni[1] and ni[2]:
            dW1, db1 = _weight_grad(dpre1, x, K.ACT_NONE, None, W1.dtype, want_db=True, W=W1)
        else:
            dW1 = _weight_grad(dpre1, x, K.ACT_NONE, None, W1.dtype, W=W1) if ni[1] else None
            db1 = K.colsum(dpre1) if ni[2] else None
        dx = K.gemm(dpre1, _wt(W1), out_dtype=x.dtype)[0] if ni[0] else None
        dres = None
        if ctx.has_res and ni[5]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)
            if ctx.sink_res is not None:
                dres = ctx.sink_res.deposit(dres)
        return dx, dW1, db1, dW2, (db2 if ni[4] else None), dres, None, None, None, None, None


def mlp2_relu(x, W1, b1, W2, b2, last_act, residual=None, out_dtype=torch.float32, p_drop=0.0):
    """One-node form of relu-Linear -> [dropout] -> Linear (-> relu -> [dropout] | + residual) under autograd; bf16 operands only
    (the caller falls back to chained `linear` nodes otherwise)."""
    seeds = torch.randint(0, 2 ** 31 - 1, (2,)).tolist() if p_drop > 0.0 else (0, 0)
    K.tn_hook(W1, b1, W2, b2)
    return _MLP2Relu.apply(x, W1, b1, W2, b2, residual, last_act, out_dtype, p_drop, int(seeds[0]), int(seeds[1]))


# ------------------------------------------------------------------------------------------------
# linear + row-dot:  s_part[m, t] = sum_{n in tile t} act_epi(act_pro(x) W^T + b)[m,n] * u[n]
# ------------------------------------------------------------------------------------------------
def _gemm_rowdot(x, W, b, u, act_pro, act_epi, store, p_drop=0.0, seed=0, fold=None, post=False):
    """GEMM with row-dot epilogue; store=True keeps the pre-activation matrix (training) -- or, with post=True, the activation's
    output after the epilogue dropout.  fold = (row_mu, row_rs, col_c): LayerNorm of the rows of x folded into the epilogue (inference)."""
    M, Kd = x.shape
    N = W.shape[0]
    nt = (N + K.GEMM_BN - 1) // K.GEMM_BN
    part = torch.empty((M, nt), dtype=torch.float32, device=x.device)
    z = torch.empty((M, N), dtype=act_dtype(), device=x.device) if store else None
    a = GemmArgs()
    a.A, a.W, a.C, a.bias, a.dotvec, a.dot_out = ptr(x), ptr(W), ptr(z), ptr(b), ptr(u), ptr(part)
    a.M, a.N, a.K, a.K1 = M, N, Kd, Kd
    a.lda, a.ldw, a.ldc, a.dot_ld = x.stride(0), W.stride(0), N, nt
    a.a_dtype, a.w_dtype, a.c_dtype = dt_code(x.dtype), dt_code(W.dtype), dt_code(z.dtype) if store else 0
    a.act_pro, a.act_epi, a.store_preact, a.alpha = act_pro, act_epi, (0 if post else 1), 1.0
    a.p_drop, a.drop_seed = float(p_drop), int(seed) & 0xFFFFFFFF
    if fold is not None:
        a.store_preact = 0
        a.row_mu, a.row_rs, a.col_c = ptr(fold[0]), ptr(fold[1]), ptr(fold[2])
    x3 = x.dtype == torch.float32 and W.dtype == torch.float32 and _cfg.get_precision() == "bf16x3"
    a.split3 = 1 if x3 else 0
    label = "%s <%s,%s%s> M=%d N=%d K=%d" % ("fabind_gemm_x3" if x3 else "fabind_gemm", str(x.dtype).replace("torch.", ""),
                                             str(W.dtype).replace("torch.", ""), ",x3" if x3 else "", M, N, Kd)
    K._profiled(label, 2.0 * M * N * Kd, lambda: check(load().fabind_gemm(ctypes.byref(a), stream()), "fabind_gemm(rowdot)"))
    return z, part


class _LinearRowdot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, u, act_pro, act_epi):
        z, part = _gemm_rowdot(x, W, b, u, act_pro, act_epi, store=True)
        ctx.act_pro, ctx.act_epi = act_pro, act_epi
        ctx.save_for_backward(x, W, u, z)
        return part

    @staticmethod
    def backward(ctx, dpart):
        x, W, u, z = ctx.saved_tensors
        dpart = dpart.contiguous()
        M, N = z.shape
        dz = torch.empty_like(z)
        nchunk = _nchunk(M)
        scratch = torch.empty((nchunk, N), dtype=torch.float32, device=z.device)
        du = torch.empty(N, dtype=torch.float32, device=z.device)
        check(load().fabind_rowdot_bwd(ptr(z), dt_code(z.dtype), ptr(dpart), dpart.shape[1], ptr(u), ctx.act_epi, M, N,
                                       ptr(dz), ptr(du), ptr(scratch), nchunk, stream()), "fabind_rowdot_bwd")
        assert ctx.act_pro == K.ACT_NONE
        Wt = W.t().contiguous()
        dx, _ = K.gemm(dz, Wt, out_dtype=x.dtype)
        dW, db = _weight_grad(dz, x, K.ACT_NONE, None, W.dtype, want_db=True, W=W)
        return dx, dW, db, du, None, None


class _LinearRowdotDrop(torch.autograd.Function):
    """Row-dot of drop(relu(x W^T + b)) with u under autograd: the epilogue stores the dropped ReLU output z (its zeros are the
    dropped and the inactive positions), so the adjoint is one pass -- dz = dpart u [z > 0] / (1 - p), du = sum_r dpart z."""

    @staticmethod
    def forward(ctx, x, W, b, u, p_drop, seed):
        z, part = _gemm_rowdot(x, W, b, u, K.ACT_NONE, K.ACT_RELU, store=True, p_drop=p_drop, seed=seed, post=True)
        thr = int(p_drop * 65536.0 + 0.5)
        ctx.scale = 1.0 / (1.0 - thr / 65536.0)
        ctx.save_for_backward(x, W, u, z)
        return part

    @staticmethod
    def backward(ctx, dpart):
        x, W, u, z = ctx.saved_tensors
        M, N = z.shape
        dps = (dpart * ctx.scale).contiguous()                 # [M, tiles]: the 1 / (1 - p) of the kept positions
        dz = torch.empty_like(z)
        nchunk = _nchunk(M)
        scratch = torch.empty((nchunk, N), dtype=torch.float32, device=z.device)
        du = torch.empty(N, dtype=torch.float32, device=z.device)
        check(load().fabind_rowdot_bwd(ptr(z), dt_code(z.dtype), ptr(dps), dps.shape[1], ptr(u), K.ACT_RELU, M, N,
                                       ptr(dz), ptr(du), ptr(scratch), nchunk, stream()), "fabind_rowdot_bwd")
        ni = ctx.needs_input_grad
        dx = K.gemm(dz, _wt(W), out_dtype=x.dtype)[0] if ni[0] else None
        dW = db = None
        if ni[1] and ni[2]:
            dW, db = _weight_grad(dz, x, K.ACT_NONE, None, W.dtype, want_db=True, W=W)
        elif ni[1]:
            dW = _weight_grad(dz, x, K.ACT_NONE, None, W.dtype, W=W)
        elif ni[2]:
            db = K.colsum(dz)
        # (dz was formed from the PRE-scaled dpart, so it carries 1 / (1 - p) once, as it must; du = sum_r dpart z wants the
        #  unscaled dpart: the kernel's du is divided by the same factor)
        return dx, dW, db, (du / ctx.scale if ni[3] else None), None, None


def linear_rowdot(x, W, b, u, act_pro=K.ACT_NONE, act_epi=K.ACT_NONE, p_drop=0.0, fold=None):
    """p_drop > 0 (no-grad paths only): dropout on act(x W^T + b) before the row-dot with u, inside the epilogue.
    fold = (row_mu, row_rs, col_c) (no-grad, bf16 x): LayerNorm of x's rows folded into the epilogue."""
    if _needs_grad(x, W, b, u):
        assert fold is None, "the LayerNorm fold has no autograd path"
        K.tn_hook(W, b)
        if p_drop > 0.0:
            assert act_pro == K.ACT_NONE and act_epi == K.ACT_RELU, "epilogue dropout under autograd: ReLU outputs only"
            assert b is not None, "linear_rowdot with epilogue dropout under autograd needs a bias"
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            return _LinearRowdotDrop.apply(x, W, b, u, p_drop, seed)
        return _LinearRowdot.apply(x, W, b, u, act_pro, act_epi)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0.0 else 0
    return _gemm_rowdot(x, W, b, u, act_pro, act_epi, store=False, p_drop=p_drop, seed=seed, fold=fold)[1]


# ------------------------------------------------------------------------------------------------
# edge tail of MC_E_GCL under autograd: (row-dot of coord_mlp, segment-sum of the messages) as ONE node, so
# that the two gradient contributions to the messages meet inside one GEMM epilogue
#   dM = dz Wc + dagg[row]      (residual gathered through `row`) instead of a gather kernel + an add pass
# ------------------------------------------------------------------------------------------------
class _EdgeTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Mm, Wc, bc, w3, rowptr, row, n_rows, act_epi):
        z, part = _gemm_rowdot(Mm, Wc, bc, w3, K.ACT_NONE, act_epi, store=True)
        agg = K.segment_sum(Mm, rowptr, n_rows)
        ctx.act_epi = act_epi
        ctx.save_for_backward(Mm, Wc, w3, z, row)
        return part, agg

    @staticmethod
    def backward(ctx, dpart, dagg):
        Mm, Wc, w3, z, row = ctx.saved_tensors
        M, N = z.shape
        dz = torch.empty_like(z)
        nchunk = _nchunk(M)
        scratch = torch.empty((nchunk, N), dtype=torch.float32, device=z.device)
        du = torch.empty(N, dtype=torch.float32, device=z.device)
        if dpart is None:
            dpart = torch.zeros((M, (N + K.GEMM_BN - 1) // K.GEMM_BN), dtype=torch.float32, device=z.device)
        dpart = dpart.contiguous()
        check(load().fabind_rowdot_bwd(ptr(z), dt_code(z.dtype), ptr(dpart), dpart.shape[1], ptr(w3), ctx.act_epi, M, N,
                                       ptr(dz), ptr(du), ptr(scratch), nchunk, stream()), "fabind_rowdot_bwd")
        Wt = Wc.t().contiguous()
        if dagg is not None:
            dM, _ = K.gemm(dz, Wt, residual=dagg.contiguous(), r_index=row, out_dtype=Mm.dtype)
        else:
            dM, _ = K.gemm(dz, Wt, out_dtype=Mm.dtype)
        dW, dbc = _weight_grad(dz, Mm, K.ACT_NONE, None, Wc.dtype, want_db=True, W=Wc)
        return dM, dW, dbc, du, None, None, None, None


def edge_tail(Mm, Wc, bc, w3, rowptr, row, n_rows, act_epi=K.ACT_SILU):
    """(s_part [E, nt], agg [n_rows, H]) = (row-dot(act(Mm Wc^T + bc), w3), segment_sum(Mm))"""
    if _needs_grad(Mm, Wc, bc, w3):
        K.tn_hook(Wc, bc)
        return _EdgeTail.apply(Mm, Wc, bc, w3, rowptr, row, n_rows, act_epi)
    part = _gemm_rowdot(Mm, Wc, bc, w3, K.ACT_NONE, act_epi, store=False)[1]
    return part, K.segment_sum(Mm, rowptr, n_rows)


EDGE_SAVE_FWD = os.environ.get("FABIND_EDGE_SAVE_FWD", "1") != "0"          # 0: the recompute backward (four contractions, nothing kept)
EDGE_SAVE_MIN_EDGES = int(os.environ.get("FABIND_EDGE_SAVE_MIN_EDGES", "0"))
# (hidden 128 -- the pocket model of the full IaBNet on whole proteins, 1.5 M edges: forward 280 -> 434 us, backward 1,086 -> 657 us per launch with
#  FABIND_EDGE_SAVE_MIN_H=128, but the full-model step is host-bound there: 1,234 / 1,301 / 1,351 against 1,308 / 1,313 / 1,304 complexes/s in
#  three interleaved pairs (profiles/r04_ab_same_box.txt) -- the default stays 256, which keeps 1.2 GB per pocket-model layer free)
EDGE_SAVE_MIN_H = int(os.environ.get("FABIND_EDGE_SAVE_MIN_H", "256"))
# (round 6, measured and NOT enabled: the saving form for large graphs at hidden 128 -- E >= EDGE_SAVE_BIG_EDGES edges -- reads 539 / 538 against
#  543 / 547 complexes/s on the config-3 headline and 1,371 / 1,342 against 1,326 / 1,335 on the production-shape model step, two interleaved
#  pairs each (profiles/r06_ab_same_box.txt): the default keeps it off)
EDGE_SAVE_BIG_EDGES = int(os.environ.get("FABIND_EDGE_SAVE_BIG_EDGES", "2000000000"))


class _FusedEdge(torch.autograd.Function):
    """Whole intra-graph edge pipeline of MC_E_GCL as one kernel each way (csrc/fused_edge.hip).  H >= 256 (EDGE_SAVE_MIN_H): the training forward leaves the messages and two bf16 per-edge tiles (silu'(pre2), pre3: 6 H bytes per edge
    until the backward has run) and the backward chains two contractions per edge (csrc/fused_edge_bwd4.hip); otherwise -- and with
    FABIND_EDGE_SAVE_FWD=0 -- the forward keeps no per-edge tensor and the backward recomputes them tile by tile in LDS."""

    @staticmethod
    def forward(ctx, AB16, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, seed, holder=None, frags=None):
        ctx.H, ctx.g, ctx.p_drop, ctx.seed, ctx.frags = H, g, p_drop, seed, frags
        W2p, Wcp = (frags[0], frags[1]) if frags is not None else (K.pack_frag(W2), K.pack_frag(Wc))
        E = g.row_ctx.shape[0]
        save = EDGE_SAVE_FWD and (H >= EDGE_SAVE_MIN_H or (H >= 128 and E >= EDGE_SAVE_BIG_EDGES)) and E >= max(1, EDGE_SAVE_MIN_EDGES) and AB16.dtype == torch.bfloat16
        out = K.gcl_edge_fused(AB16, H, g.row_ctx, g.col_ctx, rhohat, w_r, W2p, b2, Wcp, bc, w3,
                               AB16.shape[0], p_drop, seed, want16=holder is not None, rowptr=g.rp_ctx, save=save)
        ctx.n_saved = 3 if save else 0
        ctx.save_for_backward(AB16, rhohat, w_r, W2, b2, Wc, bc, w3, *(out[-1] if save else ()))
        if holder is not None:
            holder.append(out[2])                     # the aggregated messages as the node MLP's bf16 operand (same kernels)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, dagg, ds):
        AB16, rhohat, w_r, W2, b2, Wc, bc, w3 = ctx.saved_tensors[:8]
        saved = tuple(ctx.saved_tensors[8:]) if ctx.n_saved else None
        g = ctx.g
        colptr, perm = g.ctx_by_col()
        if dagg is None:
            dagg = torch.zeros((AB16.shape[0], ctx.H), dtype=torch.float32, device=AB16.device)
        if ds is None:
            ds = torch.zeros(g.row_ctx.shape[0], dtype=torch.float32, device=AB16.device)
        dAB, drh, dwr, dW2, db2, dWc, dbc, dw3 = K.gcl_edge_fused_bwd(
            AB16, ctx.H, g.row_ctx, g.col_ctx, rhohat, w_r, W2, b2, Wc, bc, w3, ds.reshape(-1).float(), dagg.float(),
            colptr, perm, ctx.p_drop, ctx.seed, dab_bf16=AB16.dtype == torch.bfloat16, w_dtype=W2.dtype, rowptr=g.rp_ctx, frags=ctx.frags,
            saved=saved)
        return (dAB.to(AB16.dtype), drh, dwr, dW2.to(W2.dtype), db2, dWc.to(Wc.dtype), dbc, dw3, None, None, None, None, None, None)


class _FusedEdgeX3(torch.autograd.Function):
    """The same pipeline in the split-bf16 mode: forward = csrc/fused_edge_fwd3.hip (fp32 AB rows, three bf16 MFMAs per product
    term: the quantities the 1e-4 A gate is stated on).  Backward, H >= EDGE_SAVE_MIN_H (round 5): the forward SAVES bf16(M),
    silu'(pre2) x keep and pre3 -- evaluated on its split-precision accumulators -- and the two-contraction kernel of the bf16 mode
    (csrc/fused_edge_bwd4.hip) runs on them and on a bf16 copy of AB: 3.2 instead of 5.2 ms per launch at the headline shape and no
    recompute on bf16-rounded operands; otherwise the bf16 recompute kernel on bf16 copies of AB and the weights.  Either way the
    gradients of this block carry bf16 operand rounding (like 'bf16' mode), everything upstream / downstream of it stays fp32."""

    @staticmethod
    def forward(ctx, AB, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, seed):
        ctx.H, ctx.g, ctx.p_drop, ctx.seed = H, g, p_drop, seed
        E = g.row_ctx.shape[0]
        save = EDGE_SAVE_FWD and X3_EDGE_SAVE_FWD and H >= EDGE_SAVE_MIN_H and E >= max(1, EDGE_SAVE_MIN_EDGES)
        out = K.gcl_edge_fused_x3(AB, H, g.row_ctx, g.col_ctx, rhohat, w_r, W2, b2, Wc, bc, w3, AB.shape[0], p_drop, seed,
                                  rowptr=g.rp_ctx, save=save)
        ctx.n_saved = 3 if save else 0
        ctx.save_for_backward(AB, rhohat, w_r, W2, b2, Wc, bc, w3, *(out[2] if save else ()))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, dagg, ds):
        AB, rhohat, w_r, W2, b2, Wc, bc, w3 = ctx.saved_tensors[:8]
        saved = tuple(ctx.saved_tensors[8:]) if ctx.n_saved else None
        g = ctx.g
        colptr, perm = g.ctx_by_col()
        if dagg is None:
            dagg = torch.zeros((AB.shape[0], ctx.H), dtype=torch.float32, device=AB.device)
        if ds is None:
            ds = torch.zeros(g.row_ctx.shape[0], dtype=torch.float32, device=AB.device)
        dAB, drh, dwr, dW2, db2, dWc, dbc, dw3 = K.gcl_edge_fused_bwd(
            AB.to(torch.bfloat16), ctx.H, g.row_ctx, g.col_ctx, rhohat, w_r, W2, b2, Wc, bc, w3, ds.reshape(-1).float(), dagg.float(),
            colptr, perm, ctx.p_drop, ctx.seed, dab_bf16=False, w_dtype=torch.float32, rowptr=g.rp_ctx, saved=saved)
        return (dAB, drh, dwr, dW2, db2, dWc, dbc, dw3, None, None, None, None)


X3_EDGE_SAVE_FWD = os.environ.get("FABIND_X3_EDGE_SAVE_FWD", "1") != "0"    # 0: 'bf16x3' keeps the recompute backward (round 3-4; A/B)


def fused_edge(AB16, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop=0.0, frags=None):
    """(agg [N,H], s [E,1]) of the fused edge pipeline (bf16, or split bf16 on fp32 AB in 'bf16x3' mode); differentiable.  frags
    (bf16 mode): (W2p, Wcp[, W2Tp, WcTp]) fragment packs made ahead by kernels.pack_frag_multi (all four when a backward will come).  p_drop > 0:
    dropout on the messages (egnn.py:82) from a counter-based mask keyed by a seed drawn from torch's CPU generator (no device sync)."""
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0.0 else 0
    if _needs_grad(AB16, rhohat, w_r, W2, b2, Wc, bc, w3):
        K.tn_hook(W2, Wc)                    # (their gradients may be queued: kernels.gcl_edge_fused_bwd)
    if _cfg.get_precision() == "bf16x3" and _cfg.x3_edge_bf16() and AB16.dtype == torch.float32:
        # the bf16 kernels on a bf16 copy of AB (config.set_x3_edge: 3.5e-6 A of ligand RMSD against 2.7 ms per launch); the cast is a
        # differentiable torch op, the weights are packed as bf16 fragments by the kernels' wrappers
        ab = AB16.to(torch.bfloat16)
        if _needs_grad(AB16, rhohat, w_r, W2, b2, Wc, bc, w3):
            agg, s = _FusedEdge.apply(ab, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, seed, None, None)
            return agg, s
        out = K.gcl_edge_fused(ab, H, g.row_ctx, g.col_ctx, rhohat, w_r, K.pack_frag(W2), b2, K.pack_frag(Wc), bc, w3, ab.shape[0], p_drop, seed,
                               want16=False, rowptr=g.rp_ctx)
        return out[0], out[1]
    if _cfg.get_precision() == "bf16x3":
        if _needs_grad(AB16, rhohat, w_r, W2, b2, Wc, bc, w3):
            return _FusedEdgeX3.apply(AB16, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, seed)
        return K.gcl_edge_fused_x3(AB16, H, g.row_ctx, g.col_ctx, rhohat, w_r, W2, b2, Wc, bc, w3, AB16.shape[0], p_drop, seed,
                                   rowptr=g.rp_ctx)
    w16 = _cfg.get_precision() == "bf16"
    if _needs_grad(AB16, rhohat, w_r, W2, b2, Wc, bc, w3):
        holder = [] if w16 else None
        agg, s = _FusedEdge.apply(AB16, rhohat, w_r, W2, b2, Wc, bc, w3, H, g, p_drop, seed, holder, frags)
        return _attach_b16(agg, holder[0] if holder else None), s
    W2p, Wcp = (frags[0], frags[1]) if frags is not None else (K.pack_frag(W2), K.pack_frag(Wc))
    out = K.gcl_edge_fused(AB16, H, g.row_ctx, g.col_ctx, rhohat, w_r, W2p, b2, Wcp, bc, w3,
                           AB16.shape[0], p_drop, seed, want16=w16, rowptr=g.rp_ctx)
    return (_attach_b16(out[0], out[2]), out[1]) if w16 else out


# ------------------------------------------------------------------------------------------------
# geometry / gathers / segmented reductions
# ------------------------------------------------------------------------------------------------
class _EdgeGeom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, row, col, rowptr, lay, bycol):
        d, rho, rhohat, norm = K.edge_geom(x, row, col, rowptr, lay.node_off, lay.B)
        ctx.lay, ctx.n, ctx.bycol = lay, x.shape[0], bycol
        ctx.save_for_backward(row, col, rowptr, d, rho, norm)
        return d, rhohat

    @staticmethod
    def backward(ctx, dd, drhohat):
        row, col, rowptr, d, rho, norm = ctx.saved_tensors
        lay, E, dev = ctx.lay, row.shape[0], d.device
        if ctx.bycol is not None:
            colptr, perm = ctx.bycol()
        else:                                     # edges grouped by sending node (index glue, no activations)
            colsorted, perm64 = torch.sort(col.long(), stable=True)
            colptr = torch.zeros(ctx.n + 1, dtype=torch.int32, device=dev)
            colptr[1:] = torch.cumsum(torch.bincount(colsorted, minlength=ctx.n), 0).to(torch.int32)
            perm = perm64.to(torch.int32)
        dx = torch.empty((ctx.n, 3), dtype=torch.float32, device=dev)
        dd = dd.contiguous() if dd is not None else torch.zeros_like(d)
        drhohat = drhohat.contiguous() if drhohat is not None else torch.zeros_like(rho)
        gbuf = torch.empty((max(E, 1), 3), dtype=torch.float32, device=dev)
        check(load().fabind_edge_geom_bwd(ptr(d), ptr(rho), ptr(norm), ptr(dd), ptr(drhohat), ptr(row), ptr(rowptr),
                                          ptr(colptr), ptr(perm), ptr(lay.node_off), lay.B, E, ctx.n, ptr(dx), ptr(gbuf),
                                          ptr(torch.empty(lay.B, dtype=torch.float32, device=dev)), stream()),
              "fabind_edge_geom_bwd")
        return dx, None, None, None, None, None


def edge_geom(x, row, col, rowptr, lay, bycol=None):
    """bycol: callable returning (colptr, perm) = the edges grouped by their sending node (Graph.ctx_by_col; for the
    symmetric inter graph (rowptr, mirror)); derived by a sort in backward when omitted."""
    if _needs_grad(x):
        return _EdgeGeom.apply(x, row, col, rowptr, lay, bycol)
    d, _, rhohat, _ = K.edge_geom(x, row, col, rowptr, lay.node_off, lay.B)
    return d, rhohat


class _GclPre(torch.autograd.Function):
    @staticmethod
    def forward(ctx, AB, rhohat, w_r, H, g, act):
        ctx.H, ctx.g, ctx.act = H, g, act
        if act == K.ACT_NONE:
            out, D = K.gcl_pre(AB, H, g.row_ctx, g.col_ctx, rhohat, w_r, act_dtype()), None
        else:
            out, D = K.gcl_pre(AB, H, g.row_ctx, g.col_ctx, rhohat, w_r, act_dtype(), act, want_dact=True)
        ctx.save_for_backward(AB, rhohat, w_r, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        AB, rhohat, w_r, D = ctx.saved_tensors
        H, g = ctx.H, ctx.g
        E, N = g.row_ctx.shape[0], AB.shape[0]
        dout = dout.contiguous()
        dpre = dout if D is None else _mul_dact(dout, D, K.ACT_STORED_DERIV, dout.dtype)
        dAB = torch.empty((N, 2 * H), dtype=torch.float32, device=AB.device)
        K.segment_sum(dpre, g.rp_ctx, N, out=dAB[:, :H])                      # receiving side: rows are CSR segments
        colptr, perm = g.ctx_by_col()
        K.segment_sum(dpre, colptr, N, eidx=perm, out=dAB[:, H:])             # sending side: permuted segments
        drh = torch.empty(E, dtype=torch.float32, device=AB.device)
        nchunk = max(1, min(2048, (E + 255) // 256))
        scratch = torch.empty((nchunk, H), dtype=torch.float32, device=AB.device)
        dw = torch.empty(H, dtype=torch.float32, device=AB.device)
        check(load().fabind_gcl_pre_bwd(ptr(dpre), dt_code(dpre.dtype), H, ptr(rhohat), ptr(w_r), E, ptr(drh), ptr(dw),
                                        ptr(scratch), nchunk, stream()), "fabind_gcl_pre_bwd")
        return dAB, drh, dw, None, None, None


def gcl_pre(AB, H, g, rhohat, w_r, act=K.ACT_NONE):
    if _needs_grad(AB, rhohat, w_r):
        return _GclPre.apply(AB, rhohat, w_r, H, g, act)
    return K.gcl_pre(AB, H, g.row_ctx, g.col_ctx, rhohat, w_r, act_dtype(), act)


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z, rowptr, row, n_rows, act):
        ctx.act = act
        ctx.save_for_backward(Z, row)
        return K.segment_sum(Z, rowptr, n_rows, act)

    @staticmethod
    def backward(ctx, dout):
        Z, row = ctx.saved_tensors
        dout = dout.contiguous()
        dZ = torch.empty_like(Z)
        check(load().fabind_gather_dact(ptr(dout), dout.stride(0), ptr(row), ptr(Z), dt_code(Z.dtype), ctx.act, ptr(dZ),
                                        dt_code(dZ.dtype), Z.shape[0], Z.shape[1], stream()), "fabind_gather_dact")
        return dZ, None, None, None, None


def segment_sum(Z, rowptr, row, n_rows, act=K.ACT_NONE):
    if _needs_grad(Z):
        return _SegmentSum.apply(Z, rowptr, row, n_rows, act)
    return K.segment_sum(Z, rowptr, n_rows, act)


class _CoordUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, d, s_part, rowptr, mean, clampv):
        x_out, s = K.coord_update(x, d, s_part, rowptr, mean, clampv, want_s=True)
        ctx.mean, ctx.clampv, ctx.np = mean, clampv, s_part.shape[1]
        ctx.save_for_backward(d, s, rowptr)
        return x_out

    @staticmethod
    def backward(ctx, dxo):
        d, s, rowptr = ctx.saved_tensors
        dxo = dxo.contiguous()
        E = s.shape[0]
        dd = torch.empty((E, 3), dtype=torch.float32, device=d.device)
        ds = torch.empty(E, dtype=torch.float32, device=d.device)
        check(load().fabind_coord_update_bwd(ptr(d), ptr(s), ptr(rowptr), dxo.shape[0], 1 if ctx.mean else 0, ctx.clampv,
                                             ptr(dxo), ptr(dd), ptr(ds), stream()), "fabind_coord_update_bwd")
        return dxo, dd, ds[:, None].expand(E, ctx.np), None, None, None


def coord_update(x, d, s_part, rowptr, mean, clampv):
    if _needs_grad(x, d, s_part):
        return _CoordUpdate.apply(x, d, s_part, rowptr, mean, clampv)
    return K.coord_update(x, d, s_part, rowptr, mean, clampv)[0]


class _TakeRows(torch.autograd.Function):
    """x[idx] whose adjoint adds the rows into x's shared gradient buffer (autograd's own index_select backward builds a
    zero [N, .] tensor and the engine then adds it to the other gradients of x)."""

    @staticmethod
    def forward(ctx, x, idx, sink):
        ctx.sink, ctx.n = sink, x.shape[0]
        ctx.save_for_backward(idx)
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, d_rows):
        idx, = ctx.saved_tensors
        return ctx.sink.deposit_rows(idx, d_rows), None, None


class _TakeUniqueRows(torch.autograd.Function):
    """x[idx] for idx WITHOUT REPEATS: the adjoint is a row copy into zeros (autograd's own index_select backward is an index_add_ -- bf16
    atomics: 1.2 ms for the 600 k x 512 pair rows of FABind+'s distance-map head, and the only float atomics of that training step)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.n = x.shape[0]
        ctx.save_for_backward(idx)
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, d_rows):
        idx, = ctx.saved_tensors
        g = torch.zeros((ctx.n,) + tuple(d_rows.shape[1:]), dtype=d_rows.dtype, device=d_rows.device)
        g.index_copy_(0, idx, d_rows)
        return g, None


class _TakeRowsFew(torch.autograd.Function):
    """x[idx] for a NARROW x (<= 8 columns, fp32: coordinates) gathered MANY times per row: autograd's adjoint is an index_put with
    accumulate -- contended fp32 atomics in arrival order (0.45 ms for 614,600 pair rows onto 2,560 ligand atoms in the distance head, and
    the gradient's last bits change from run to run).  Here: a stable sort of idx and one fixed-order segment sum."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.n, ctx.w = x.shape[0], x.shape[1]
        ctx.save_for_backward(idx)
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, d_rows):
        idx, = ctx.saved_tensors
        n, w = ctx.n, ctx.w
        srt, perm = torch.sort(idx, stable=True)
        ptr = torch.searchsorted(srt, torch.arange(n + 1, dtype=idx.dtype, device=idx.device)).to(torch.int32)
        g8 = torch.zeros((d_rows.shape[0], 8), dtype=torch.float32, device=d_rows.device)
        g8[:, :w] = d_rows
        out = K.segment_sum(g8, ptr, n, eidx=perm.to(torch.int32))
        return out[:, :w].to(d_rows.dtype), None


def take_rows_few(x, index64):
    """Row gather of a narrow fp32 tensor with a deterministic, atomics-free adjoint (see _TakeRowsFew)."""
    if _needs_grad(x) and x.dim() == 2 and x.shape[1] <= 8 and x.dtype == torch.float32 and index64.numel() > 0:
        return _TakeRowsFew.apply(x, index64)
    return x.index_select(0, index64)


class _DropMix(torch.autograd.Function):
    """h + dropout(hn - h) as one pass each way (csrc/bwd.hip: drop_mix_kernel); the mask is keyed by (seed, element) and regenerated in
    the adjoint."""

    @staticmethod
    def forward(ctx, h, hn, p_drop, seed):
        out = torch.empty_like(h)
        check(load().fabind_drop_mix(ptr(h), ptr(hn), ptr(out), h.numel(), float(p_drop), int(seed) & 0xFFFFFFFF, stream()), "fabind_drop_mix")
        ctx.pd, ctx.seed = float(p_drop), int(seed)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dhn, dh = torch.empty_like(g), torch.empty_like(g)
        check(load().fabind_drop_mix_bwd(ptr(g), ptr(dhn), ptr(dh), g.numel(), ctx.pd, ctx.seed & 0xFFFFFFFF, stream()), "fabind_drop_mix_bwd")
        return dh, dhn, None, None


def drop_mix(h, hn, p_drop):
    """h + nn.Dropout(p)(hn - h) (train mode; egnn.py:236).  fp32, contiguous, numel % 4 == 0: one kernel; else torch ops."""
    if p_drop <= 0.0:
        return hn
    if (h.dtype == torch.float32 and hn.dtype == torch.float32 and h.is_contiguous() and hn.is_contiguous() and h.shape == hn.shape
            and h.numel() % 4 == 0 and h.is_cuda):
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        if _needs_grad(h, hn):
            return _DropMix.apply(h, hn, p_drop, seed)
        out = torch.empty_like(h)
        check(load().fabind_drop_mix(ptr(h), ptr(hn), ptr(out), h.numel(), float(p_drop), seed, stream()), "fabind_drop_mix")
        return out
    return h + torch.nn.functional.dropout(hn - h, p_drop, True)


def sum_sorted_segments(x, counts):
    """Per-segment sums of the rows of x whose segments are CONTIGUOUS (a sorted batch vector) with `counts` rows each -- one thread
    walks a segment in order (torch.segment_reduce): no float atomics, the same bits every run.  (zeros(B, .).index_add_(0, batch, x),
    the form the reference's scatter_mean suggests, adds in arrival order: the stage-2 input coordinates then differ in their last bits
    from run to run, and with them everything downstream.)  Index glue on inputs: not differentiable.
    The lengths are validated (segment_reduce without `unsafe`: they must be non-negative and sum to the row count -- one small read-back on
    a path, stage 2, that reads sizes back anyway)."""
    return torch.segment_reduce(x.detach(), "sum", lengths=counts.to(torch.int64), axis=0)


def take_unique_rows(x, index64):
    """Row gather with an index free of repeats (the caller's contract)."""
    return _TakeUniqueRows.apply(x, index64) if _needs_grad(x) else x.index_select(0, index64)


def take_rows(x, index64):
    """Row gather (pure data movement).  With a shared gradient buffer on x the adjoint is a row scatter-add into it."""
    sink = _sink_of(x)
    if sink is not None and _needs_grad(x):
        return _TakeRows.apply(x, index64, sink)
    return x.index_select(0, index64)


class _PutRows(torch.autograd.Function):
    """base[idx] = rows, in place on `base` (a fresh intermediate); gradients: d rows = dout[idx], d base = dout with
    those rows zeroed."""

    @staticmethod
    def forward(ctx, base, rows, idx, inplace):
        ctx.save_for_backward(idx)
        if not inplace:                      # `base` is saved by its producer (e.g. a ReLU epilogue): leave it untouched
            out = base.clone()
            out.index_copy_(0, idx, rows.to(base.dtype))
            return out
        _put_rows_inplace(base, rows, idx)
        ctx.mark_dirty(base)
        return base

    @staticmethod
    def backward(ctx, dout):
        idx, = ctx.saved_tensors
        d_rows = dout.index_select(0, idx)
        # d base = dout with the overwritten rows zeroed.  In place ONLY when this node provably owns the tensor: it carries the
        # `_fab_owned` mark of `_SinkOwner.backward` (the shared gradient buffer this library allocated and handed to autograd
        # itself, or a tensor created there) and nobody but the engine's frame and this Python wrapper references it -- a
        # producer that hands one gradient tensor to several edges (AddBackward on the pdrop > 0 paths) shows a higher use
        # count, and anything autograd summed or copied on the way has lost the mark.  Everything else gets a fresh tensor.
        if PUT_ROWS_INPLACE_GRAD and dout.is_contiguous() and _owned_grad(dout):
            d_base = dout
            d_base.index_fill_(0, idx, 0.0)
        else:
            d_base = dout.index_fill(0, idx, 0.0)
        return d_base, d_rows, None, None


PUT_ROWS_INPLACE_GRAD = True


def _owned_grad(g):
    """True if the incoming gradient `g` is a buffer this library created for exactly this edge of the graph (marked by
    `_SinkOwner.backward`) and no other holder exists (TensorImpl use count: the engine's input list + this Python wrapper)."""
    return bool(getattr(g, "_fab_owned", False)) and g._use_count() <= 2


def _put_rows_inplace(base, rows, idx):
    """base[idx] = rows in place; a remembered bf16 operand copy of base gets the same rows (and stays valid)."""
    c = getattr(base, "_fab_b16", None)
    keep = c is not None and c[0] == base._version
    base.index_copy_(0, idx, rows.to(base.dtype))
    if keep:
        c[1].index_copy_(0, idx, rows.to(torch.bfloat16))
        base._fab_b16 = (base._version, c[1])


def put_rows(base, rows, index64, inplace=True):
    """Row scatter of a small compact array into a node-layout array (index glue: pure data movement)."""
    if _needs_grad(base, rows):
        c = getattr(base, "_fab_b16", None)
        keep = c is not None and c[0] == base._version
        out = _PutRows.apply(base, rows, index64, inplace)
        if keep and inplace:                   # (the Function patched the copy's rows; autograd bumped the version once more)
            out._fab_b16 = (out._version, c[1])
        return out
    _put_rows_inplace(base, rows, index64)
    return base


def select_rows(x, z, mask_u8):
    if _needs_grad(x, z):
        return torch.where(mask_u8.bool()[:, None], z, x)
    return K.select_rows(x, z, mask_u8)


# ------------------------------------------------------------------------------------------------
# attention ops
# ------------------------------------------------------------------------------------------------
class GradSink:
    """Shared gradient buffer of a tensor with several consumers: autograd neither pads nor sums [N, .] tensors for it.

    Three kinds of contribution (all consumers run on one stream before the producer's backward reads the buffer):
    * accumulating adjoint kernels (atomics / accumulating GEMM epilogues): `take()` -- the first one allocates the zeroed
      buffer and hands it to autograd as ITS gradient, the others add in place and return None;
    * an input-gradient GEMM: `gemm_into()` -- the first writer stores (folding deferred contributions in through the
      epilogue's residual operand), later ones run with accumulate = 1;
    * a gradient that already exists as a tensor (the residual branch y = f(x) + x): `deposit()` -- held without any
      kernel until the next GEMM folds it in.
    `_SinkOwner` (an identity node between producer and consumers) adds whatever is still deferred when all consumers
    have run, and drops the references, so a second backward over a retained graph starts fresh."""

    def __init__(self):
        self.buf = None
        self.pending = None          # full-size fp32 tensor deferred by deposit()
        self.rows = []               # (index64, rows) deferred by deposit_rows()
        self.deferred = []           # (A, Wt) input-gradient GEMMs held back by gemm_into() until `_SinkOwner.backward` (DEFER_DX)

    def take(self, like):
        """-> (buffer, True if this call created it and must return it as the gradient)."""
        if self.buf is None:
            self.buf = torch.zeros_like(like)
            self._flush_into_buf()
            return self.buf, True
        return self.buf, False

    def _flush_into_buf(self):
        if self.pending is not None:
            self.buf.add_(self.pending)
            self.pending = None
        for idx, r in self.rows:
            self.buf.index_add_(0, idx, r.to(self.buf.dtype))
        self.rows = []

    def deposit(self, g):
        """Residual-branch gradient g (same shape as the tensor).  Returns None: the contribution reaches autograd through
        the buffer (or through `_SinkOwner` if no buffer is ever made)."""
        if self.buf is not None:
            self.buf.add_(g)
        elif self.pending is None:
            self.pending = g
        else:
            self.pending = self.pending + g
        return None

    def deposit_rows(self, idx, rows):
        """Gradient of a row gather x[idx]: rows added at idx."""
        if self.buf is not None:
            self.buf.index_add_(0, idx, rows.to(self.buf.dtype))
        else:
            self.rows.append((idx, rows))
        return None

    def gemm_into(self, A, Wt, like, dtype=torch.float32):
        """d x (+)= A @ Wt^T-form GEMM (K.gemm(A, Wt)) into the shared buffer; returns the buffer if this call created it
        (the caller hands it to autograd), else None."""
        if (DEFER_DX and A.dtype == torch.bfloat16 and Wt.dtype == torch.bfloat16 and A.dim() == 2 and A.stride(1) == 1 and A.stride(0) % 8 == 0
                and A.shape[1] % 64 == 0 and A.shape[0] >= DEFER_DX_MIN_ROWS and A.data_ptr() % 16 == 0):
            # held back: a tensor with two input-gradient GEMMs (first edge Linear + node MLP on h; k | v projection + Transition on the
            # attention update; pair projections + q | k | v on the attention layer's input) gets ONE launch over the K-concatenated
            # operands [A1 | A2] x [Wt1 | Wt2] when its last consumer has reported -- one pass over the fp32 gradient instead of a
            # store + a read-modify-write (8 of the 12 bytes per element of the second launch), and the small-K launch disappears
            self.deferred.append((A, Wt))
            return None
        if self.buf is None:
            out = torch.empty(like.shape, dtype=dtype, device=like.device)      # (`like` may be the bf16 operand copy of an fp32 tensor)
            pend = self.pending
            if pend is not None and pend.dtype != out.dtype and not (pend.dtype == torch.bfloat16 and out.dtype == torch.float32):
                pend = pend.to(out.dtype)
            K.gemm(A, Wt, out=out, residual=pend)
            self.pending = None
            self.buf = out
            rows, self.rows = self.rows, []
            for idx, r in rows:
                out.index_add_(0, idx, r.to(out.dtype))
            return out
        K.gemm(A, Wt, out=self.buf, accumulate=True)
        return None


DEFER_DX = os.environ.get("FABIND_DEFER_DX", "1") == "1"     # 0: every input-gradient GEMM into a shared buffer is launched where it arises (round 4)
DEFER_DX_MIN_ROWS = int(os.environ.get("FABIND_DEFER_DX_MIN_ROWS", "1024"))


def _run_deferred(defs, out, base, accumulate):
    """The held-back input-gradient GEMMs of one tensor: pairs as one K-concatenated launch each ([A1 | A2] x [Wt1 | Wt2]^T; the weight
    operand is concatenated here: K_in x (N1 + N2) bf16, a parameter-sized copy).  The first launch stores `out` (+ base, the gradient
    that already exists as a tensor, through the epilogue's residual operand) unless `accumulate`; the others accumulate."""
    defs = sorted(defs, key=lambda d_: -d_[0].shape[1])          # largest K first: it becomes segment one of the first launch
    first = True
    while defs:
        if len(defs) >= 2:
            (A1, W1), (A2, W2) = defs[0], defs[1]
            defs = defs[2:]
            Wcat = torch.cat([W1, W2], 1)
            args = dict(A2=A2)
        else:
            (A1, Wcat), = defs
            defs, args = [], {}
        if first and not accumulate:
            K.gemm(A1, Wcat, out=out, residual=base, **args)
        else:
            K.gemm(A1, Wcat, out=out, accumulate=True, **args)
        first = False


class _SinkOwner(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, sink):
        ctx.sink = sink
        ctx.meta = (t.shape, t.dtype, t.device)
        ctx.set_materialize_grads(False)
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        sink = ctx.sink
        pend, rows, defs = sink.pending, sink.rows, sink.deferred
        owned = g is not None and g is sink.buf    # the buffer this sink allocated and returned to autograd as the one gradient
        sink.buf, sink.pending, sink.rows, sink.deferred = None, None, [], []
        if defs:                                   # the input-gradient GEMMs held back by gemm_into(): K-concatenated, one pass
            if owned:
                _run_deferred(defs, g, None, True)
            else:
                base = g
                if pend is not None:
                    base, pend = (pend if base is None else base + pend), None
                if base is not None and base.dtype != ctx.meta[1] and not (base.dtype == torch.bfloat16 and ctx.meta[1] == torch.float32):
                    base = base.to(ctx.meta[1])
                g = torch.empty(ctx.meta[0], dtype=ctx.meta[1], device=ctx.meta[2])
                _run_deferred(defs, g, base, False)
                owned = True
        if pend is not None:                       # a deposit that no later GEMM folded in
            if g is None:
                g, owned = pend, False             # `pend` may alias another node's incoming gradient (deposit() does not copy)
            else:
                g, owned = g + pend, True
        if rows:
            if g is None:
                g = torch.zeros(ctx.meta[0], dtype=ctx.meta[1], device=ctx.meta[2])
            elif not owned:
                g = g.clone()                      # never index_add_ into a tensor this node does not own
            owned = True
            for idx, r in rows:
                g.index_add_(0, idx, r.to(g.dtype))
        if owned:
            g._fab_owned = True                    # consumers upstream (_PutRows.backward) may reuse it in place
        return g, None


def shared_grad(t):
    """Mark t (about to be consumed several times by sink-aware ops) as having a shared gradient buffer."""
    if not (torch.is_grad_enabled() and t.requires_grad):
        return t
    sink = GradSink()
    out = _SinkOwner.apply(t, sink)
    out._fab_gsink = sink
    c = getattr(t, "_fab_b16", None)
    if c is not None and c[0] == t._version:       # the bf16 operand copy travels with the view
        out._fab_b16 = (out._version, c[1])
    return out


def _sink_of(t):
    return getattr(t, "_fab_gsink", None) if t is not None else None


def _sink_zeros(t, sink):
    """Gradient buffer of t for an accumulating adjoint: (buffer, what to return to autograd)."""
    if sink is None:
        buf = torch.zeros_like(t)
        return buf, buf
    buf, first = sink.take(t)
    return buf, (buf if first else None)


class _CrossAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qg, kv, bias, lin_col, gate_col, desc, B, max_nq, max_nk, scale):
        out = torch.zeros((qg.shape[0], 128), dtype=torch.float32, device=qg.device)   # rows outside every block: 0
        _, lse = K.cross_attn_fwd(qg[:, :128], kv[:, :128], kv[:, 128:], qg[:, 128:], bias, lin_col, gate_col, desc, B,
                                  max_nq, scale, out, want_lse=True, max_nk=max_nk)
        ctx.args = (lin_col, gate_col, desc, B, max_nq, max_nk, scale)
        ctx.mfma = K.cross_attn_use_mfma()        # the backward recomputes the probabilities the way the forward computed them
        ctx.covers_all = bool(getattr(desc, "_fab_covers_all", False))      # set by engine.Layout: the blocks tile the pair list
        ctx.save_for_backward(qg, kv, bias, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qg, kv, bias, out, lse = ctx.saved_tensors
        lin_col, gate_col, desc, B, max_nq, max_nk, scale = ctx.args
        dout = dout.contiguous()
        dqg, dkv = torch.zeros_like(qg), torch.zeros_like(kv)                                  # uncovered rows: 0
        # pass Q writes both 16-B segments of every pair the descriptors cover; with an 8-column bias tensor that is all of it
        dbias = torch.empty_like(bias) if (bias.shape[1] == 8 and lin_col == 0 and gate_col == 4 and ctx.covers_all) else torch.zeros_like(bias)
        dO = torch.empty((qg.shape[0], 128), dtype=torch.float32, device=qg.device)
        Dv = torch.empty((qg.shape[0], 4), dtype=torch.float32, device=qg.device)
        n_scr = int(load().fabind_cross_attn_bwd_scratch(B, max_nq, max_nk))     # per-split partials of the short side (or 0)
        scr = torch.empty(n_scr, dtype=torch.float32, device=qg.device) if n_scr else None
        fn = load().fabind_cross_attn_mfma_bwd if ctx.mfma else load().fabind_cross_attn_bwd
        check(fn(ptr(qg), qg.stride(0), ptr(kv), kv.stride(0), ptr(bias), bias.stride(0), lin_col, gate_col, ptr(desc), B, max_nq,
                 max_nk, scale, ptr(out), ptr(lse), ptr(dout), ptr(dqg), ptr(dkv), ptr(dbias), ptr(dO), ptr(Dv), ptr(scr), stream()),
              "fabind_cross_attn_bwd")
        return dqg, dkv, dbias, None, None, None, None, None, None, None


def cross_attn(qg, kv, bias, lin_col, gate_col, desc, B, max_nq, max_nk, scale):
    if _needs_grad(qg, kv, bias):
        return _CrossAttn.apply(qg, kv, bias, lin_col, gate_col, desc, B, max_nq, max_nk, scale)
    out = torch.zeros((qg.shape[0], 128), dtype=torch.float32, device=qg.device)
    K.cross_attn_fwd(qg[:, :128], kv[:, :128], kv[:, 128:], qg[:, 128:], bias, lin_col, gate_col, desc, B, max_nq, scale, out,
                     max_nk=max_nk)
    return out


class _PairHadamard(torch.autograd.Function):
    """hd[e, :H] = T0[p, :H] * T0[c, H:2H];  hd[e, H:H+H2] = T1[p, :H2] * T1[c, H2:2*H2]"""

    @staticmethod
    def forward(ctx, T0, T1, H, H2, red_p, red_c, graph):
        ctx.dims = (H, H2)
        ctx.sink = getattr(T0, "_fab_gsink", None)        # shared gradient buffer of T0 (shared_grad), or None
        ctx.graph = graph
        ctx.save_for_backward(T0, T1, red_p, red_c)
        return K.pair_hadamard(T0[:, :H], T0[:, H:], T1[:, :H2], T1[:, H2:], red_p, red_c, act_dtype())

    @staticmethod
    def backward(ctx, dhd):
        T0, T1, red_p, red_c = ctx.saved_tensors
        H, H2 = ctx.dims
        dhd = dhd.contiguous()
        d0, d0_ret = _sink_zeros(T0, ctx.sink)           # the kernel adds into d0 / d1
        d1 = torch.zeros_like(T1)
        g = ctx.graph
        if (PAIRHAD_ROWS and g is not None and T0.dtype == torch.float32 and T1.dtype == torch.float32 and H % 4 == 0 and H2 % 4 == 0
                and H <= 1024 and H2 <= 256
                and dhd.stride(0) % 4 == 0):
            # the pairs are the inter graph's own rows: one wave per node, one writer per element, fixed order (no float atomics)
            check(load().fabind_pair_hadamard_bwd_rows(
                ptr(dhd), dt_code(dhd.dtype), dhd.stride(0), ptr(T0), T0.stride(0), H, ptr(T1), T1.stride(0), H2, ptr(g.rp_int),
                ptr(g.col_int), ptr(g.red_idx), ptr(red_c), T0.shape[0], ptr(d0), d0.stride(0), ptr(d1), d1.stride(0), stream()),
                "fabind_pair_hadamard_bwd_rows")
            return d0_ret, d1, None, None, None, None, None
        check(load().fabind_pair_hadamard_bwd(
            ptr(dhd), dt_code(dhd.dtype), dhd.stride(0), ptr(T0[:, :H]), ptr(T0[:, H:]), T0.stride(0), H, ptr(T1[:, :H2]),
            ptr(T1[:, H2:]), T1.stride(0), H2, ptr(red_p), ptr(red_c), red_p.shape[0], ptr(d0[:, :H]), ptr(d0[:, H:]),
            d0.stride(0), ptr(d1[:, :H2]), ptr(d1[:, H2:]), d1.stride(0), stream()), "fabind_pair_hadamard_bwd")
        return d0_ret, d1, None, None, None, None, None


PAIRHAD_ROWS = os.environ.get("FABIND_PAIRHAD_ROWS", "1") == "1"   # adjoint of the pair Hadamard as a row walk of the inter graph (no atomics)


def pair_hadamard(a0b0, H, ab32, H2, red_p, red_c, graph=None):
    """graph: the inter graph whose reduced pairs (red_p, red_c) are (rp_int / col_int / red_idx) -- enables the atomic-free adjoint."""
    if _needs_grad(a0b0, ab32):
        return _PairHadamard.apply(a0b0, ab32, H, H2, red_p, red_c, graph)
    return K.pair_hadamard(a0b0[:, :H], a0b0[:, H:], ab32[:, :H2], ab32[:, H2:], red_p, red_c, act_dtype())


class _RowsHadamard(torch.autograd.Function):
    """out[e,:] = t[ia[e],:] * t[ib[e],:]"""

    @staticmethod
    def forward(ctx, t, ia, ib, a_sorted=False, blocks=None, n_a=0):
        ctx.save_for_backward(t, ia, ib)
        ctx.a_sorted, ctx.blocks, ctx.n_a = a_sorted, blocks, n_a
        fwd = _block_hadamard_fwd(t, blocks, n_a)
        if fwd is not None:
            return fwd
        dummy = t[:, :0]
        return K.pair_hadamard(t, t, dummy, dummy, ia, ib, act_dtype())

    @staticmethod
    def backward(ctx, dout):
        t, ia, ib = ctx.saved_tensors
        dout = dout.contiguous()
        W = t.shape[1]
        bl = ctx.blocks
        if bl is not None and dout.dtype == torch.bfloat16 and t.dtype == torch.float32 and W in (64, 128, 256, 512) and t.stride(0) % 2 == 0 \
                and ctx.n_a == bl.n_prows and t.shape[0] == bl.n_prows + bl.n_crows and dout.shape[0] == bl.n_pairs:
            # round 6: the pairs are dense per-complex blocks -> one pass over dout (read once), d t rows written once each, no per-call CSR
            dt_ = torch.empty_like(t)
            part = torch.empty((bl.n_tiles, bl.nchunk_max, bl.chunk, W), dtype=torch.float32, device=t.device)
            tc, dtc = t[ctx.n_a:], dt_[ctx.n_a:]
            check(load().fabind_block_hadamard_bwd(bl.desc_ptr, bl.B, bl.n_tiles, ptr(dout), dout.stride(0), ptr(t), t.stride(0), ptr(tc),
                                                   t.stride(0), W, bl.row_b_ptr, bl.n_crows, bl.nchunk_max, ptr(part), ptr(dt_),
                                                   dt_.stride(0), ptr(dtc), dt_.stride(0), stream()), "fabind_block_hadamard_bwd")
            return dt_, None, None, None, None, None
        if ROWS_HADAMARD_WALK and t.dtype == torch.float32 and W % 4 == 0 and W <= 1024 and t.stride(0) % 4 == 0 and dout.stride(0) % 4 == 0:
            # every row of t gets its pairs as a CSR: as first factor (the pairs sorted by ia) and as second factor (sorted by ib);
            # one wave per row sums them -- no float atomics, fixed order (index glue: two stable sorts and two histograms per call)
            n = t.shape[0]
            rowptr, pair_idx, partner = _rows_hadamard_csr(ia, ib, n, ctx.a_sorted)
            dt_ = torch.empty_like(t)
            check(load().fabind_rows_hadamard_bwd(ptr(dout), dt_code(dout.dtype), dout.stride(0), ptr(t), t.stride(0), W,
                                                  ptr(rowptr), ptr(pair_idx), ptr(partner), n, ptr(dt_), dt_.stride(0),
                                                  stream()), "fabind_rows_hadamard_bwd")
            return dt_, None, None, None, None, None
        dt_ = torch.zeros_like(t)
        check(load().fabind_pair_hadamard_bwd(ptr(dout), dt_code(dout.dtype), dout.stride(0), ptr(t), ptr(t), t.stride(0), W,
                                              None, None, 0, 0, ptr(ia), ptr(ib), ia.shape[0], ptr(dt_), ptr(dt_),
                                              dt_.stride(0), None, None, 0, stream()), "fabind_pair_hadamard_bwd")
        return dt_, None, None, None, None, None


def _rows_hadamard_csr(ia, ib, n, a_sorted=False):
    """CSR of the pairs of every row of t for the adjoint of out[e] = t[ia[e]] * t[ib[e]]: row r lists first its pairs as first factor
    (partner = ib of the pair), then its pairs as second factor (partner = ia), each block in pair order.
    -> (rowptr int32 [n + 1], pair_idx int32 [2 P], partner int32 [2 P])."""
    P = ia.shape[0]
    dev = ia.device
    ia64, ib64 = ia.long(), ib.long()
    pa = torch.arange(P, device=dev) if a_sorted else torch.argsort(ia64, stable=True)     # (a_sorted: ia is non-decreasing)
    pb = torch.argsort(ib64, stable=True)
    one = torch.ones_like(ia64)
    ca = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, ia64, one)      # (no torch.bincount: it reads its maximum back)
    cb_ = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, ib64, one)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(ca + cb_, 0)
    starts = rowptr[:-1]
    sa, sb = ia64[pa], ib64[pb]
    ar = torch.arange(P, device=dev)
    ra = ar - (torch.cumsum(ca, 0) - ca)[sa]                          # rank inside the row's first-factor block
    rb = ar - (torch.cumsum(cb_, 0) - cb_)[sb]
    pair_idx = torch.empty(2 * P, dtype=torch.int32, device=dev)
    partner = torch.empty(2 * P, dtype=torch.int32, device=dev)
    posa, posb = starts[sa] + ra, starts[sb] + ca[sb] + rb
    pair_idx[posa], partner[posa] = pa.to(torch.int32), ib64[pa].to(torch.int32)
    pair_idx[posb], partner[posb] = pb.to(torch.int32), ia64[pb].to(torch.int32)
    return rowptr.to(torch.int32), pair_idx, partner


ROWS_HADAMARD_WALK = os.environ.get("FABIND_ROWS_HADAMARD_WALK", "1") == "1"      # adjoint of rows_hadamard as a row walk (0: float atomics)


def _block_hadamard_fwd(t, bl, n_a):
    """The forward on the block descriptors (bf16 result, W in {64 .. 512}); None when the shapes do not fit."""
    W = t.shape[1]
    if not (bl is not None and act_dtype() == torch.bfloat16 and t.dtype == torch.float32 and W in (64, 128, 256, 512) and t.stride(1) == 1
            and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and n_a == bl.n_prows and t.shape[0] == bl.n_prows + bl.n_crows and n_a % 4 == 0):
        return None
    out = torch.empty((bl.n_pairs, W), dtype=torch.bfloat16, device=t.device)
    check(load().fabind_block_hadamard_fwd(bl.desc_ptr, bl.B, bl.n_tiles, ptr(t), t.stride(0), ptr(t[n_a:]), t.stride(0), W, ptr(out), W,
                                           stream()), "fabind_block_hadamard_fwd")
    return out


def rows_hadamard(t, idx_a, idx_b, a_sorted=False, blocks=None, n_a=0):
    """einsum('bik,bjk->bijk') restricted to the valid pairs (reference model.py:355).  a_sorted: idx_a is non-decreasing (a
    protein-major pair list) -- the adjoint's index glue skips one sort.  blocks (PairBlocks) + n_a: the same pairs as dense per-complex
    blocks over t = [n_a pocket rows | ligand rows]: the adjoint then needs no index glue at all (csrc/heads.hip)."""
    ia, ib = idx_a.to(torch.int32).contiguous(), idx_b.to(torch.int32).contiguous()
    if _needs_grad(t):
        return _RowsHadamard.apply(t, ia, ib, a_sorted, blocks, n_a)
    fwd = _block_hadamard_fwd(t, blocks, n_a)
    if fwd is not None:
        return fwd
    dummy = t[:, :0]
    return K.pair_hadamard(t, t, dummy, dummy, ia, ib, act_dtype())


class _InterAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cv, h, x, d, rhohat, bias_part, w_rk, w_rv, wcr, w3, H, g, clampv, s_ext=None, Wc=None, bc=None, holder=None, Wc32=None):
        v_in = None
        if Wc is not None:
            # cv = Linear(V) (egnn.py:225) evaluated inside this node of the autograd graph: its input gradient then accumulates
            # into the V columns of dqkv in place, instead of autograd padding it to [N,3H] and adding two [N,3H] tensors
            c16 = getattr(qkv, "_fab_b16", None) if _cfg.get_precision() == "bf16" else None    # (bf16x3 attaches copies for TN operands only)
            v_in = c16[1][:, 2 * H:] if (c16 is not None and c16[0] == qkv._version) else _mm_in(qkv[:, 2 * H:])
            if Wc32 is not None:     # split-precision site (config.set_split_sites): fp32 v rows x the fp32 master weight; v_in / Wc serve the backward
                cv, _ = K.gemm(qkv[:, 2 * H:], Wc32, bias=bc, force_x3=True)
            else:
                cv, _ = K.gemm(v_in, Wc, bias=bc)
        h16 = None
        if holder is not None:
            h16 = torch.empty((h.shape[0], H), dtype=torch.bfloat16, device=h.device)
            holder.append(h16)
        h_out, x_out, alpha, cvs = K.inter_attn_fwd(qkv, cv, H, h, x, d, rhohat, g.rp_int, g.col_int, g.red_idx,
                                                    bias_part, w_rk, w_rv, wcr, w3, clampv, s_ext, h16, deal=getattr(g, "int_deal", None))
        ctx.g, ctx.H, ctx.clampv, ctx.np = g, H, clampv, bias_part.shape[1]
        ctx.has_ext, ctx.has_cv = s_ext is not None, Wc is not None
        ctx.sink_h = _sink_of(h)
        ctx.save_for_backward(qkv, cv, d, rhohat, w_rk, w_rv, wcr, w3, alpha, cvs, v_in, Wc)
        ctx.mark_non_differentiable(alpha)
        return h_out, x_out, alpha

    @staticmethod
    def backward(ctx, dh_out, dx_out, _dalpha):
        qkv, cv, d, rhohat, w_rk, w_rv, wcr, w3, alpha, cvs, v_in, Wc = ctx.saved_tensors
        g, H = ctx.g, ctx.H
        N, E = qkv.shape[0], g.col_int.shape[0]
        dev = qkv.device
        f32 = dict(dtype=torch.float32, device=dev)
        dh_out = dh_out.contiguous() if dh_out is not None else torch.zeros((N, H), **f32)
        dx_out = dx_out.contiguous() if dx_out is not None else torch.zeros((N, 3), **f32)
        dqkv = torch.empty_like(qkv)          # every row is written: d q by pass a (zeros for nodes without inter edges), d k | d v by pass b
        dcv = torch.empty_like(cv)
        # (pass a writes d d, d rhohat, d logit and d cp of EVERY edge -- each edge lies in one row's range: no pre-zeroing)
        dd, drh = torch.empty((max(E, 1), 3), **f32), torch.empty(max(E, 1), **f32)
        n_red = E // 2
        dbias_red = torch.zeros(max(n_red, 1), **f32)
        dlogit, dcp = torch.empty(max(E, 1), **f32), torch.empty(max(E, 1), **f32)
        if E == 0:
            dd.zero_(); drh.zero_()
        # (pass a strides its waves over the rows; 7 % of the rows hold all the edges, but the draw is not what bounds it: 1,024 / 2,048 /
        #  4,096 work-groups measured 1,034 / 1,035 / 1,088 us forward + backward at the bench shape, tools/probes/inter_attn_time.py)
        nblk = min((N + 3) // 4, INTER_BWD_BLOCKS)
        wpart = torch.empty((nblk, 4 * H), **f32)                     # [block][4][H] partials of the four vector gradients
        deal = getattr(g, "int_deal", None) if K.INTER_ATTN_ROWS else None
        if deal is not None:             # the rows dealt by degree: heavy rows on four waves (csrc/inter_attn_rows.hip)
            check(load().fabind_inter_attn_bwd_rows(ptr(qkv), qkv.stride(0), ptr(cv), cv.stride(0), H, ptr(d), ptr(rhohat),
                                                    ptr(g.rp_int), ptr(g.col_int), ptr(g.mirror), ptr(g.red_idx), ptr(w_rk),
                                                    ptr(w_rv), ptr(wcr), ptr(w3), ptr(alpha), ptr(cvs), ctx.clampv, N, ptr(dh_out),
                                                    ptr(dx_out), ptr(dqkv), ptr(dcv), ptr(dd), ptr(drh), ptr(dbias_red), ptr(dlogit),
                                                    ptr(dcp), ptr(wpart), nblk, ptr(deal[0]), int(deal[1]), int(deal[2]), stream()),
                  "fabind_inter_attn_bwd_rows")
        else:
            check(load().fabind_inter_attn_bwd(ptr(qkv), qkv.stride(0), ptr(cv), cv.stride(0), H, ptr(d), ptr(rhohat),
                                               ptr(g.rp_int), ptr(g.col_int), ptr(g.mirror), ptr(g.red_idx), ptr(w_rk),
                                               ptr(w_rv), ptr(wcr), ptr(w3), ptr(alpha), ptr(cvs), ctx.clampv, N, ptr(dh_out),
                                               ptr(dx_out), ptr(dqkv), ptr(dcv), ptr(dd), ptr(drh), ptr(dbias_red), ptr(dlogit),
                                               ptr(dcp), ptr(wpart), nblk, stream()), "fabind_inter_attn_bwd")
        dw_all = K.colsum(wpart)                                      # one launch pair for all four
        dw = [dw_all[i * H:(i + 1) * H] for i in range(4)]
        dWc = dbc = None
        if ctx.has_cv:                                               # adjoint of cv = V Wc^T + bc, dV added to dqkv[:, 2H:]
            md = mm_dtype()
            dpre, dbc = dcv, None
            if dcv.dtype != md:
                in_tn = ctx.needs_input_grad[15] and ctx.needs_input_grad[16] and _db_in_tn(dcv)
                dpre, dbc = _mul_dact_colsum(dcv, None, K.ACT_NONE, md) if (FUSE_DB and not in_tn) else (_mul_dact(dcv, None, K.ACT_NONE, md), None)
            K.gemm(dpre, Wc.t().contiguous(), out=dqkv[:, 2 * H:], accumulate=True)
            if ctx.needs_input_grad[15] and ctx.needs_input_grad[16] and dbc is None:
                dWc, dbc = _weight_grad(dpre, v_in, K.ACT_NONE, None, Wc.dtype, want_db=True, W=Wc)
            elif ctx.needs_input_grad[15]:
                dWc = _weight_grad(dpre, v_in, K.ACT_NONE, None, Wc.dtype, W=Wc)
            if ctx.needs_input_grad[16]:
                dbc = dbc if dbc is not None else K.colsum(dpre)
            else:
                dbc = None
            dcv = None
        if ctx.sink_h is not None and ctx.needs_input_grad[2]:
            dh_out = ctx.sink_h.deposit(dh_out)           # h_out = h + ...: the residual gradient joins h's shared buffer
        return (dqkv, dcv, dh_out, dx_out, dd[:E], drh[:E], dbias_red[:n_red, None].expand(n_red, ctx.np), dw[0], dw[1],
                dw[2], dw[3], None, None, None, dcp[:E] if ctx.has_ext else None, dWc, dbc, None, None)


INTER_BWD_BLOCKS = int(os.environ.get("FABIND_INTER_BWD_BLOCKS", "1024"))


INTER_ATTN_INPLACE = os.environ.get("FABIND_INTER_ATTN_INPLACE", "1") == "1"


def inter_attn(qkv, cv, H, h, x, d, rhohat, g, bias_part, w_rk, w_rv, wcr, w3, clampv, s_ext=None, Wc=None, bc=None, own_h=False, Wc32=None):
    """s_ext [E] (optional): per-edge scalar added to the coordinate-MLP value inside the kernel (FABind+ evaluates its
    LN-MLP coord_mlp outside); differentiable.  cv=None with (Wc, bc): cv = Linear(qkv[:, 2H:]) is evaluated here.
    own_h: the caller hands h over (a fresh intermediate nobody reads afterwards): a no-grad pass then updates it in place -- only the
    rows with inter edges are rewritten, and the bf16 operand copy that travels with h gets the same rows."""
    assert (cv is None) != (Wc is None), "inter_attn: pass either cv or its Linear (Wc, bc)"
    w16 = _cfg.get_precision() == "bf16" and h.is_contiguous()
    if Wc32 is not None and not (Wc is not None and _cfg.get_precision() == "bf16" and qkv.dtype == torch.float32 and Wc32.dtype == torch.float32
                                 and tuple(Wc32.shape) == tuple(Wc.shape) and qkv.stride(0) % 4 == 0):
        Wc32 = None
    if _needs_grad(qkv, cv, h, x, d, rhohat, bias_part, w_rk, w_rv, wcr, w3, s_ext, Wc, bc):
        K.tn_hook(Wc, bc)
        holder = [] if w16 else None
        h_out, x_out, alpha = _InterAttn.apply(qkv, cv, h, x, d, rhohat, bias_part, w_rk, w_rv, wcr, w3, H, g, clampv, s_ext, Wc, bc, holder, Wc32)
        return _attach_b16(h_out, holder[0] if holder else None), x_out, alpha
    if cv is None:
        cv = linear(qkv[:, 2 * H:], Wc, bc, W32=Wc32)
    deal = getattr(g, "int_deal", None)
    if own_h and INTER_ATTN_INPLACE and deal is not None and K.INTER_ATTN_ROWS and h.is_contiguous() and h.dtype == torch.float32 and not h.requires_grad:
        c = getattr(h, "_fab_b16", None)
        h16 = (c[1] if (c is not None and c[0] == h._version and c[1].is_contiguous()) else h.to(torch.bfloat16)) if w16 else None
        h_out, x_out, alpha, _ = K.inter_attn_fwd(qkv, cv, H, h, x, d, rhohat, g.rp_int, g.col_int, g.red_idx, bias_part,
                                                  w_rk, w_rv, wcr, w3, clampv, s_ext, h16, deal=deal, inplace=True)
        return _attach_b16(h_out, h16), x_out, alpha      # (raw writes do not bump the version counter: the copy stays attached, and is current)
    h16 = torch.empty((h.shape[0], H), dtype=torch.bfloat16, device=h.device) if w16 else None
    h_out, x_out, alpha, _ = K.inter_attn_fwd(qkv, cv, H, h, x, d, rhohat, g.rp_int, g.col_int, g.red_idx, bias_part,
                                              w_rk, w_rv, wcr, w3, clampv, s_ext, h16, deal=deal)
    return _attach_b16(h_out, h16), x_out, alpha


class _LasStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x0, las, lay, step, clampv):
        out = K.las_step(x, x0, las[0], las[1], las[2], lay.node_off, lay.c_cnt, lay.B, lay.max_n, step, clampv)
        ctx.las, ctx.lay, ctx.step, ctx.clampv = las, lay, step, clampv
        ctx.save_for_backward(x, x0, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, x0, out = ctx.saved_tensors
        las, lay = ctx.las, ctx.lay
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        check(load().fabind_las_step_bwd(ptr(x), ptr(x0), ptr(out), ptr(las[0]), ptr(las[1]), ptr(las[2]),
                                         ptr(lay.node_off), ptr(lay.c_cnt), lay.B, lay.max_n, ctx.step, ctx.clampv,
                                         ptr(dout), ptr(dx), stream()), "fabind_las_step_bwd")
        return dx, None, None, None, None, None


def las_step(x, x0, las, lay, step, clampv):
    if _needs_grad(x):
        return _LasStep.apply(x, x0, las, lay, step, clampv)
    return K.las_step(x, x0, las[0], las[1], las[2], lay.node_off, lay.c_cnt, lay.B, lay.max_n, step, clampv)


# ------------------------------------------------------------------------------------------------
# pair biases of all layers/blocks:  out[k][pair(b,i,j), o] = sum_h a0[i,h] b0[j,h] wcomp[k,o,h] + bconst[k,o]
# (k = 2*layer + {0: protein-query block, 1: ligand-query block}; o = 4 "linear" heads then 4 "gate" heads)
# ------------------------------------------------------------------------------------------------
def _pair_bias_fwd(a0b0, H, wcomp, bconst, lay):
    nblk, NO, _ = wcomp.shape
    outs = []
    # bf16 mode: one cast of the A operand shared by all blocks -- an fp32 A sends every one of these ragged-batched
    # contractions to the register-staged fallback kernel (which rounds A to bf16 for the MFMA anyway)
    a0 = a0b0[:, :H].to(torch.bfloat16) if mm_dtype() == torch.bfloat16 else a0b0[:, :H]
    for k in range(nblk):
        bmat = K.pair_bmat(a0b0[:, H:], wcomp[k], lay.c_index, mm_dtype())       # [(sumC*NO), H]
        out = torch.empty((lay.n_pairs, NO), dtype=torch.float32, device=a0b0.device)
        K.gemm(a0, bmat, bias=bconst[k].repeat(lay.sumC), out=out, groups=lay.pb_groups, n_groups=lay.B,
               max_m=lay.max_P, max_n=lay.pb_max_n, M=lay.N, N=bmat.shape[0], ldc=NO, flops=2.0 * lay.n_pairs * NO * H)
        outs.append(out)
    return outs


def _pair_bias_groups(lay, H, ld_ab, nblk, Kp):
    """Plain-group descriptors of the two adjoint contractions (cached on the layout): d a0 += Acat . BTcat^T (all blocks
    concatenated along K) and T_k = Dt_k . At^T (K = the padded protein length)."""
    key = ("pbg", H, ld_ab, nblk, Kp)
    if getattr(lay, "_pbg_key", None) != key:
        import numpy as np
        from .param_pack import upload
        B, NO = lay.B, 8
        dev = lay.node_off.device
        P, C = np.asarray(lay.P, dtype=np.int64), np.asarray(lay.C, dtype=np.int64)
        off, coff = np.asarray(lay.off[:-1], dtype=np.int64), np.asarray(lay.coff[:-1], dtype=np.int64)
        poff = np.concatenate([[0], np.cumsum(P)])[:-1]
        bb = np.arange(B, dtype=np.int64)
        lo = lambda v: (v & 0xFFFFFFFF)
        hi = lambda v: v >> 32
        zero = np.zeros(B, dtype=np.int64)
        co = (off + C) * ld_ab
        cat = np.stack([poff, P, bb * H, np.full(B, H), lo(co), hi(co), np.full(B, ld_ab), zero], 1)
        # T_k[(j,o), :] of complex b = Dt_k[b*Kp + (j,o), :] . At[b*H + h, :]^T  (K = padded protein length)
        ch = coff * NO * H
        tg = np.stack([bb * Kp, C * NO, bb * H, np.full(B, H), lo(ch), hi(ch), np.full(B, H), zero], 1)
        # one asynchronous upload from the pinned staging ring (param_pack.upload): `torch.tensor(list, device=...)` is a pageable
        # host-to-device copy, i.e. it waits for everything the stream still has queued -- in the middle of the backward pass that was
        # 2.4 ms per tensor, 4.8 ms per step at B = 64 (tools/probes/stack_hostprof.py, backward in the calling thread)
        both = (np.concatenate([cat, tg], 0) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        devt = upload(both, dev, torch.int32)
        lay._pbg = (devt[:B], devt[B:])
        lay._pbg_key = key
    return lay._pbg


class _PairBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a0b0, wcomp, bconst, H, lay):
        ctx.H, ctx.lay = H, lay
        ctx.sink = getattr(a0b0, "_fab_gsink", None)      # shared gradient buffer of a0b0 (shared_grad), or None
        ctx.save_for_backward(a0b0, wcomp)
        return tuple(_pair_bias_fwd(a0b0.detach(), H, wcomp.detach(), bconst.detach(), lay))

    @staticmethod
    def backward(ctx, *douts):
        a0b0, wcomp = ctx.saved_tensors
        H, lay = ctx.H, ctx.lay
        nblk, NO, _ = wcomp.shape
        dev = a0b0.device
        da0b0, da0b0_ret = _sink_zeros(a0b0, ctx.sink)    # every writer below accumulates (+=, accumulating GEMM, atomics)
        dwcomp = None
        dbconst = torch.zeros((nblk, NO), dtype=torch.float32, device=dev)
        # ('bf16x3': the adjoint of the pair-bias contraction takes the bf16 route too -- see _x3_tn_ok; FABIND_X3_PAIRBIAS_BWD=fp32
        #  restores the fp32 atomics kernels, 30 ms per step at the headline shape)
        mode = _cfg.get_precision()
        bf16 = (mode == "bf16" or (mode == "bf16x3" and _cfg.x3_backward_bf16())) and NO == 8 and H % 8 == 0 and nblk <= 16
        if bf16:
            # all blocks at once: the gradients of every block, bf16, concatenated along K and padded per complex (Acat) and
            # the matching b0 * wcomp operand (BTcat) -> d a0 is ONE plain-group GEMM on the pipelined kernel (one accumulating
            # epilogue pass over the strided fp32 gradient instead of one per block); T_k = D_k^T a0 reads column slice k
            Kp = (lay.max_C * NO + 31) // 32 * 32
            cat_g, t_g = _pair_bias_groups(lay, H, a0b0.stride(0), nblk, Kp)
            a16 = a0b0.to(torch.bfloat16)
            T = torch.empty((lay.sumC * NO, H), dtype=torch.float32, device=dev)
            ds = [None if d is None else d.contiguous() for d in douts]
            Acat = torch.empty((lay.sumP, nblk * Kp), dtype=torch.bfloat16, device=dev)
            BTcat = torch.empty((lay.B * H, nblk * Kp), dtype=torch.bfloat16, device=dev)
            ptrs = (ctypes.c_void_p * nblk)(*[None if d is None else d.data_ptr() for d in ds])
            wc = wcomp.contiguous()
            n_part = int(load().fabind_pair_bias_cat_parts(lay.B, lay.max_P))
            colpart = torch.empty((n_part, nblk * NO), dtype=torch.float32, device=dev)   # per-work-group column sums of the douts
            check(load().fabind_pair_bias_cat(ptrs, nblk, ptr(lay.desc_p), lay.B, lay.max_P, Kp, ptr(Acat), Acat.stride(0),
                                              ptr(a0b0[:, H:]), a0b0.stride(0), ptr(lay.c_index), ptr(wc), H, ptr(BTcat),
                                              BTcat.stride(0), ptr(colpart), stream()), "fabind_pair_bias_cat")
            dbconst = K.colsum(colpart).reshape(nblk, NO)                                  # = column sums of every dout_k
            K.gemm(Acat, BTcat, out=da0b0, accumulate=True, groups=cat_g, n_groups=lay.B, max_m=lay.max_P, max_n=H,
                   M=lay.sumP, N=lay.B * H, ldc=a0b0.stride(0), flops=2.0 * lay.n_pairs * NO * nblk * H)
            # the T_k = D_k^T a0 contractions on the pipelined NT kernel: K-major, uniformly padded copies of both operands
            Pp = (lay.max_P + 63) // 64 * 64          # K of these contractions: a multiple of 64 selects the LDS-DMA pipelined kernel
            Dt = torch.empty((nblk, lay.B * Kp, Pp), dtype=torch.bfloat16, device=dev)
            At = torch.empty((lay.B * H, Pp), dtype=torch.bfloat16, device=dev)
            check(load().fabind_batched_transpose_pad(ptr(Acat), Acat.stride(0), ptr(lay.desc_p), lay.B, nblk, Kp, Pp, ptr(Dt),
                                                      stream()), "fabind_batched_transpose_pad")
            check(load().fabind_batched_transpose_pad(ptr(a16), a16.stride(0), ptr(lay.desc_pf), lay.B, 1, H, Pp, ptr(At), stream()),
                  "fabind_batched_transpose_pad")
            # per-group partials of every block's d wcomp side by side: [groups of 16 ligand-side nodes, nblk * 8 * H] -> one column sum
            n_fin = int(load().fabind_pair_bias_finish_parts(lay.sumC))
            fin = (torch.empty if all(d is not None for d in douts) else torch.zeros)((n_fin, nblk * NO * H), dtype=torch.float32, device=dev)
        for k, dout in enumerate(douts):
            if dout is None:
                continue
            dout = dout.contiguous()
            if not bf16:
                dbconst[k] = K.colsum(dout)
            if bf16:
                K.gemm(Dt[k], At, out=T, groups=t_g, n_groups=lay.B, max_m=lay.max_C * NO, max_n=H, M=lay.B * Kp, N=lay.B * H, ldc=H,
                       flops=2.0 * lay.n_pairs * NO * H)
                check(load().fabind_pair_bias_finish(ptr(T), ptr(a0b0), a0b0.stride(0), H, ptr(wcomp[k]), ptr(lay.c_index),
                                                     lay.sumC, ptr(da0b0), ptr(fin[:, k * NO * H:]), fin.stride(0), stream()),
                      "fabind_pair_bias_finish")
            else:
                if dwcomp is None:
                    dwcomp = torch.zeros_like(wcomp)
                dwk = torch.zeros((lay.B, NO * H), dtype=torch.float32, device=dev)
                check(load().fabind_pair_bias_bwd(ptr(dout), NO, ptr(a0b0), a0b0.stride(0), H, ptr(wcomp[k]), ptr(lay.desc_p),
                                                  lay.B, lay.max_P, lay.max_C, ptr(lay.p_index), ptr(lay.c_index), ptr(da0b0),
                                                  ptr(dwk), stream()), "fabind_pair_bias_bwd")
                dwcomp[k] = K.colsum(dwk).reshape(NO, H)
        if bf16:
            dwcomp = K.colsum(fin).view(nblk, NO, H)           # fixed order: no float atomics anywhere on this path
        return da0b0_ret, dwcomp, dbconst, None, None


def pair_bias(a0b0, H, wcomp, bconst, lay):
    if _needs_grad(a0b0, wcomp, bconst):
        return list(_PairBias.apply(a0b0, wcomp, bconst, H, lay))
    return _pair_bias_fwd(a0b0, H, wcomp, bconst, lay)


class PairBias:
    """The RowAttention pair biases of every layer and block of one stack call, in the two forms the attention kernels take:
    * `tensors()` -- eight [pairs, 8] fp32 tensors (differentiable; what the autograd path and the fp32-storage modes use);
    * `fused(k)` -- the operands of the kernels that recompute block k's bias on the matrix cores (bf16 forward-only passes:
      inference, and the no-grad refinement iterations of training): the bf16 a0 rows, block k's packed ligand-side operand and
      its eight constants.  Both are built on first use and kept for the refinement iterations that follow."""

    def __init__(self, a0b0, H, wcomp, bconst, lay):
        self.a0b0, self.H, self.wcomp, self.bconst, self.lay = a0b0, H, wcomp, bconst, lay
        self._tensors, self._a16, self._bo = None, None, {}
        self._bot, self._bw = {}, None            # fused backward: Bo^T packs per block; state shared by the blocks of one backward pass
        self._n_fused = 0
        self._fused_blocks = set()                # blocks with a differentiable fused forward whose backward has not run yet
        # a differentiable pass will come (training: the last refinement iteration) and its backward cannot recompute the bias in the
        # kernels, or nothing can be fused: build the tensors NOW, under the caller's grad mode -- a first use inside a no-grad refinement
        # iteration would cache tensors without a graph
        if (_needs_grad(a0b0, wcomp, bconst) and not self.can_fuse_train()) or not self.can_fuse():
            self.tensors()

    def can_fuse(self):
        # (the kernels assume wcomp[k] rows ordered lin0..3 | gate0..3 for 4 heads of 32 channels: engine._stack_requests builds them so)
        return (K.CROSS_ATTN_FUSED and _cfg.get_precision() == "bf16"
                and self.lay.max_C <= K.CROSS_ATTN_FUSED_MAX_C
                and self.H in (64, 128, 256, 512) and tuple(self.wcomp.shape[1:]) == (8, self.H))

    def can_fuse_train(self):
        """The differentiable pass takes the fused kernels each way (forward with the log-sum-exp saved, backward recomputing the bias)."""
        # (large batches only: one work-group per CU and 64 protein rows per work-group -- a pocket-sized batch is 128 work-groups per
        #  launch, and its tensor-path kernels are 25-45 us each: same-box pocket step 3,086 fused vs 3,532 complexes/s, headline 731 vs 722)
        tiles = self.lay.B * ((self.lay.max_P + 63) // 64)
        return FUSED_ATTN_TRAIN and tiles >= FUSED_ATTN_TRAIN_MIN_TILES and self.can_fuse() and self.a0b0.dtype == torch.float32 \
            and self.a0b0.stride(0) == 2 * self.H

    def has_tensors(self):
        return self._tensors is not None

    def bot(self, k):
        with torch.no_grad():
            if not self._bot:
                self._bot = dict(enumerate(K.pair_bot_pack(self.a0b0[:, self.H:], self.wcomp, self.H, self.lay)))
        return self._bot[k]

    def tensors(self):
        if self._tensors is None:
            self._tensors = pair_bias(self.a0b0, self.H, self.wcomp, self.bconst, self.lay)
        return self._tensors

    def fused(self, k):
        with torch.no_grad():
            if self._a16 is None:
                self._a16 = self.a0b0[:, :self.H].to(torch.bfloat16)
            if not self._bo:                 # the packed operands of ALL blocks from one launch
                self._bo = dict(enumerate(K.pair_bo_pack(self.a0b0[:, self.H:], self.wcomp, self.H, self.lay)))
                self._bc = self.bconst.float().contiguous()
            bo = self._bo[k]
        return self._a16, bo, self._bc[k]


ATTN_DA0_IN_KERNEL = os.environ.get("FABIND_ATTN_DA0_IN_KERNEL", "0") == "1"    # 1: d a0 contracted inside every block's backward kernel (A/B)
FUSED_ATTN_TRAIN_MIN_TILES = int(os.environ.get("FABIND_ATTN_FUSED_TRAIN_MIN_TILES", "512"))
FUSED_ATTN_TRAIN = os.environ.get("FABIND_ATTN_FUSED_TRAIN", "1") == "1"     # 0: the differentiable pass reads / writes [pairs, 8] bias tensors (round 3)
_SLOT_TO_ROW = [0, 4, 1, 5, 2, 6, 3, 7]          # slot order of the packed operands (lin0, gate0, lin1, gate1, ...) -> row of wcomp / bconst (lin0..3, gate0..3)


class _CrossAttnFused(torch.autograd.Function):
    """Gated cross attention of one block with the pair bias recomputed in the kernels, BOTH ways (csrc/attn_mfma.hip): no [pairs, 8]
    tensor exists in either direction.  Inputs with a gradient: qg (q | gate pre-activations), kv, a0b0 (the factored pair embedding:
    a0 through the kernel's own contraction, b0 through the T = D^T a0 GEMM + finishing pass), wcomp, bconst (all blocks' composed
    weights / constants: every block's backward fills its slice of ONE gradient tensor, the last block to run hands it to autograd)."""

    @staticmethod
    def forward(ctx, qg, kv, a0b0, wcomp, bconst, pb, k_blk, mode, scale):
        lay = pb.lay
        a16, bo, bc = pb.fused(k_blk)
        nq = qg.shape[0]
        out = (torch.zeros if mode == 0 else torch.empty)((nq, 128), dtype=torch.float32, device=qg.device)      # rows outside every block: 0
        lse = torch.empty((nq, 4), dtype=torch.float32, device=qg.device)
        K.cross_attn_fused_fwd(qg[:, :128], kv[:, :128], kv[:, 128:], qg[:, 128:], a16, bo, bc, lay, pb.H, mode, scale, out, lse)
        ctx.pb, ctx.k, ctx.mode, ctx.scale = pb, k_blk, mode, scale
        ctx.sink = getattr(a0b0, "_fab_gsink", None)
        ctx.save_for_backward(qg, kv, out, lse, a0b0, wcomp)
        pb._n_fused += 1
        return out

    @staticmethod
    def backward(ctx, dout):
        qg, kv, out, lse, a0b0, wcomp = ctx.saved_tensors
        pb, k, mode = ctx.pb, ctx.k, ctx.mode
        lay, H = pb.lay, pb.H
        dev = qg.device
        nblk, NO = wcomp.shape[0], 8
        dout = dout.contiguous()
        bw = pb._bw
        if bw is None:                       # first block of this backward pass: the state the blocks share
            Kp = (lay.max_C * NO + 31) // 32 * 32
            n_tiles = int(load().fabind_cross_attn_fused_bwd_parts(lay.B, lay.max_P))
            bw = pb._bw = dict(Kp=Kp, left=pb._n_fused, done=[],
                               acat=torch.empty((lay.sumP, nblk * Kp), dtype=torch.bfloat16, device=dev),
                               colpart=torch.zeros((n_tiles, nblk * NO), dtype=torch.float32, device=dev))
        Kp = bw["Kp"]
        da0b0, da0b0_ret = _sink_zeros(a0b0, ctx.sink)
        a16, bo, bc = pb.fused(k)
        # uncovered rows (the ligand rows of a node-layout array) must read zero: they flow into the projections' adjoints
        dqg = (torch.zeros_like if mode == 0 else torch.empty_like)(qg)
        dkv = (torch.empty_like if mode == 0 else torch.zeros_like)(kv)
        in_k = ATTN_DA0_IN_KERNEL
        K.cross_attn_fused_bwd(qg, kv, a16, bo, pb.bot(k) if in_k else None, bc, lay, H, mode, ctx.scale, out, lse, dout, dqg, dkv,
                               da0b0 if in_k else None, bw["acat"], k * Kp, bw["colpart"][:, k * NO:], Kp)
        bw["done"].append(k)
        bw["left"] -= 1
        dw_ret = db_ret = None
        if bw["left"] <= 0:
            # The last block of the pass finishes ALL of them: the ligand-side and weight gradients need T_k[(j, slot), h] = sum_i
            # D_k[i, (j, slot)] a0[i, h] over ALL rows of a complex -- one batched transpose of the bf16 gradient rows, one ragged GEMM and
            # one finishing pass per block, one column sum for all weight gradients and one for all constants -- and hands the shared
            # gradients to autograd.
            Pp = (lay.max_P + 63) // 64 * 64
            cat_g, t_g = _pair_bias_groups(lay, H, a0b0.stride(0), nblk, Kp)
            acat = bw["acat"]
            wslot = wcomp.float()[:, _SLOT_TO_ROW].contiguous()              # [nblk, 8, H], rows in the packed operands' slot order
            ran = sorted(bw["done"])
            if not in_k:
                # d a0 of ALL blocks as ONE ragged GEMM over the stored gradient rows (K = nblk x C x 8): the gradient is written once
                if len(ran) != nblk:
                    for kk in set(range(nblk)) - set(ran):
                        acat[:, kk * Kp:(kk + 1) * Kp].zero_()
                BTcat = torch.empty((lay.B * H, nblk * Kp), dtype=torch.bfloat16, device=dev)
                check(load().fabind_pair_bias_btcat(ptr(a0b0[:, H:]), a0b0.stride(0), ptr(lay.c_index), ptr(lay.desc_p), ptr(wslot), nblk, H, Kp,
                                                    ptr(BTcat), BTcat.stride(0), lay.B, stream()), "fabind_pair_bias_btcat")
                K.gemm(acat, BTcat, out=da0b0, accumulate=True, groups=cat_g, n_groups=lay.B, max_m=lay.max_P, max_n=H,
                       M=lay.sumP, N=lay.B * H, ldc=a0b0.stride(0), flops=2.0 * lay.n_pairs * NO * nblk * H)
            Dt = torch.empty((nblk, lay.B * Kp, Pp), dtype=torch.bfloat16, device=dev)
            At = torch.empty((lay.B * H, Pp), dtype=torch.bfloat16, device=dev)
            check(load().fabind_batched_transpose_pad(ptr(acat), acat.stride(0), ptr(lay.desc_p), lay.B, nblk, Kp, Pp, ptr(Dt), stream()),
                  "fabind_batched_transpose_pad")
            check(load().fabind_batched_transpose_pad(ptr(a16), a16.stride(0), ptr(lay.desc_pf), lay.B, 1, H, Pp, ptr(At), stream()),
                  "fabind_batched_transpose_pad")
            n_fin = int(load().fabind_pair_bias_finish_parts(lay.sumC))
            fin = (torch.empty if len(ran) == nblk else torch.zeros)((n_fin, nblk * NO * H), dtype=torch.float32, device=dev)
            T = torch.empty((lay.sumC * NO, H), dtype=torch.float32, device=dev)
            for kk in ran:
                K.gemm(Dt[kk], At, out=T, groups=t_g, n_groups=lay.B, max_m=lay.max_C * NO, max_n=H, M=lay.B * Kp, N=lay.B * H, ldc=H,
                       flops=2.0 * lay.n_pairs * NO * H)
                check(load().fabind_pair_bias_finish(ptr(T), ptr(a0b0), a0b0.stride(0), H, ptr(wslot[kk]), ptr(lay.c_index), lay.sumC,
                                                     ptr(da0b0), ptr(fin[:, kk * NO * H:]), fin.stride(0), stream()), "fabind_pair_bias_finish")
            inv = [_SLOT_TO_ROW.index(r_) for r_ in range(NO)]              # wcomp / bconst row -> slot
            dw_ret = K.colsum(fin).view(nblk, NO, H)[:, inv].contiguous().to(wcomp.dtype)
            db_ret = K.colsum(bw["colpart"]).view(nblk, NO)[:, inv].contiguous()
            pb._bw = None
            pb._fused_blocks.clear()                  # the pass is complete: the context may serve another one (_n_fused stays: a second
            #                                           backward over a retained graph counts the same blocks down again)
        return dqg, dkv, da0b0_ret, dw_ret, db_ret, None, None, None, None


def cross_attn_fused_train(qg, kv, pb, k_blk, mode, scale):
    """Differentiable gated cross attention of block k_blk with the pair bias recomputed in the kernels both ways.
    The blocks of ONE differentiable pass share state on `pb` (the K-concatenated gradient rows, the count of blocks still to run their
    backward): a PairBias / StackContext serves one differentiable pass at a time -- a second differentiable forward of the same block
    before the first pass's backward has run would overwrite its slice, so it is refused here."""
    if k_blk in pb._fused_blocks:
        raise RuntimeError("fabind_amd: block %d of this PairBias already has a differentiable fused forward whose backward has not run -- a "
                           "StackContext serves ONE differentiable pass at a time (build a new context, or run backward first)" % k_blk)
    if not pb._fused_blocks:
        pb._bw, pb._n_fused = None, 0             # a new pass starts (also drops the leftovers of a pass whose backward never completed)
    pb._fused_blocks.add(k_blk)
    return _CrossAttnFused.apply(qg, kv, pb.a0b0, pb.wcomp, pb.bconst, pb, k_blk, mode, scale)


def cross_attn_fused(qg, kv, pb, k_blk, mode, lay, scale):
    """Forward-only gated cross attention of block k_blk with the pair bias recomputed in the kernel (csrc/attn_mfma.hip)."""
    a16, bo, bc = pb.fused(k_blk)
    out = torch.zeros((qg.shape[0], 128), dtype=torch.float32, device=qg.device) if mode == 0 else \
        torch.empty((qg.shape[0], 128), dtype=torch.float32, device=qg.device)
    return K.cross_attn_fused_fwd(qg[:, :128], kv[:, :128], kv[:, 128:], qg[:, 128:], a16, bo, bc, lay, pb.H, mode, scale, out)


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        R, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        check(load().fabind_layernorm_fwd(ptr(x), ptr(w), ptr(b), eps, R, C, ptr(y), ptr(mean), ptr(rstd), stream()),
              "fabind_layernorm_fwd")
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        R, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nb = (R + 3) // 4
        dwp = torch.empty((nb, C), dtype=torch.float32, device=x.device)
        dbp = torch.empty((nb, C), dtype=torch.float32, device=x.device)
        check(load().fabind_layernorm_bwd(ptr(x), ptr(w), ptr(dy), ptr(mean), ptr(rstd), R, C, ptr(dx), ptr(dwp), ptr(dbp),
                                          stream()), "fabind_layernorm_bwd")
        return dx, K.colsum(dwp), K.colsum(dbp), None


def layernorm(x, w, b, eps=1e-5):
    return _LayerNorm.apply(x.float().contiguous(), w, b, eps)


# ------------------------------------------------------------------------------------------------
# Heads (round 6; csrc/heads.hip): the Gumbel-softmax pocket centre and the six-term training loss with its gradient
# seeds -- one launch each way instead of chains of element-wise torch ops
# ------------------------------------------------------------------------------------------------
class _PocketCenter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mask_u8, xyz, noise, tau, hard):
        B, L = logits.shape
        center = torch.empty((B, 3), dtype=torch.float32, device=logits.device)
        wsum = torch.empty(B, dtype=torch.float32, device=logits.device)
        check(load().fabind_pocket_center_fwd(ptr(logits), ptr(mask_u8), ptr(xyz), ptr(noise), B, L, float(tau), int(bool(hard)),
                                              ptr(center), ptr(wsum), stream()), "fabind_pocket_center_fwd")
        ctx.save_for_backward(logits, mask_u8, xyz, noise if noise is not None else logits.new_empty(0), center, wsum)
        ctx.tau, ctx.has_noise = float(tau), noise is not None
        return center

    @staticmethod
    def backward(ctx, dcenter):
        logits, mask_u8, xyz, noise, center, wsum = ctx.saved_tensors
        B, L = logits.shape
        dlogits = torch.empty_like(logits)
        check(load().fabind_pocket_center_bwd(ptr(logits), ptr(mask_u8), ptr(xyz), ptr(noise) if ctx.has_noise else None, B, L, ctx.tau,
                                              ptr(center), ptr(wsum), ptr(dcenter.contiguous().float()), ptr(dlogits), stream()),
              "fabind_pocket_center_bwd")
        return dlogits, None, None, None, None, None


def pocket_center(logits, mask, xyz, tau=1.0, hard=False, noise=None):
    """center [B, 3] of the reference's Gumbel-softmax pocket head (models/model.py:146-158): logits / mask [B, L], xyz [B, L, 3];
    noise: the Gumbel samples [B, L, 2] of train mode (None: gumbel_softmax_no_random).  One kernel forward, one backward."""
    lg = logits.float().contiguous()
    m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
    return _PocketCenter.apply(lg, m8, xyz.float().contiguous(), None if noise is None else noise.float().contiguous(), tau, hard)


_LOSS_TICKET = {}
_CLS_DT = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.uint8: 3, torch.bool: 3}
LOSS_TERM_NAMES = ("pocket_cls", "pocket_center", "contact", "contact_by_pred", "distill", "coord")


class _SixTermLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, y_pred, y_by, logits, center, coords_true, dis_map, cls, mask_u8, center_true, w):
        dev = coords.device
        key = (dev, stream())                      # (one arrival counter per device AND stream: two streams must not share it)
        tk = _LOSS_TICKET.get(key)
        if tk is None:
            tk = _LOSS_TICKET[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        n_pair, n_coord, n_cls, n_center = y_pred.numel(), coords.numel(), logits.numel(), center.numel()
        nblk = load().fabind_loss_blocks(n_pair, n_coord, n_cls)
        part = torch.empty((nblk, 8), dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        terms = torch.empty(6, dtype=torch.float32, device=dev)
        cls_dt = _CLS_DT[cls.dtype]
        check(load().fabind_loss_fwd(ptr(coords), ptr(coords_true), n_coord, ptr(y_pred), ptr(y_by), ptr(dis_map), n_pair, ptr(logits),
                                     ptr(cls), cls_dt, ptr(mask_u8), n_cls, ptr(center), ptr(center_true), n_center, w['coord'], w['pair'],
                                     w['distill'], w['cls'], w['center'], w['delta'], ptr(part), ptr(tk), ptr(out), ptr(loss), ptr(terms),
                                     stream()), "fabind_loss_fwd")
        ctx.save_for_backward(coords, y_pred, y_by, logits, center, coords_true, dis_map, cls, center_true, out)
        ctx.w, ctx.cls_dt = w, cls_dt
        return loss, terms

    @staticmethod
    def backward(ctx, g_loss, g_terms):
        coords, y_pred, y_by, logits, center, coords_true, dis_map, cls, center_true, out = ctx.saved_tensors
        w = ctx.w
        need = ctx.needs_input_grad
        d = [torch.empty_like(t) if need[i] else None for i, t in enumerate((coords, y_pred, y_by, logits, center))]
        gl = None if g_loss is None else g_loss.contiguous().float()
        gt = None if g_terms is None else g_terms.contiguous().float()
        check(load().fabind_loss_bwd(ptr(coords), ptr(coords_true), coords.numel(), ptr(y_pred), ptr(y_by), ptr(dis_map), y_pred.numel(),
                                     ptr(logits), ptr(cls), ctx.cls_dt, logits.numel(), ptr(center), ptr(center_true), center.numel(),
                                     w['coord'], w['pair'], w['distill'], w['cls'], w['center'], w['delta'], ptr(out), ptr(gl), ptr(gt),
                                     ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), stream()), "fabind_loss_bwd")
        return d[0], d[1], d[2], d[3], d[4], None, None, None, None, None, None


def six_term_loss(coords, y_pred, y_by, logits, center, coords_true, dis_map, pocket_cls, p_mask, center_true, w):
    """The reference's train-step loss (main_fabind.py:398-417) -> (loss, {name: term}); ONE launch for the six terms and their sum, ONE for
    the five gradient seeds.  w: dict(coord, pair, distill, cls, center, delta)."""
    f = lambda t: t.float().contiguous()
    cls = pocket_cls.contiguous()
    if cls.dtype not in _CLS_DT:
        cls = cls.float()
    m8 = p_mask.contiguous().view(torch.uint8) if p_mask.dtype == torch.bool else p_mask.to(torch.uint8).contiguous()
    w = {k: float(v) for k, v in w.items()}
    loss, terms = _SixTermLoss.apply(f(coords), f(y_pred), f(y_by), f(logits), f(center), f(coords_true), f(dis_map), cls, m8,
                                     f(center_true), w)
    tt = terms.unbind(0)
    return loss, {n: tt[i] for i, n in enumerate(LOSS_TERM_NAMES)}


# ------------------------------------------------------------------------------------------------
# Distance-map head on per-complex pair blocks (round 6; csrc/heads.hip): the valid (pocket residue, ligand atom) pairs of a complex are a
# dense P_b x C_b block, protein-major -- the adjoints no longer sort 3.84 M pair keys into a CSR per step
# ------------------------------------------------------------------------------------------------
_PAIRBLOCK = np.dtype([("pair_off", np.int64), ("P", np.int32), ("C", np.int32), ("p_row0", np.int32), ("c_row0", np.int32),
                       ("tile0", np.int32), ("pad", np.int32)])          # = struct PairBlock of csrc/heads.hip


class PairBlocks:
    """Block descriptors of the pair lists `IaBNet._pair_lists` builds: kcnt[b] pocket residues x ncnt[b] ligand atoms per complex (host
    lists).  Pocket rows and ligand rows are complex-contiguous in their own arrays."""

    def __init__(self, kcnt, ncnt, dev):
        from .param_pack import _upload
        kc, nc = np.asarray(kcnt, dtype=np.int64), np.asarray(ncnt, dtype=np.int64)
        B = kc.shape[0]
        tp, ch = load().fabind_pair_block_tile(), load().fabind_pair_block_chunk()
        tiles = (kc + tp - 1) // tp
        d = np.zeros(B, dtype=_PAIRBLOCK)
        d["pair_off"] = np.cumsum(kc * nc) - kc * nc
        d["P"], d["C"] = kc, nc
        d["p_row0"], d["c_row0"] = np.cumsum(kc) - kc, np.cumsum(nc) - nc
        d["tile0"] = np.cumsum(tiles) - tiles
        self.B, self.n_tiles, self.n_pairs = int(B), int(tiles.sum()), int((kc * nc).sum())
        self.max_C, self.n_prows, self.n_crows = int(nc.max()) if B else 0, int(kc.sum()), int(nc.sum())
        self.nchunk_max = (self.max_C + ch - 1) // ch
        self.chunk = ch
        row_b = np.repeat(np.arange(B, dtype=np.int32), nc)
        raw = np.concatenate([d.view(np.uint8), row_b.view(np.uint8)]) if B else np.zeros(0, dtype=np.uint8)
        self.buf = _upload(raw, dev) if raw.shape[0] else None            # one host-to-device copy: descriptors, then row_b
        self.desc_bytes = d.view(np.uint8).shape[0]

    def tensors(self):
        return [self.buf] if self.buf is not None else []

    @property
    def desc_ptr(self):
        return self.buf.data_ptr()

    @property
    def row_b_ptr(self):
        return self.buf.data_ptr() + self.desc_bytes


class _PairDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, xc, blocks, scale, lo, hi):
        y = torch.empty(blocks.n_pairs, dtype=torch.float32, device=xc.device)
        check(load().fabind_pair_dist_fwd(blocks.desc_ptr, blocks.B, blocks.n_tiles, ptr(xp), ptr(xc), scale, lo, hi, ptr(y), stream()),
              "fabind_pair_dist_fwd")
        ctx.save_for_backward(xp, xc)
        ctx.blocks, ctx.k = blocks, (scale, lo, hi)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, xc = ctx.saved_tensors
        bl = ctx.blocks
        part = torch.empty((bl.B, 8, bl.max_C, 3), dtype=torch.float32, device=xc.device)
        dxc = torch.empty_like(xc)
        check(load().fabind_pair_dist_bwd(bl.desc_ptr, bl.B, bl.max_C, ptr(xp), ptr(xc), ptr(dy.contiguous().float()), ctx.k[0], ctx.k[1],
                                          ctx.k[2], ptr(part), ptr(dxc), stream()), "fabind_pair_dist_bwd")
        return None, dxc, None, None, None, None


def pair_dist(xp, xc, blocks, scale=1.0, lo=0.0, hi=10.0):
    """clamp(scale * |xp_i - xc_j|, lo, hi) over the valid pairs of every complex (reference model.py:349,363-365: cdist -> [z_mask] ->
    unnormalize -> clamp).  xp [sum P, 3] (data: no gradient), xc [sum C, 3]; the adjoint to xc is a fixed-order sum."""
    assert not xp.requires_grad, "pair_dist: the pocket coordinates are data"
    xp, xc = xp.float().contiguous(), xc.float().contiguous()
    assert xp.shape[0] == blocks.n_prows and xc.shape[0] == blocks.n_crows
    return _PairDist.apply(xp, xc, blocks, float(scale), float(lo), float(hi))
