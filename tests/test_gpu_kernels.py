"""GPU: unit parity of the HIP kernels (through the C ABI) against plain torch fp32 on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture
def x3_mode():
    """split-bf16 contractions for fp32 operands (config 'bf16x3'), restored afterwards"""
    from fabind_amd import config
    old = config.get_precision()
    config.set_precision("bf16x3")
    yield
    config.set_precision(old)


@pytest.mark.parametrize("wm", [2, 4])
def test_gemm_split_bf16_accuracy_and_forms(x3_mode, wm):
    """fabind_gemm with split3: fp32 operands contracted as hi/lo bf16 pairs (three MFMAs per term).  Against float64: the error
    must sit at the 2^-16 operand-split level (two orders below bf16, within ~30x of the exact-fp32 MFMA), in every form the fp32
    path uses: plain, K-concatenated A | A2, bias / SiLU epilogue with the stored derivative, residual, accumulate, row-dot,
    split-K partials and ragged groups.  Both tile heights (fabind_gemm_set_x3_tile)."""
    from fabind_amd import _lib, config
    from fabind_amd import kernels as K
    dev = _dev()
    _lib.load().fabind_gemm_set_x3_tile(wm)
    try:
        g = torch.Generator().manual_seed(11 + wm)
        for (M, N, Kd) in [(1000, 512, 512), (257, 136, 96), (3000, 1024, 1536), (64, 8, 32)]:
            A = torch.randn(M, Kd, generator=g, dtype=torch.float64)
            W = torch.randn(N, Kd, generator=g, dtype=torch.float64) / Kd ** 0.5
            ref = A @ W.T
            Ad, Wd = A.float().to(dev), W.float().to(dev)
            ref32 = Ad.double().cpu() @ Wd.double().cpu().T                     # what exact arithmetic gives on the fp32-rounded operands
            out, _ = K.gemm(Ad, Wd)
            e_x3 = float((out.double().cpu() - ref32).abs().max() / ref32.abs().max())
            config.set_precision("fp32")
            out32, _ = K.gemm(Ad, Wd)
            config.set_precision("bf16x3")
            e_32 = float((out32.double().cpu() - ref32).abs().max() / ref32.abs().max())
            e_16 = float(((Ad.bfloat16().double().cpu() @ Wd.bfloat16().double().cpu().T) - ref32).abs().max() / ref32.abs().max())
            print("split-bf16 GEMM %dx%dx%d (tile %d rows): max err / max |C|: x3 %.2e, exact-fp32 MFMA %.2e, plain bf16 operands %.2e"
                  % (M, N, Kd, 64 * wm, e_x3, e_32, e_16))
            assert e_x3 < 2e-5 and e_x3 < e_16 / 100
        # forms
        M, N, Kd = 777, 200, 256
        A = torch.randn(M, Kd, generator=g)
        W = torch.randn(N, Kd, generator=g) / Kd ** 0.5
        b, R, u = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(N, generator=g)
        Ad, Wd = A.to(dev), W.to(dev)
        pre = (A.double() @ W.double().T + b.double())
        tol = lambda r: 3e-5 * max(1.0, float(r.abs().max()))
        out, _ = K.gemm(Ad[:, :128].contiguous(), Wd, A2=Ad[:, 128:].contiguous(), bias=b.to(dev), residual=R.to(dev))
        assert (out.double().cpu() - (pre + R.double())).abs().max() <= tol(pre)
        D = torch.empty(M, N, device=dev)
        out, _ = K.gemm(Ad, Wd, bias=b.to(dev), act_epi=K.ACT_SILU, out2=D)
        sg = torch.sigmoid(pre)
        assert (out.double().cpu() - pre * sg).abs().max() <= tol(pre)
        assert (D.double().cpu() - sg * (1 + pre * (1 - sg))).abs().max() <= 3e-5 * 2
        C = R.to(dev).clone()
        K.gemm(Ad, Wd, out=C, accumulate=True)
        assert (C.double().cpu() - (A.double() @ W.double().T + R.double())).abs().max() <= tol(pre)
        _, part = K.gemm(Ad, Wd, bias=b.to(dev), act_epi=K.ACT_RELU, dotvec=u.to(dev), want_out=False)
        refd = (torch.relu(pre) * u.double()).sum(1)
        assert (part.sum(1).double().cpu() - refd).abs().max() <= 1e-4 * max(1.0, float(refd.abs().max()))
        parts, _ = K.gemm(Ad, Wd, k_splits=2)
        assert (parts.sum(0).double().cpu() - A.double() @ W.double().T).abs().max() <= tol(pre)
        # ragged groups {a_row0, M, w_row0, N, c_off lo, c_off hi, ldc, -}: three blocks of different sizes into one output list
        ms, ns = [100, 33, 260], [40, 8, 136]
        a0 = np.concatenate([[0], np.cumsum(ms)]); w0 = np.concatenate([[0], np.cumsum(ns)])
        c0 = np.concatenate([[0], np.cumsum([m * n for m, n in zip(ms, ns)])])
        Ag = torch.randn(int(a0[-1]), Kd, generator=g); Wg = torch.randn(int(w0[-1]), Kd, generator=g) / Kd ** 0.5
        grp = torch.tensor([[a0[i], ms[i], w0[i], ns[i], c0[i], 0, ns[i], 0] for i in range(3)], dtype=torch.int32, device=dev)
        outg = torch.zeros(int(c0[-1]), device=dev)
        K.gemm(Ag.to(dev), Wg.to(dev), out=outg, groups=grp, n_groups=3, max_m=max(ms), max_n=max(ns), M=int(a0[-1]), N=int(w0[-1]), ldc=1)
        for i in range(3):
            r = Ag[a0[i]:a0[i + 1]].double() @ Wg[w0[i]:w0[i + 1]].double().T
            got = outg[c0[i]:c0[i + 1]].view(ms[i], ns[i]).double().cpu()
            assert (got - r).abs().max() <= tol(r), i
    finally:
        _lib.load().fabind_gemm_set_x3_tile(2)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(300, 96, 64), (129, 130, 56), (1000, 512, 544), (17, 1, 2048), (5, 256, 128)])
def test_gemm_epilogues(mode, shape):
    from fabind_amd import kernels as K
    dev = _dev()
    M, N, Kd = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, Kd, generator=g)
    W = torch.randn(N, Kd, generator=g) / Kd ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    u = torch.randn(N, generator=g)
    wd = torch.float32 if mode == "fp32" else torch.bfloat16
    tol = 2e-5 if mode == "fp32" else 3e-2
    Ad, Wd = A.to(dev), W.to(dev).to(wd)
    # plain + bias + silu prologue + residual
    ref = torch.nn.functional.silu(A) @ W.T + b + R
    out, _ = K.gemm(Ad, Wd, bias=b.to(dev), act_pro=K.ACT_SILU, residual=R.to(dev))
    assert (out.cpu() - ref).abs().max() <= tol * max(1.0, ref.abs().max())
    # relu epilogue + rowdot, no C
    pre = A @ W.T + b
    ref_dot = (torch.relu(pre) * u).sum(1)
    _, part = K.gemm(Ad, Wd, bias=b.to(dev), act_epi=K.ACT_RELU, dotvec=u.to(dev), want_out=False)
    assert (part.sum(1).cpu() - ref_dot).abs().max() <= tol * max(1.0, ref_dot.abs().max()) * 4
    # bf16 output + K-split operand
    if Kd % 64 == 0:
        out, _ = K.gemm(Ad[:, :Kd // 2].contiguous(), Wd, A2=Ad[:, Kd // 2:].contiguous(), out_dtype=torch.bfloat16)
        assert (out.float().cpu() - A @ W.T).abs().max() <= 3e-2 * max(1.0, (A @ W.T).abs().max())


def test_gemm_transpose_detecting():
    from fabind_amd import kernels as K
    dev = _dev()
    A = torch.eye(64, 64)
    W = torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64) / 100.0   # asymmetric
    out, _ = K.gemm(A.to(dev), W.to(dev))
    assert torch.equal(out.cpu(), W.T.contiguous())


def test_segment_sum_and_scan():
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    deg = torch.randint(0, 40, (3000,), generator=g, dtype=torch.int32)
    deg[5] = 1700
    rp = K.exclusive_scan(deg.to(dev))
    ref_rp = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(deg.long(), 0)])
    assert torch.equal(rp.cpu().long(), ref_rp)
    E = int(ref_rp[-1])
    for H in (32, 128, 512):
        Z = torch.randn(E, H, generator=g)
        out = K.segment_sum(Z.to(dev), rp, 3000, K.ACT_SILU)
        row = torch.repeat_interleave(torch.arange(3000), deg.long())
        ref = torch.zeros(3000, H).index_add_(0, row, torch.nn.functional.silu(Z))
        assert (out.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max())
    # bf16 rows take the edge-balanced path (64 positions per wave, cut rows finish with atomics); also the
    # permuted reduction (eidx) used for the sending side, on a buffer that holds stale values before the call
    deg[:7] = 0
    deg[2990:] = 0
    rp = K.exclusive_scan(deg.to(dev))
    row = torch.repeat_interleave(torch.arange(3000), deg.long())
    E = row.shape[0]
    for H in (64, 512, 1024):
        Z = torch.randn(E, H, generator=g).bfloat16()
        out = torch.full((3000, H), 7.0, device=dev)
        K.segment_sum(Z.to(dev), rp, 3000, K.ACT_SILU, out=out)
        ref = torch.zeros(3000, H).index_add_(0, row, torch.nn.functional.silu(Z.float()))
        assert (out.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max())
        perm = torch.randperm(E, generator=g)
        out = K.segment_sum(Z.to(dev), rp, 3000, eidx=perm.to(torch.int32).to(dev))
        ref = torch.zeros(3000, H).index_add_(0, row, Z.float()[perm])
        assert (out.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("bm,p_drop", [(64, 0.0), (64, 0.25), (5, 0.0), (5, 0.25), (6, 0.0), (6, 0.25)])
@pytest.mark.parametrize("H", [64, 128, 256, 512])
def test_fused_edge_backward_matches_autograd(H, bm, p_drop):
    """csrc/fused_edge.hip backward (recompute + 4 chained contractions) vs torch autograd of the same bf16-rounded
    forward on the CPU: every gradient the kernel produces, including the partial column sums and both halves of dAB.
    bm = 64: the two-LDS-tile kernel of round 1 (variant 0: the default for H <= 128); bm = 5: the row-wise, operand-swapped kernel with a
    store wave (csrc/fused_edge_bwd3.hip; the recompute default above 128), 64 edges per tile (variants 1-4 and the 32-edge tile were
    knob-only and were retired in round 5);
    bm = 6: the two-contraction kernel over what the SAVING forward left (csrc/fused_edge_bwd4.hip: M, silu'(pre2) x keep, pre3 from
    fabind_gcl_edge_fused_train -- no recompute; the dropout mask reaches the backward through the saved tiles only)."""
    from fabind_amd import kernels as K, _lib
    dev = _dev()
    _lib.load().fabind_gcl_edge_fused_bwd_set_variant(5 if bm in (5, 6) else 0)
    _lib.load().fabind_gcl_edge_fused_bwd_set_tile(64)
    g = torch.Generator().manual_seed(100 + H)
    N = 300
    deg = torch.randint(0, 12, (N,), generator=g)
    deg[7] = 333
    deg[250:] = 0
    row = torch.repeat_interleave(torch.arange(N), deg)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)]).to(torch.int32)
    E = row.shape[0]
    col = torch.randint(0, N, (E,), generator=g)
    AB16 = torch.randn(N, 2 * H, generator=g).bfloat16()
    rh = torch.rand(E, generator=g)
    w_r, b2, bc, w3 = [torch.randn(H, generator=g) * 0.5 for _ in range(4)]
    W2 = (torch.randn(H, H, generator=g) / H ** 0.5).bfloat16()
    Wc = (torch.randn(H, H, generator=g) / H ** 0.5).bfloat16()
    ds = torch.randn(E, generator=g)
    dagg = torch.randn(N, H, generator=g)
    silu = torch.nn.functional.silu
    leaf = lambda t: t.float().clone().requires_grad_(True)
    ABf, rhf, wrf, W2f, b2f, Wcf, bcf, w3f = map(leaf, (AB16, rh, w_r, W2, b2, Wc, bc, w3))
    rb = lambda t: t + (t.bfloat16().float() - t).detach()           # bf16 rounding, straight-through
    from helpers import fused_edge_keep_mask
    seed = 123457
    keep = fused_edge_keep_mask(seed, E, H, p_drop)
    if p_drop > 0:
        assert abs(float((keep == 0).float().mean()) - p_drop) < 0.02      # Bernoulli(p) zeros, survivors scaled 1/(1-p)
    S1 = rb(silu(ABf[row, :H] + ABf[col, H:] + rhf[:, None] * wrf))
    M = rb(silu(S1 @ W2f.T + b2f) * keep)
    agg = torch.zeros(N, H).index_add(0, row, M)
    sv = (silu(M @ Wcf.T + bcf) * w3f).sum(1)
    ((agg * dagg).sum() + (sv * ds).sum()).backward()
    i32 = lambda t: t.to(torch.int32).to(dev)
    colsorted, perm = torch.sort(col, stable=True)
    colptr = torch.zeros(N + 1, dtype=torch.int32)
    colptr[1:] = torch.cumsum(torch.bincount(colsorted, minlength=N), 0)
    saved = None
    if bm == 6:
        fw = K.gcl_edge_fused(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), K.pack_frag(W2.to(dev)),
                              b2.to(dev), K.pack_frag(Wc.to(dev)), bc.to(dev), w3.to(dev), N, p_drop, seed, save=True)
        saved = fw[-1]
        assert len(saved) == 3 and saved[0].shape == (E, H) and saved[1].shape == ((E + 63) // 64 * 64, H)
        assert (saved[0].float().cpu() - M.detach()).abs().max() <= 2e-2 * max(1.0, float(M.abs().max()))      # the messages, row-major
        fw0 = K.gcl_edge_fused(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), K.pack_frag(W2.to(dev)),
                               b2.to(dev), K.pack_frag(Wc.to(dev)), bc.to(dev), w3.to(dev), N, p_drop, seed)
        assert torch.equal(fw[0], fw0[0]) and torch.equal(fw[1], fw0[1])                                       # same outputs as the plain forward
    out = K.gcl_edge_fused_bwd(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev),
                               Wc.to(dev), bc.to(dev), w3.to(dev), ds.to(dev), dagg.to(dev), colptr.to(dev), i32(perm),
                               p_drop, seed, saved=saved)
    if bm == 6:     # twice: bit-identical (no float atomics); and the bf16 dAB form
        out2 = K.gcl_edge_fused_bwd(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev),
                                    Wc.to(dev), bc.to(dev), w3.to(dev), ds.to(dev), dagg.to(dev), colptr.to(dev), i32(perm),
                                    p_drop, seed, saved=saved)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(out, out2))
        out16 = K.gcl_edge_fused_bwd(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev),
                                     Wc.to(dev), bc.to(dev), w3.to(dev), ds.to(dev), dagg.to(dev), colptr.to(dev), i32(perm),
                                     p_drop, seed, dab_bf16=True, rowptr=rowptr.to(dev), saved=saved)
        assert out16[0].dtype == torch.bfloat16 and torch.equal(out16[0].cpu(), out[0].cpu().bfloat16())
    if bm == 5:     # the default form can write both halves of dAB as bf16 itself (scan / fix-up / segment_sum out16): same values, rounded
        out16 = K.gcl_edge_fused_bwd(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev),
                                     Wc.to(dev), bc.to(dev), w3.to(dev), ds.to(dev), dagg.to(dev), colptr.to(dev), i32(perm),
                                     p_drop, seed, dab_bf16=True, rowptr=rowptr.to(dev))
        assert out16[0].dtype == torch.bfloat16 and torch.equal(out16[0].cpu(), out[0].cpu().bfloat16())
    # the forward kernel evaluates the same mask
    agg_k, s_k = K.gcl_edge_fused(AB16.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), K.pack_frag(W2.to(dev)),
                                  b2.to(dev), K.pack_frag(Wc.to(dev)), bc.to(dev), w3.to(dev), N, p_drop, seed)
    assert (agg_k.cpu() - agg.detach()).abs().max() <= 2e-2 * max(1.0, float(agg.abs().max()))
    assert (s_k[:, 0].cpu() - sv.detach()).abs().max() <= 2e-2 * max(1.0, float(sv.abs().max()))
    _lib.load().fabind_gcl_edge_fused_bwd_set_tile(64)
    _lib.load().fabind_gcl_edge_fused_bwd_set_variant(5)
    names = ("dAB", "drh", "dw_r", "dW2", "db2", "dWc", "dbc", "dw3")
    refs = (ABf.grad, rhf.grad, wrf.grad, W2f.grad, b2f.grad, Wcf.grad, bcf.grad, w3f.grad)
    for name, got, ref in zip(names, out, refs):
        err = (got.float().cpu() - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        assert err <= 3e-2 * scale, (name, err, scale)
        rel = ((got.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-6)).item()
        assert rel <= 2e-2, (name, rel)


@pytest.mark.parametrize("H", [36, 64, 128, 512, 640])
@pytest.mark.parametrize("ab_dtype", [torch.float32, torch.bfloat16])
def test_gcl_pre_gather(H, ab_dtype):
    """First edge Linear evaluated per node and gathered (egnn.py:72-76 split column-wise), with the SiLU and its
    derivative emitted by the producer; both the 8-column vector kernel (H = 64..512) and the generic one."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H)
    N, E = 211, 3001                                    # E not a multiple of the edges-per-wave of any variant
    AB = torch.randn(N, 2 * H, generator=g).to(ab_dtype)
    row = torch.sort(torch.randint(0, N, (E,), generator=g))[0]
    col = torch.randint(0, N, (E,), generator=g)
    rh, w_r = torch.rand(E, generator=g), torch.randn(H, generator=g)
    z = (AB[row, :H].float() + AB[col, H:].float() + rh[:, None] * w_r).requires_grad_(True)
    ref = torch.nn.functional.silu(z)
    dref, = torch.autograd.grad(ref.sum(), z)
    i32 = lambda t: t.to(torch.int32).to(dev)
    for out_dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        pre, dact = K.gcl_pre(AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), out_dtype, K.ACT_SILU,
                              want_dact=True)
        assert (pre.float().cpu() - ref.detach()).abs().max() <= tol * max(1.0, float(ref.abs().max()))
        assert (dact.float().cpu() - dref).abs().max() <= tol * 2
        pre0 = K.gcl_pre(AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), out_dtype)
        assert (pre0.float().cpu() - z.detach()).abs().max() <= tol * max(1.0, float(z.abs().max()))


def test_layernorm():
    from fabind_amd import ops
    dev = _dev()
    x = torch.randn(77, 512)
    w, b = torch.randn(512), torch.randn(512)
    y = ops.layernorm(x.to(dev), w.to(dev), b.to(dev))
    ref = torch.nn.functional.layer_norm(x, (512,), w, b)
    assert (y.cpu() - ref).abs().max() < 1e-4


@pytest.mark.parametrize("cfg", [0, 3, 6, 7, 13])
@pytest.mark.parametrize("shape", [(1000, 512, 512), (257, 200, 128), (4096, 1024, 576), (130, 64, 64)])
def test_gemm_bf16_lds_dma_fast_path(shape, cfg):
    """bf16 A and W, K % 64 == 0 -> global_load_lds pipelined kernels (swizzled LDS image), every tile config."""
    from fabind_amd import kernels as K, _lib
    dev = _dev()
    _lib.load().fabind_gemm_set_config(cfg)
    M, N, Kd = shape
    g = torch.Generator().manual_seed(M + N + Kd)
    A = torch.randn(M, Kd, generator=g).bfloat16()
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16()
    b = torch.randn(N, generator=g)
    u = torch.randn(N, generator=g)
    ref = torch.nn.functional.silu(A.float() @ W.float().T + b)
    out, part = K.gemm(A.to(dev), W.to(dev), bias=b.to(dev), act_epi=K.ACT_SILU, dotvec=u.to(dev), out_dtype=torch.bfloat16)
    assert (out.float().cpu() - ref).abs().max() <= 2e-2 * max(1.0, ref.abs().max())
    rd = (ref * u).sum(1)
    assert (part.sum(1).cpu() - rd).abs().max() <= 2e-2 * max(1.0, rd.abs().max())
    # transposition check with an asymmetric operand
    eye = torch.eye(128, 128).bfloat16()
    Wa = (torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) % 251 / 256.0).bfloat16()
    o2, _ = K.gemm(eye.to(dev), Wa.to(dev))
    o3, _ = K.gemm(eye.to(dev), Wa.to(dev), out_dtype=torch.bfloat16)
    _lib.load().fabind_gemm_set_config(K.GEMM_DEFAULT_CONFIG)
    assert torch.equal(o2.cpu(), Wa.float().T.contiguous())
    assert torch.equal(o3.float().cpu(), Wa.float().T.contiguous())


@pytest.mark.parametrize("variant", [1])
@pytest.mark.parametrize("H", [64, 128, 512])
def test_fused_edge_pipeline_matches_unfused(H, variant):
    """csrc/fused_edge_fwd2.hip (gather -> GEMM -> SiLU -> {segment-sum, GEMM -> row-dot}) vs fp32 torch on the CPU;
    variant 1 = the row-wise / operand-swapped form (the only one built since round 5), which also emits the bf16 copy of agg."""
    from fabind_amd import kernels as K, _lib
    dev = _dev()
    _lib.load().fabind_gcl_edge_fused_set_variant(variant)
    g = torch.Generator().manual_seed(H)
    N, deg = 300, torch.randint(0, 40, (300,), generator=g)
    deg[7] = 333                                                        # a heavy row spanning several 64-edge tiles
    row = torch.repeat_interleave(torch.arange(N), deg)
    E = row.shape[0]
    col = torch.randint(0, N, (E,), generator=g)
    AB = torch.randn(N, 2 * H, generator=g).bfloat16()
    rh = torch.rand(E, generator=g)
    w_r, b2, bc, w3 = [torch.randn(H, generator=g) * 0.5 for _ in range(4)]
    W2 = (torch.randn(H, H, generator=g) / H ** 0.5).bfloat16()
    Wc = (torch.randn(H, H, generator=g) / H ** 0.5).bfloat16()
    silu = torch.nn.functional.silu
    S1 = silu(AB[row, :H].float() + AB[col, H:].float() + rh[:, None] * w_r).bfloat16().float()
    M = silu(S1 @ W2.float().T + b2).bfloat16().float()
    agg_ref = torch.zeros(N, H).index_add_(0, row, M)
    s_ref = (silu(M @ Wc.float().T + bc) * w3).sum(1)
    i32 = lambda t: t.to(torch.int32).to(dev)
    agg, s, agg16 = K.gcl_edge_fused(AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), K.pack_frag(W2.to(dev)),
                                     b2.to(dev), K.pack_frag(Wc.to(dev)), bc.to(dev), w3.to(dev), N, want16=True,
                                     rowptr=torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)]).to(torch.int32).to(dev))
    _lib.load().fabind_gcl_edge_fused_set_variant(1)
    assert (agg.cpu() - agg_ref).abs().max() <= 2e-2 * max(1.0, float(agg_ref.abs().max()))
    assert (s[:, 0].cpu() - s_ref).abs().max() <= 2e-2 * max(1.0, float(s_ref.abs().max()))
    assert torch.equal(agg16.cpu(), agg.cpu().bfloat16())                # the bf16 copy is the rounded fp32 result, every row


@pytest.mark.parametrize("shape", [(5000, 512, 512), (777, 256, 128), (100000, 512, 1024), (33, 64, 64)])
def test_gemm_tn_transpose_reads(shape):
    """Weight-gradient contraction sum_e Y[e,m] X[e,n] with ds_read_b64_tr_b16 fragments (no transposes)."""
    from fabind_amd import kernels as K
    dev = _dev()
    E, M, N = shape
    g = torch.Generator().manual_seed(E)
    Y = torch.randn(E, M, generator=g).bfloat16()
    X = torch.randn(E, N, generator=g).bfloat16()
    ref = Y.double().T @ X.double()
    out = K.gemm_tn(Y.to(dev), X.to(dev))
    assert (out.cpu().double() - ref).abs().max() <= 2e-3 * float(ref.abs().max()) + 1e-3 * E ** 0.5
    # strided views (column slices of wider tensors) and a forced odd split count
    Yw = torch.randn(E, M + 64, generator=g).bfloat16().to(dev)
    out2 = K.gemm_tn(Yw[:, 64:], X.to(dev), splits=3)
    ref2 = Yw[:, 64:].cpu().double().T @ X.double()
    assert (out2.cpu().double() - ref2).abs().max() <= 2e-3 * float(ref2.abs().max()) + 1e-3 * E ** 0.5


def test_pair_bias_adjoint_gemm_path_matches_valu_path():
    """ops.pair_bias backward: ragged-batched MFMA formulation (bf16 mode) vs the fp32 VALU kernels vs torch autograd."""
    from fabind_amd import engine, ops, synthetic
    dev = _dev()
    H = 64
    inp = synthetic.make_stack_batch([(70, 9), (33, 17), (120, 5)], H, seed=3)
    lay = engine.Layout(inp["batch_id"].to(dev), inp["segment_id"].to(dev))
    g = torch.Generator().manual_seed(1)
    a0b0 = torch.randn(lay.N, 2 * H, generator=g)
    wcomp = torch.randn(4, 8, H, generator=g) / H ** 0.5
    bconst = torch.randn(4, 8, generator=g)
    cots = [torch.randn(lay.n_pairs, 8, generator=g) for _ in range(4)]
    # torch reference on the CPU
    a_ref = a0b0.clone().requires_grad_(True); w_ref = wcomp.clone().requires_grad_(True); b_ref = bconst.clone().requires_grad_(True)
    loss = 0
    pi = lay.p_index64.cpu(); ci = lay.c_index64.cpu()
    po, co = 0, 0
    for b in range(lay.B):
        P_, C_ = int(lay.P[b]), int(lay.C[b])
        a = a_ref[pi[po:po + P_], :H]; bb = a_ref[ci[co:co + C_], H:]
        for k in range(4):
            out = torch.einsum("ih,jh,oh->ijo", a, bb, w_ref[k]) + b_ref[k]
            off = int(lay.pair_off_np[b])
            loss = loss + (out.reshape(-1, 8) * cots[k][off:off + P_ * C_]).sum()
        po += P_; co += C_
    loss.backward()
    res = {}
    for mode in ("fp32", "bf16"):
        engine.set_precision(mode)
        a = a0b0.to(dev).requires_grad_(True); w = wcomp.to(dev).requires_grad_(True); bc_ = bconst.to(dev).requires_grad_(True)
        outs = ops.pair_bias(a, H, w, bc_, lay)
        sum((o * c.to(dev)).sum() for o, c in zip(outs, cots)).backward()
        res[mode] = (a.grad.cpu(), w.grad.cpu(), bc_.grad.cpu())
    engine.set_precision("fp32")
    for mode, tol in (("fp32", 2e-4), ("bf16", 3e-2)):
        for got, ref in zip(res[mode], (a_ref.grad, w_ref.grad, b_ref.grad)):
            assert (got - ref).abs().max() <= tol * float(ref.abs().max()), mode


def test_gemm_epilogue_dropout_statistics_and_order():
    """FabindGemmArgs.p_drop: Bernoulli(p) zeros on act(x W^T + b), survivors scaled by 1/(1-p), applied BEFORE the residual;
    the same seed reproduces the mask, the row-dot epilogue sees the dropped values."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    M, N, Kd, p = 3000, 256, 128, 0.25
    A = torch.randn(M, Kd, generator=g).bfloat16().to(dev)
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16().to(dev)
    b = (torch.rand(N, generator=g) + 3.0).to(dev)                       # keeps relu(.) > 0: zeros come from dropout only
    R = torch.randn(M, N, generator=g).to(dev)
    base, _ = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU)
    y1, _ = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, residual=R, p_drop=p, seed=77)
    y2, _ = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, residual=R, p_drop=p, seed=77)
    y3, _ = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, residual=R, p_drop=p, seed=78)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    v = y1 - R                                                          # dropped rows equal the residual exactly
    dropped = v.abs() < 1e-6
    assert abs(float(dropped.float().mean()) - p) < 0.01
    thr = round(p * 65536) / 65536.0
    kept = ~dropped
    assert ((v[kept] - base[kept] / (1 - thr)).abs() <= 1e-3 * base[kept].abs() + 1e-3).all()
    u = torch.randn(N, generator=g).to(dev)
    _, part = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, dotvec=u, want_out=False, p_drop=p, seed=77)
    ref = ((y1 - R) * u).sum(1)
    assert (part.sum(1) - ref).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
    # the staged bf16 epilogue (fast path) draws the same mask as the generic fp32 one
    y16, _ = K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, out_dtype=torch.bfloat16, p_drop=p, seed=77)
    assert torch.equal(y16 == 0, dropped)
    assert ((y16.float() - v).abs() <= 1e-2 * v.abs() + 1e-3).all()


@pytest.mark.parametrize("H,n_pairs,p_drop", [(512, 1000, 0.0), (512, 4133, 0.1), (128, 257, 0.0), (64, 63, 0.25), (256, 640, 0.1)])
def test_pair_update_fused_matches_torch(H, n_pairs, p_drop):
    """fabind_pair_update_fused (FABind+ cross_att.py:42-44): Hadamard + linear_out + residual + LayerNorm + the two
    transition Linears (+ the next layer's 16 bias columns) per 64-pair tile, against fp32 torch on the same bf16-rounded
    operands; the dropout masks are the counter-based ones of include/fabind_hip.h."""
    from fabind_amd import kernels as K
    from helpers import fused_edge_keep_mask
    dev = _dev()
    g = torch.Generator().manual_seed(H + n_pairs)
    n_nodes = 200
    T = torch.randn(n_nodes, 128, generator=g)
    pn = torch.randint(0, n_nodes, (n_pairs,), generator=g, dtype=torch.int32)
    cn = torch.randint(0, n_nodes, (n_pairs,), generator=g, dtype=torch.int32)
    z = torch.randn(n_pairs, H, generator=g).bfloat16()
    r16 = lambda t: t.bfloat16().float()
    Wo = r16(torch.randn(H, 32, generator=g) / 32 ** 0.5)
    W1, W2 = r16(torch.randn(H, H, generator=g) / H ** 0.5), r16(torch.randn(H, H, generator=g) / H ** 0.5)
    Wb = r16(torch.randn(16, H, generator=g) / H ** 0.5)
    bo, b1, b2, bb = (torch.randn(n, generator=g) * 0.3 for n in (H, H, H, 16))
    lw, lb = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.2
    seed = 1234
    # fp32 reference (bf16 rounding at the points where the kernel stores bf16 tiles)
    hd = r16(T[pn.long(), :32] * T[cn.long(), 64:96])
    z1 = z.float() + hd @ Wo.T + bo
    y = r16(torch.nn.functional.layer_norm(z1, (H,), lw, lb, 1e-5))
    t = torch.relu(y @ W1.T + b1)
    if p_drop > 0:
        t = t * fused_edge_keep_mask(seed, n_pairs, H, p_drop)
    t = r16(t)
    zn = torch.relu(t @ W2.T + b2)
    if p_drop > 0:
        zn = zn * fused_edge_keep_mask((seed + 0x9e3779b9) & 0xFFFFFFFF, n_pairs, H, p_drop)
    zn = r16(zn)
    bias_ref = zn @ Wb.T + bb
    d = lambda t_: t_.to(dev)
    z_out, bias = K.pair_update_fused(d(T), 64, d(pn), d(cn), d(z), K.pack_frag(d(Wo)), d(bo), d(lw), d(lb), 1e-5,
                                      K.pack_frag(d(W1)), d(b1), K.pack_frag(d(W2)), d(b2), K.pack_frag(d(Wb)), d(bb), p_drop, seed)
    torch.cuda.synchronize()
    scale = max(1.0, float(zn.abs().max()))
    err = (z_out.float().cpu() - zn).abs()
    # bf16 tiles between the contractions: a rounding flip moves an element by one bf16 ulp of the intermediate
    assert float(err.max()) <= 4e-2 * scale, float(err.max())
    assert float(err.mean()) <= 2e-3 * scale
    assert (bias.cpu() - bias_ref).abs().max() <= 4e-2 * max(1.0, float(bias_ref.abs().max()))
    if p_drop > 0:
        assert abs(float((z_out == 0).float().mean()) - float((zn == 0).float().mean())) < 5e-3
    # without the bias projection, and in place
    z_in = d(z).clone()
    a = K.pair_update_fused(d(T), 64, d(pn), d(cn), z_in, K.pack_frag(d(Wo)), d(bo), d(lw), d(lb), 1e-5,
                            K.pack_frag(d(W1)), d(b1), K.pack_frag(d(W2)), d(b2), None, None, p_drop, seed)
    assert a[1] is None and torch.equal(a[0], z_out)


@pytest.mark.parametrize("H,E,p_drop", [(512, 3000, 0.0), (128, 777, 0.0), (64, 500, 0.2)])
def test_edge_lnfold_matches_layernorm_then_linear(H, E, p_drop):
    """fabind_edge_lnfold: relu(W1 LN([h_r | h_c | rho]) + b1) from per-node projections + per-node statistics equals
    the straightforward LayerNorm -> Linear -> relu on the concatenated edge input (egnn.py:52-58 of FABind+)."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H + E)
    N, Cn = 300, 2 * H + 1
    Kp = (Cn + 63) // 64 * 64
    h = torch.randn(N, H, generator=g) * 1.5 + 4.0                      # large feature mean: the centred form must not cancel
    row = torch.randint(0, N, (E,), generator=g, dtype=torch.int32)
    col = torch.randint(0, N, (E,), generator=g, dtype=torch.int32)
    rho = torch.rand(E, generator=g) * 3.0
    lw, lb = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    W1, b1 = torch.randn(Cn, Cn, generator=g) / Cn ** 0.5, torch.randn(Cn, generator=g) * 0.3
    x = torch.cat([h[row.long()], h[col.long()], rho[:, None]], 1)
    ref = torch.relu(torch.nn.functional.layer_norm(x, (Cn,), lw, lb, 1e-5) @ W1.T + b1)         # [E, Cn]
    W1w = torch.zeros(Kp, Cn); W1w[:Cn] = W1 * lw[None, :]
    mean = h.mean(1)
    hcen = h - mean[:, None]
    AB = (hcen @ torch.cat([W1w[:, :H], W1w[:, H:2 * H]]).T).bfloat16()                          # [N, 2 Kp]
    stat = torch.stack([mean, (hcen * hcen).sum(1)], 1).contiguous()
    dvec = torch.zeros(Kp); dvec[:Cn] = W1 @ lb + b1
    d = lambda t: t.to(dev)
    out = K.edge_lnfold(d(AB), Kp, H, d(row), d(col), d(rho), d(stat), 1e-5, d(W1w[:, 2 * H].contiguous()),
                        d(W1w[:, :H].sum(1).contiguous()), d(W1w[:, H:2 * H].sum(1).contiguous()), d(dvec), p_drop, 99).float().cpu()
    assert out.shape == (E, Kp) and float(out[:, Cn:].abs().max()) == 0.0
    if p_drop == 0.0:
        assert (out[:, :Cn] - ref).abs().max() <= 3e-2 * max(1.0, float(ref.abs().max()))
        assert (out[:, :Cn] - ref).abs().mean() <= 3e-3 * max(1.0, float(ref.abs().mean()))
    else:
        kept = out[:, :Cn] != 0
        thr = round(p_drop * 65536) / 65536.0
        live = ref > 1e-3
        assert abs(float((~kept)[live].float().mean()) - p_drop) < 0.01
        assert ((out[:, :Cn] * (1 - thr) - ref)[kept].abs() <= 3e-2 * max(1.0, float(ref.abs().max()))).all()


@pytest.mark.parametrize("H,E,p_drop", [(512, 3000, 0.0), (64, 500, 0.0), (128, 900, 0.2)])
def test_inter_coord_fold_matches_layernorm_mlp(H, E, p_drop):
    """fabind_inter_coord_fold: w3 . relu(W1 LN(V[col] + rho w_rv) + b1) from the per-node projection equals the
    straightforward LayerNorm -> Linear -> relu -> dot on the per-edge value rows (FABind+ egnn.py:277-300)."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H * 3 + E)
    N = 250
    V = torch.randn(N, H, generator=g) * 1.2 + 2.0
    w_rv = torch.randn(H, generator=g) * 0.5 + 0.3
    col = torch.randint(0, N, (E,), generator=g, dtype=torch.int32)
    rho = torch.rand(E, generator=g) * 3.0
    lw, lb = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.2
    W1, b1, w3 = torch.randn(H, H, generator=g) / H ** 0.5, torch.randn(H, generator=g) * 0.3, torch.randn(H, generator=g)
    v_e = V[col.long()] + rho[:, None] * w_rv
    act = torch.relu(torch.nn.functional.layer_norm(v_e, (H,), lw, lb, 1e-5) @ W1.T + b1)
    W1w = W1 * lw[None, :]
    Vc, wc = V - V.mean(1, keepdim=True), w_rv - w_rv.mean()
    stat = torch.stack([(Vc * Vc).sum(1), Vc @ wc], 1).contiguous()
    d = lambda t: t.to(dev)
    seed = 4242
    s = K.inter_coord_fold(d((Vc @ W1w.T).bfloat16()), H, d(col), d(rho), d(stat), float((wc * wc).sum()), 1e-5, d(W1w @ wc),
                           d(W1 @ lb + b1), d(w3), p_drop, seed).cpu()
    if p_drop > 0:
        thr = int(p_drop * 65536.0 + 0.5)
        e = torch.arange(E, dtype=torch.int64)[:, None]; c = torch.arange(H, dtype=torch.int64)[None, :]
        x = (seed + e * H + c) & 0xFFFFFFFF
        x = x ^ (x >> 16); x = (x * 0x7feb352d) & 0xFFFFFFFF
        x = x ^ (x >> 15); x = (x * 0x846ca68b) & 0xFFFFFFFF
        x = x ^ (x >> 16)
        act = act * ((x & 0xFFFF) >= thr).float() / (1.0 - thr / 65536.0)
    ref = act @ w3
    assert (s - ref).abs().max() <= 3e-2 * max(1.0, float(ref.abs().max()))
    assert (s - ref).abs().mean() <= 5e-3 * max(1.0, float(ref.abs().mean()))


@pytest.mark.parametrize("M,N,Kd,p_drop", [(3000, 512, 512, 0.0), (777, 256, 128, 0.0), (2048, 512, 512, 0.2)])
def test_gemm_layernorm_fold_epilogue(M, N, Kd, p_drop):
    """FabindGemmArgs.row_mu / row_rs / col_c: w3 . relu(W1 LN(x) + b1) evaluated as a GEMM on the un-normalised bf16 rows
    with the LayerNorm folded into the epilogue equals LayerNorm -> Linear -> relu -> dot (FABind+ coord_mlp on the
    messages, egnn.py:104-118); fabind_row_stats gives the row statistics."""
    from fabind_amd import kernels as K, ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, Kd, generator=g).abs() * 1.3 + 0.5).bfloat16()            # post-relu-like rows: large positive mean
    lw, lb = torch.rand(Kd, generator=g) + 0.5, torch.randn(Kd, generator=g) * 0.2
    W1, b1, w3 = torch.randn(N, Kd, generator=g) / Kd ** 0.5, torch.randn(N, generator=g) * 0.3, torch.randn(N, generator=g)
    xf = x.float()
    mu_ref, var = xf.mean(1), xf.var(1, unbiased=False)
    W1w = (W1 * lw[None, :]).bfloat16()
    cvec = W1w.float().sum(1)
    act = torch.relu(((xf - mu_ref[:, None]) * torch.rsqrt(var + 1e-5)[:, None]) @ W1w.float().T + (W1 @ lb + b1))
    d = lambda t: t.to(dev)
    mu, rs = K.row_stats(d(x), 1e-5)
    assert (mu.cpu() - mu_ref).abs().max() < 1e-5 and (rs.cpu() - torch.rsqrt(var + 1e-5)).abs().max() < 1e-4
    import torch as _t
    _t.manual_seed(3)
    part = ops.linear_rowdot(d(x), d(W1w), d(W1 @ lb + b1), d(w3), act_epi=K.ACT_RELU, p_drop=p_drop, fold=(mu, rs, d(cvec)))
    s = part.sum(1).cpu()
    if p_drop == 0.0:
        ref = act @ w3
        assert (s - ref).abs().max() <= 3e-2 * max(1.0, float(ref.abs().max()))
        assert (s - ref).abs().mean() <= 3e-3 * max(1.0, float(ref.abs().mean()))
    else:                                   # mask drawn inside the epilogue: statistics only
        ref = act @ w3
        assert abs(float(s.mean()) - float(ref.mean())) < 0.15 * float(ref.std()) and torch.isfinite(s).all()


def test_gemm_tn_work_group_layouts_are_bitwise_equal():
    """fabind_gemm_tn has four work-group layouts (16 = 256x256 tile on 8 waves with a 4-slot ring, the default; 20 = 5 slots;
    4 = 256x128 on 4 waves; 8 = 256x128 on 8 waves): same k order per output element, so for the SAME split count the results must be
    identical bit for bit -- on a ragged E, with M, N that do not fill a tile -- and so must the column sums that ride along."""
    from fabind_amd import _lib, kernels as K
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    try:
        for E, M, N, sp in ((4096 + 40, 512, 512, 8), (1000, 328, 136, 3), (70000, 512, 8, 16), (33000, 1024, 576, 8)):
            Y = torch.randn(E, M, device=dev).to(torch.bfloat16)
            X = torch.randn(E, N, device=dev).to(torch.bfloat16)
            ref = Y.float().t() @ X.float()
            outs = []
            for w in (16, 20, 4, 8):
                lib.fabind_gemm_tn_set_waves(w)
                outs.append(K.gemm_tn(Y, X, splits=sp, with_colsum=True))
            for o in outs[1:]:
                assert torch.equal(outs[0][0], o[0]), (E, M, N)
                assert torch.equal(outs[0][1], o[1]), (E, M, N)
            assert float((outs[0][0] - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), (E, M, N)
    finally:
        lib.fabind_gemm_tn_set_waves(16)


@pytest.mark.parametrize("shape", [(98688 // 8, 512, 512), (1000, 328, 136), (5000, 1536, 512), (777, 8, 512), (40000, 1024, 576)])
def test_gemm_tn_column_sums_ride_along(shape):
    """fabind_gemm_tn(with_colsum): the bias gradient dY^T 1 from the same launches as the weight gradient dY^T X (fp32 sums of the
    bf16 operand; one split reduction over M*N + M elements) -- against float64 sums, and the weight part against the plain call."""
    from fabind_amd import kernels as K
    dev = _dev()
    E, M, N = shape
    g = torch.Generator().manual_seed(E + M)
    Y = (torch.randn(E, M, generator=g) * 0.3 + 0.05).bfloat16().to(dev)
    X = torch.randn(E, N, generator=g).bfloat16().to(dev)
    dW, db = K.gemm_tn(Y, X, with_colsum=True)
    assert dW.shape == (M, N) and db.shape == (M,)
    assert torch.equal(dW, K.gemm_tn(Y, X))
    ref = Y.double().sum(0)
    assert float((db.double() - ref).abs().max()) <= 1e-5 * float(Y.double().abs().sum(0).max()) + 1e-6
    # one split, bf16 result dtype, a column slice as Y
    Yw = (torch.randn(E, M + 64, generator=g) * 0.3).bfloat16().to(dev)
    dW1, db1 = K.gemm_tn(Yw[:, 64:], X, splits=1, with_colsum=True)
    assert float((db1.double() - Yw[:, 64:].double().sum(0)).abs().max()) <= 1e-5 * float(Yw.double().abs().sum(0).max()) + 1e-6
    dWb, dbb = K.gemm_tn(Y, X, out_dtype=torch.bfloat16, with_colsum=True)
    assert dWb.dtype == torch.bfloat16 and float((dbb.double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pair_hadamard_adjoint_row_walk_matches_atomics(dt):
    """Adjoint of the pair Hadamard over the inter graph's reduced pairs: the row walk (one wave per node over its own inter edges,
    one writer per element, fabind_pair_hadamard_bwd_rows) against the float-atomics kernel and against torch autograd; the row walk
    must give the same bits twice, and it must ADD to a gradient buffer that already holds something (shared gradient sink)."""
    from fabind_amd import engine, ops, synthetic
    import fabind_amd.ops as ops_mod
    dev = _dev()
    engine.set_precision("fp32")
    H, H2 = 64, 8
    inp = synthetic.make_stack_batch([(70, 9), (33, 17), (120, 5)], H, seed=5)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    lay = engine.Layout(t["batch_id"], t["segment_id"])
    br, bc_ = t["compound_edge_index"][0].to(torch.int32), t["compound_edge_index"][1].to(torch.int32)
    g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
    n_red, N = int(g.red_p.shape[0]), lay.N
    assert n_red > 50
    gen = torch.Generator().manual_seed(3)
    T0 = torch.randn(N, 2 * H, generator=gen).to(dev)
    T1 = torch.randn(N, 2 * H2, generator=gen).to(dev)
    w = torch.randn(n_red, H + H2, generator=gen).to(dev)

    def run(graph, shared=False):
        t0, t1 = T0.clone().requires_grad_(True), T1.clone().requires_grad_(True)
        base = t0 * 1.0
        s = ops.shared_grad(base) if shared else base
        hd = ops.pair_hadamard(s, H, t1, H2, g.red_p, g.red_c, graph=graph)
        loss = (hd.float() * w).sum()
        if shared:
            loss = loss + (ops.pair_hadamard(s, H, t1, H2, g.red_p, g.red_c, graph=graph).float() * w).sum() * 0.5
        loss.backward()
        return t0.grad, t1.grad

    old = ops_mod.act_dtype
    try:
        ops_mod.act_dtype = lambda: dt                       # the gradient arrives in the activation dtype of the forward output
        a0, a1 = run(g)
        b0, b1 = run(g)
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
        r0, r1 = run(None)                                    # float atomics
        assert float((a0 - r0).abs().max()) <= 1e-5 * float(r0.abs().max())
        assert float((a1 - r1).abs().max()) <= 1e-5 * float(r1.abs().max())
        s0, s1 = run(g, shared=True)
        assert float((s0 - 1.5 * a0).abs().max()) <= 1e-5 * float(a0.abs().max())
        assert float((s1 - 1.5 * a1).abs().max()) <= 1e-5 * float(a1.abs().max())
    finally:
        ops_mod.act_dtype = old
    # torch autograd on the same expression
    t0, t1 = T0.clone().requires_grad_(True), T1.clone().requires_grad_(True)
    p, c = g.red_p.long(), g.red_c.long()
    ref = torch.cat([t0[p, :H] * t0[c, H:], t1[p, :H2] * t1[c, H2:]], 1)
    if dt == torch.float32:
        (ref * w).sum().backward()
        assert float((a0 - t0.grad).abs().max()) <= 1e-5 * float(t0.grad.abs().max())
        assert float((a1 - t1.grad).abs().max()) <= 1e-5 * float(t1.grad.abs().max())


@pytest.mark.parametrize("case", ["one_consumer", "two_consumers", "second_output_unused"])
def test_shared_gradient_buffer_matches_plain_autograd(case):
    """ops.shared_grad / ops.GradSink: consumers with accumulating adjoints add into ONE gradient buffer (the first to run hands it
    to autograd, the others return None).  Whatever subset of the consumers takes part in the backward pass, the gradient must
    equal what plain autograd computes by summing separate per-consumer gradients."""
    from fabind_amd import engine, ops
    engine.set_precision("fp32")
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    N, H, H2, n = 70, 32, 8, 400
    T0 = torch.randn(N, 2 * H, generator=gen).to(dev)
    T1 = torch.randn(N, 2 * H2, generator=gen).to(dev)
    idx = [torch.randint(0, N, (n,), generator=gen).to(torch.int32).to(dev) for _ in range(4)]
    w = [torch.randn(n, H + H2, generator=gen).to(dev) for _ in range(2)]

    def run(shared):
        t0, t1 = T0.clone().requires_grad_(True), T1.clone().requires_grad_(True)
        base = t0 * 1.0                                     # a non-leaf producer, like the Linear in engine.StackContext
        s = ops.shared_grad(base) if shared else base
        a = ops.pair_hadamard(s, H, t1, H2, idx[0], idx[1])
        loss = (a * w[0]).sum()
        if case != "one_consumer":
            b = ops.pair_hadamard(s, H, t1, H2, idx[2], idx[3])
            if case == "two_consumers":
                loss = loss + (b * w[1]).sum()
        loss.backward()
        return t0.grad, t1.grad

    (g0, g1), (r0, r1) = run(True), run(False)
    assert g0 is not None and r0 is not None
    # float atomics inside the adjoint reorder sums: equal to round-off
    assert float((g0 - r0).abs().max()) <= 1e-5 * float(r0.abs().max())
    assert float((g1 - r1).abs().max()) <= 1e-5 * float(r1.abs().max())


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("act", ["relu", "silu"])
@pytest.mark.parametrize("with_x2,with_sink", [(False, False), (True, False), (False, True), (True, True)])
def test_mlp2_node_matches_autograd(prec, tol, act, with_x2, with_sink):
    """ops.mlp2 (Linear -> act -> Linear (+ residual) as one autograd node: activation adjoint inside the GEMM epilogue,
    residual gradient folded into the input-gradient GEMM or x's shared buffer) against plain torch autograd -- forward,
    d x, d x2, all four parameter gradients; with and without a second consumer of x sharing the gradient buffer."""
    from fabind_amd import engine, kernels as K, ops
    dev = _dev()
    engine.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(7)
        M, Hh, K2, N1 = 1000, 128, 64, 256
        x = torch.randn(M, Hh, generator=g)
        x2 = torch.randn(M, K2, generator=g) if with_x2 else None
        W1 = torch.randn(N1, Hh + (K2 if with_x2 else 0), generator=g) / 12
        b1, W2, b2 = torch.randn(N1, generator=g) * 0.1, torch.randn(Hh, N1, generator=g) / 16, torch.randn(Hh, generator=g) * 0.1
        W3 = torch.randn(64, Hh, generator=g) / 12                       # the second consumer of x
        cot = torch.randn(M, Hh, generator=g)
        fa = torch.relu if act == "relu" else torch.nn.functional.silu
        leaves = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2, W3)] + ([x2.clone().requires_grad_(True)] if with_x2 else [])
        xr, W1r, b1r, W2r, b2r, W3r = leaves[:6]
        # bf16 mode: the reference sees the same rounded operands (straight-through), otherwise ReLU masks of entries within
        # 2^-9 of zero differ and single gradient entries move by O(1) of a term
        rb = (lambda t_: t_ + (t_.bfloat16().float() - t_).detach()) if prec == "bf16" else (lambda t_: t_)
        xin = torch.cat([rb(xr), rb(leaves[6])], 1) if with_x2 else rb(xr)
        yr = rb(fa(xin @ rb(W1r).T + b1r)) @ rb(W2r).T + b2r + xr
        extra = (xr @ W3r.T).pow(2).sum() if with_sink else 0.0
        ((yr * cot).sum() + extra).backward()
        wd = ops.mm_dtype()
        dl = [t.clone().to(dev).requires_grad_(True) for t in (x, W1, b1, W2, b2, W3)] + ([x2.clone().to(dev).requires_grad_(True)] if with_x2 else [])
        xd, W1d, b1d, W2d, b2d, W3d = dl[:6]
        xs = ops.shared_grad(xd) if with_sink else xd
        code = K.ACT_RELU if act == "relu" else K.ACT_SILU
        y = ops.mlp2(xs, W1d.to(wd), b1d, code, W2d.to(wd), b2d, residual=xs, x2=dl[6] if with_x2 else None)
        extra = ops.linear(xs, W3d.to(wd)).pow(2).sum() if with_sink else 0.0
        ((y * cot.to(dev)).sum() + extra).backward()
        assert float((y.detach().cpu() - yr.detach()).abs().max()) <= tol * float(yr.abs().max())
        for name, a, b in zip(("x", "W1", "b1", "W2", "b2", "W3", "x2"), dl, leaves):
            if name == "W3" and not with_sink:
                continue
            err = float((a.grad.float().cpu() - b.grad).abs().max() / b.grad.abs().max())
            assert err <= tol, (name, err)
    finally:
        engine.set_precision("fp32")


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
@pytest.mark.parametrize("H", [64, 128, 256, 512])
def test_fused_edge_split_bf16_matches_float64(H, p_drop):
    """csrc/fused_edge_fwd3.hip: the fused forward edge pipeline in split-bf16 arithmetic (fp32 AB rows, hi | lo LDS planes and weight
    packs, three MFMAs per product term) against the same pipeline in float64 -- fp32-grade agreement (1e-5 of the largest entry;
    the bf16 kernel's bound on this test is 2e-2), ragged last tile, a heavy row spanning tiles, rows without edges, dropout mask."""
    from fabind_amd import kernels as K
    from helpers import fused_edge_keep_mask
    dev = _dev()
    g = torch.Generator().manual_seed(H + 1)
    N, deg = 300, torch.randint(0, 40, (300,), generator=g)
    deg[7] = 333
    deg[11] = 0
    row = torch.repeat_interleave(torch.arange(N), deg)
    E = row.shape[0]
    col = torch.randint(0, N, (E,), generator=g)
    AB = torch.randn(N, 2 * H, generator=g)
    rh = torch.rand(E, generator=g)
    w_r, b2, bc, w3 = [torch.randn(H, generator=g) * 0.5 for _ in range(4)]
    W2 = torch.randn(H, H, generator=g) / H ** 0.5
    Wc = torch.randn(H, H, generator=g) / H ** 0.5
    silu = torch.nn.functional.silu
    d = lambda t: t.double()
    S1 = silu(d(AB)[row, :H] + d(AB)[col, H:] + d(rh)[:, None] * d(w_r))
    M = silu(S1 @ d(W2).T + d(b2))
    seed = 1234
    if p_drop > 0.0:
        M = M * fused_edge_keep_mask(seed, E, H, p_drop).double()
    agg_ref = torch.zeros(N, H, dtype=torch.float64).index_add_(0, row, M)
    s_ref = (silu(M @ d(Wc).T + d(bc)) * d(w3)).sum(1)
    i32 = lambda t: t.to(torch.int32).to(dev)
    rp = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)]).to(torch.int32).to(dev)
    agg, s = K.gcl_edge_fused_x3(AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev), Wc.to(dev),
                                 bc.to(dev), w3.to(dev), N, p_drop=p_drop, seed=seed, rowptr=rp)
    e_agg = float((agg.double().cpu() - agg_ref).abs().max() / agg_ref.abs().max())
    e_s = float((s[:, 0].double().cpu() - s_ref).abs().max() / s_ref.abs().max())
    print("fused edge forward, split bf16, H=%d p=%.2f: agg err %.2e, s err %.2e (of the largest entry)" % (H, p_drop, e_agg, e_s))
    assert e_agg <= 1e-5 and e_s <= 1e-5
    assert float(agg[11].abs().max()) == 0.0                            # a node without edges
    agg2, s2 = K.gcl_edge_fused_x3(AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev), Wc.to(dev),
                                   bc.to(dev), w3.to(dev), N, p_drop=p_drop, seed=seed, rowptr=rp)
    assert torch.equal(agg, agg2) and torch.equal(s, s2)                # deterministic


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
@pytest.mark.parametrize("H", [64, 256, 512])
def test_fused_edge_split_bf16_saving_forward_feeds_the_two_contraction_backward(H, p_drop):
    """Round 5: the SAVING form of csrc/fused_edge_fwd3.hip (fabind_gcl_edge_fused_x3_train) -- identical agg / s
    to the plain split-bf16 forward, bf16(M) row-major, silu'(pre2) x keep and pre3 in fragment order -- and the two-contraction
    backward of csrc/fused_edge_bwd4.hip run on what it left + a bf16 copy of AB: every gradient against fp32 torch autograd of the
    same pipeline (bf16-grade bounds: the operands of the backward are bf16 roundings), bit-identical repeats."""
    from fabind_amd import kernels as K
    from helpers import fused_edge_keep_mask
    dev = _dev()
    g = torch.Generator().manual_seed(7 * H + 3)
    N = 300
    deg = torch.randint(0, 12, (N,), generator=g)
    deg[7] = 333
    deg[250:] = 0
    row = torch.repeat_interleave(torch.arange(N), deg)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)]).to(torch.int32)
    E = row.shape[0]
    col = torch.randint(0, N, (E,), generator=g)
    AB = torch.randn(N, 2 * H, generator=g)
    rh = torch.rand(E, generator=g)
    w_r, b2, bc, w3 = [torch.randn(H, generator=g) * 0.5 for _ in range(4)]
    W2 = torch.randn(H, H, generator=g) / H ** 0.5
    Wc = torch.randn(H, H, generator=g) / H ** 0.5
    ds = torch.randn(E, generator=g)
    dagg = torch.randn(N, H, generator=g)
    silu = torch.nn.functional.silu
    leaf = lambda t: t.clone().requires_grad_(True)
    ABf, rhf, wrf, W2f, b2f, Wcf, bcf, w3f = map(leaf, (AB, rh, w_r, W2, b2, Wc, bc, w3))
    seed = 4242
    keep = fused_edge_keep_mask(seed, E, H, p_drop)
    S1 = silu(ABf[row, :H] + ABf[col, H:] + rhf[:, None] * wrf)
    M = silu(S1 @ W2f.T + b2f) * keep
    agg = torch.zeros(N, H).index_add(0, row, M)
    sv = (silu(M @ Wcf.T + bcf) * w3f).sum(1)
    ((agg * dagg).sum() + (sv * ds).sum()).backward()
    i32 = lambda t: t.to(torch.int32).to(dev)
    rp = rowptr.to(dev)
    args = (AB.to(dev), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev), Wc.to(dev), bc.to(dev), w3.to(dev), N)
    a0, s0 = K.gcl_edge_fused_x3(*args, p_drop=p_drop, seed=seed, rowptr=rp)
    a1, s1, saved = K.gcl_edge_fused_x3(*args, p_drop=p_drop, seed=seed, rowptr=rp, save=True)
    assert torch.equal(a0, a1) and torch.equal(s0, s1)                                   # the stores do not change the results
    assert saved[0].dtype == torch.bfloat16 and saved[0].shape == (E, H) and saved[1].shape == ((E + 63) // 64 * 64, H)
    assert (saved[0].float().cpu() - M.detach()).abs().max() <= 1e-2 * max(1.0, float(M.abs().max()))      # bf16(M), row-major
    colsorted, perm = torch.sort(col, stable=True)
    colptr = torch.zeros(N + 1, dtype=torch.int32)
    colptr[1:] = torch.cumsum(torch.bincount(colsorted, minlength=N), 0)
    bw = lambda: K.gcl_edge_fused_bwd(AB.to(dev).bfloat16(), H, i32(row), i32(col), rh.to(dev), w_r.to(dev), W2.to(dev), b2.to(dev), Wc.to(dev),
                                      bc.to(dev), w3.to(dev), ds.to(dev), dagg.to(dev), colptr.to(dev), i32(perm), p_drop, seed,
                                      dab_bf16=False, w_dtype=torch.float32, rowptr=rp, saved=saved)
    out, out2 = bw(), bw()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(out, out2))
    names = ("dAB", "drh", "dw_r", "dW2", "db2", "dWc", "dbc", "dw3")
    refs = (ABf.grad, rhf.grad, wrf.grad, W2f.grad, b2f.grad, Wcf.grad, bcf.grad, w3f.grad)
    for name, got, ref in zip(names, out, refs):
        err = (got.float().cpu() - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        assert err <= 3e-2 * scale, (name, err, scale)
        rel = ((got.float().cpu() - ref).norm() / ref.norm().clamp_min(1e-6)).item()
        assert rel <= 2e-2, (name, rel)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("R,C,ld", [(1000, 512, 512), (333, 1025, 1088), (77, 128, 128), (50, 36, 36), (129, 1088, 1088), (64, 96, 200),
                                    (1003, 256, 256), (333, 200, 256), (4099, 64, 64), (9001, 128, 136)])
def test_layernorm_rows_and_adjoint(R, C, ld, dt):
    """Row LayerNorm of the FABind+ LN-MLPs (model_utils.py:10-74) and its adjoint: the eight-columns-per-lane kernels (16-byte
    accesses; rows and leading dimensions that are multiples of 8) and the scalar fallback (C = 36), incl. the 1025-wide edge input
    inside a 1088-wide buffer, against torch's layer_norm and its autograd."""
    from fabind_amd.plus import engine as pe
    dev = _dev()
    g = torch.Generator().manual_seed(R + C)
    xb = torch.randn(R, ld, generator=g).to(dt)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    cot = torch.randn(R, C, generator=g)
    xr = xb[:, :C].float().clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5)
    (ref * cot).sum().backward()
    xd = xb.to(dev)[:, :C].requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    pad = (C + 7) // 8 * 8
    y = pe.ln_rows(xd, wd, bd, torch.float32, pad)
    assert y.shape[1] == pad and (pad == C or float(y[:, C:].abs().max()) == 0.0)
    tol = 2e-2 if dt == torch.bfloat16 else 1e-4
    assert (y[:, :C].detach().cpu() - ref.detach()).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
    (y[:, :C] * cot.to(dev)).sum().backward()
    assert (xd.grad.float().cpu() - xr.grad).abs().max() <= tol * max(1.0, float(xr.grad.abs().max()))
    assert (wd.grad.cpu() - wr.grad).abs().max() <= 1e-3 * max(1.0, float(wr.grad.abs().max()))
    assert (bd.grad.cpu() - br.grad).abs().max() <= 1e-3 * max(1.0, float(br.grad.abs().max()))


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_linear_relu_with_epilogue_dropout_under_autograd(mode):
    """ops.linear(act_epi=RELU, p_drop > 0) under autograd (FABind+ training: every LN-MLP is Linear -> ReLU -> Dropout,
    model_utils.py:10-74): the mask is applied in the GEMM epilogue and never stored; the backward takes it from the zeros of the saved
    output.  Checked against torch autograd of relu(x W^T + b) * mask / (1 - p) with the epilogue's hash mask restated on the host."""
    from fabind_amd import config, ops
    from fabind_amd import kernels as K
    dev = _dev()
    config.set_precision(mode)
    try:
        g = torch.Generator().manual_seed(5)
        M, N, Kd, p = 700, 256, 192, 0.25
        wd = torch.float32 if mode == "fp32" else torch.bfloat16
        x = torch.randn(M, Kd, generator=g)
        W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(wd).float()
        b = torch.randn(N, generator=g)
        cot = torch.randn(M, N, generator=g)
        xd = x.to(dev).requires_grad_(True)
        Wd = W.to(dev).to(wd).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True)
        # the epilogue's mask is a counter-based hash of (seed, row, column) with the seed drawn from torch's CPU generator
        # (include/fabind_hip.h: FabindGemmArgs.p_drop): reproduced here exactly -- reading the mask off the output would misjudge
        # the units whose activation is tiny, and their derivative is a full-size term
        torch.manual_seed(11)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        torch.manual_seed(11)
        y = ops.linear(xd, Wd, bd, act_epi=K.ACT_RELU, p_drop=p, out_dtype=torch.float32 if mode == "fp32" else torch.bfloat16)
        r_ = torch.arange(M, dtype=torch.int64)[:, None]
        c_ = torch.arange(N, dtype=torch.int64)[None, :]
        MSK = 0xFFFFFFFF
        hsh = (seed + r_ * N + c_) & MSK
        hsh = hsh ^ (hsh >> 16); hsh = (hsh * 0x7feb352d) & MSK
        hsh = hsh ^ (hsh >> 15); hsh = (hsh * 0x846ca68b) & MSK
        hsh = hsh ^ (hsh >> 16)
        keep = ((hsh & 0xFFFF) >= round(p * 65536)).float()
        xr = (x.bfloat16().float() if mode == "bf16" else x).clone().requires_grad_(True)
        Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        pre = xr @ Wr.T + br
        plain = torch.relu(pre)
        frac = 1.0 - float(keep.mean())
        assert abs(frac - p) < 0.02, frac                                                       # the mask has the requested rate
        ref = plain * keep / (1.0 - round(p * 65536) / 65536.0)
        tol = 2e-5 if mode == "fp32" else 2e-2
        assert (y.detach().float().cpu() - ref.detach()).abs().max() <= tol * max(1.0, float(ref.abs().max()))
        (y.float() * cot.to(dev)).sum().backward()
        (ref * cot).sum().backward()
        for got, want, name in ((xd.grad, xr.grad, "dx"), (Wd.grad, Wr.grad, "dW"), (bd.grad, br.grad, "db")):
            e = float((got.float().cpu() - want).abs().max() / want.abs().max())
            assert e <= (1e-4 if mode == "fp32" else 3e-2), (name, e)
    finally:
        config.set_precision("fp32")


def _hash_keep(seed, M, N, p):
    """The GEMM epilogue's dropout mask (counter-based hash of seed + row * N + column; include/fabind_hip.h FabindGemmArgs.p_drop)."""
    r_ = torch.arange(M, dtype=torch.int64)[:, None]
    c_ = torch.arange(N, dtype=torch.int64)[None, :]
    MSK = 0xFFFFFFFF
    hsh = (seed + r_ * N + c_) & MSK
    hsh = hsh ^ (hsh >> 16); hsh = (hsh * 0x7feb352d) & MSK
    hsh = hsh ^ (hsh >> 15); hsh = (hsh * 0x846ca68b) & MSK
    hsh = hsh ^ (hsh >> 16)
    return ((hsh & 0xFFFF) >= round(p * 65536)).float()


@pytest.mark.parametrize("dims", [(900, 136, 264, 128), (1100, 128, 256, 128)])      # register-staged fallback GEMM / LDS-DMA pipelined GEMM
@pytest.mark.parametrize("last_act,with_res,p", [(True, False, 0.25), (False, True, 0.25), (False, False, 0.0), (True, False, 0.0)])
def test_relu_mlp_as_one_autograd_node(last_act, with_res, p, dims):
    """ops.mlp2_relu (the two Linears of a FABind+ LN-MLP, model_utils.py:10-74, as one autograd node: ReLU + dropout in both GEMM
    epilogues, the hidden layer's adjoint inside the input-gradient GEMM of the second Linear) against torch autograd with the
    epilogue masks restated on the host."""
    from fabind_amd import config, ops
    dev = _dev()
    config.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(17)
        M, Kd, N1, N2 = dims
        x = torch.randn(M, Kd, generator=g).bfloat16()
        W1 = (torch.randn(N1, Kd, generator=g) / Kd ** 0.5).bfloat16()
        W2 = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).bfloat16()
        b1, b2 = torch.randn(N1, generator=g), torch.randn(N2, generator=g)
        res = torch.randn(M, N2, generator=g) if with_res else None
        cot = torch.randn(M, N2, generator=g)
        xd = x.to(dev).requires_grad_(True)
        W1d, W2d = W1.to(dev).requires_grad_(True), W2.to(dev).requires_grad_(True)
        b1d, b2d = b1.to(dev).requires_grad_(True), b2.to(dev).requires_grad_(True)
        rd = res.to(dev).requires_grad_(True) if with_res else None
        torch.manual_seed(23)
        seeds = torch.randint(0, 2 ** 31 - 1, (2,)).tolist() if p > 0.0 else (0, 0)
        torch.manual_seed(23)
        y = ops.mlp2_relu(xd, W1d, b1d, W2d, b2d, last_act, rd, torch.float32, p)
        sc = 1.0 / (1.0 - round(p * 65536) / 65536.0)
        k1 = _hash_keep(seeds[0], M, N1, p) if p > 0.0 else torch.ones(M, N1)
        k2 = _hash_keep(seeds[1], M, N2, p) if p > 0.0 else torch.ones(M, N2)
        xr = x.float().clone().requires_grad_(True)
        W1r, W2r = W1.float().clone().requires_grad_(True), W2.float().clone().requires_grad_(True)
        b1r, b2r = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
        rr = res.clone().requires_grad_(True) if with_res else None
        t = torch.relu(xr @ W1r.T + b1r) * k1 * sc
        t = t + (t.bfloat16().float() - t).detach()                      # the hidden activation is stored / contracted as bf16
        ref = t @ W2r.T + b2r
        ref = torch.relu(ref) * k2 * sc if last_act else ref
        ref = ref + rr if with_res else ref
        assert (y.detach().float().cpu() - ref.detach()).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
        (y.float() * cot.to(dev)).sum().backward()
        (ref * cot).sum().backward()
        pairs = [(xd.grad, xr.grad, "dx"), (W1d.grad, W1r.grad, "dW1"), (b1d.grad, b1r.grad, "db1"), (W2d.grad, W2r.grad, "dW2"),
                 (b2d.grad, b2r.grad, "db2")] + ([(rd.grad, rr.grad, "dres")] if with_res else [])
        for got, want, name in pairs:
            e = float((got.float().cpu() - want).abs().max() / want.abs().max())
            assert e <= 3e-2, (name, e)
    finally:
        config.set_precision("fp32")


@pytest.mark.parametrize("Kd", [136, 128])
def test_linear_rowdot_with_epilogue_dropout_under_autograd(Kd):
    """ops.linear_rowdot(act_epi=RELU, p_drop > 0) under autograd (coord_mlp of the FABind+ layers in train mode: Linear -> ReLU ->
    Dropout -> bias-free Linear to one scalar, model_utils.py:45-58): the dropped ReLU output is what the epilogue stores and dots."""
    from fabind_amd import config, ops
    from fabind_amd import kernels as K
    dev = _dev()
    config.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(29)
        M, N, p = 777, 256, 0.25
        x = torch.randn(M, Kd, generator=g).bfloat16()
        W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16()
        b, u = torch.randn(N, generator=g), torch.randn(N, generator=g)
        cot = torch.randn(M, generator=g)
        xd, Wd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
        bd, ud = b.to(dev).requires_grad_(True), u.to(dev).requires_grad_(True)
        torch.manual_seed(31)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        torch.manual_seed(31)
        part = ops.linear_rowdot(xd, Wd, bd, ud, act_epi=K.ACT_RELU, p_drop=p)
        s = part.sum(1)
        keep = _hash_keep(seed, M, N, p)
        sc = 1.0 / (1.0 - round(p * 65536) / 65536.0)
        xr, Wr = x.float().clone().requires_grad_(True), W.float().clone().requires_grad_(True)
        br, ur = b.clone().requires_grad_(True), u.clone().requires_grad_(True)
        ref = ((torch.relu(xr @ Wr.T + br) * keep * sc) * ur).sum(1)
        assert (s.detach().cpu() - ref.detach()).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
        (s * cot.to(dev)).sum().backward()
        (ref * cot).sum().backward()
        for got, want, name in ((xd.grad, xr.grad, "dx"), (Wd.grad, Wr.grad, "dW"), (bd.grad, br.grad, "db"), (ud.grad, ur.grad, "du")):
            e = float((got.float().cpu() - want).abs().max() / want.abs().max())
            assert e <= 3e-2, (name, e)
    finally:
        config.set_precision("fp32")


@pytest.mark.parametrize("M,N,Kd", [(1000, 512, 64), (777, 136, 72), (300, 128, 512)])
def test_gemm_with_bf16_residual(M, N, Kd):
    """FabindGemmArgs.r_dtype = bf16 (the FABind+ pair embedding z <- z + linear_out(a * b), cross_att.py:41-46, kept in bf16):
    the row-contiguous epilogue (bf16 C, no activation) and the generic one (fp32 C) against fp32 torch, and autograd through
    ops.linear hands the residual its gradient in its own dtype."""
    from fabind_amd import config, ops
    from fabind_amd import kernels as K
    dev = _dev()
    config.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(M + N)
        x = torch.randn(M, Kd, generator=g).bfloat16()
        W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16()
        b = torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g).bfloat16()
        ref = x.float() @ W.float().T + b + R.float()
        for od in (torch.bfloat16, torch.float32):
            y, _ = K.gemm(x.to(dev), W.to(dev), bias=b.to(dev), residual=R.to(dev), out_dtype=od)
            assert (y.float().cpu() - ref).abs().max() <= (3e-2 if od == torch.bfloat16 else 2e-2) * max(1.0, float(ref.abs().max()))
        xd, Wd, Rd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True), R.to(dev).requires_grad_(True)
        y = ops.linear(xd, Wd, b.to(dev), residual=Rd, out_dtype=torch.bfloat16)
        cot = torch.randn(M, N, generator=g)
        (y.float() * cot.to(dev)).sum().backward()
        assert Rd.grad.dtype == torch.bfloat16
        assert (Rd.grad.float().cpu() - cot).abs().max() <= 2e-2 * float(cot.abs().max())
        dx_ref = cot.bfloat16().float() @ W.float()
        assert (xd.grad.float().cpu() - dx_ref).abs().max() <= 3e-2 * float(dx_ref.abs().max())
    finally:
        config.set_precision("fp32")


def test_linear_backward_pads_a_narrow_contraction_to_64():
    """ops.linear with 16 output columns over >= 65536 rows (the FABind+ pair-bias Linear, cross_att.py:118-134): the input-gradient
    GEMM's contraction dim is zero-padded to 64 (LDS-DMA kernel instead of the register-staged fallback); same gradients."""
    from fabind_amd import config, ops
    dev = _dev()
    config.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(3)
        M, N, Kd = 70000, 16, 128
        x = torch.randn(M, Kd, generator=g).bfloat16()
        W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16()
        b = torch.randn(N, generator=g)
        cot = torch.randn(M, N, generator=g)
        xd, Wd, bd = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = ops.linear(xd, Wd, bd)
        (y * cot.to(dev)).sum().backward()
        xr, Wr, br = x.float().requires_grad_(True), W.float().requires_grad_(True), b.clone().requires_grad_(True)
        ((xr @ Wr.T + br) * cot).sum().backward()
        for got, want, name in ((xd.grad, xr.grad, "dx"), (Wd.grad, Wr.grad, "dW"), (bd.grad, br.grad, "db")):
            e = float((got.float().cpu() - want).abs().max() / want.abs().max())
            assert e <= 3e-2, (name, e)
    finally:
        config.set_precision("fp32")


@pytest.mark.parametrize("H,dt", [(512, torch.bfloat16), (64, torch.bfloat16), (20, torch.bfloat16), (64, torch.float32)])
def test_edge_concat_rows(H, dt):
    """[h[row] | h[col] | rhohat | 0 ...] per edge (the FABind+ edge-MLP input, egnn.py:104-118): the eight-columns-per-lane bf16
    path and the scalar path against torch indexing."""
    from fabind_amd.plus import engine as pe
    dev = _dev()
    g = torch.Generator().manual_seed(H)
    N, E = 300, 5000
    h = torch.randn(N, H, generator=g)
    row = torch.sort(torch.randint(0, N, (E,), generator=g))[0].to(torch.int32)
    col = torch.randint(0, N, (E,), generator=g).to(torch.int32)
    rh = torch.rand(E, generator=g)
    pad = (2 * H + 1 + 7) // 8 * 8 + 8
    rp = torch.zeros(N + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(torch.bincount(row.long(), minlength=N), 0).to(torch.int32)
    y = pe._EdgeConcat.apply(h.to(dev), rh.to(dev), row.to(dev), col.to(dev), rp.to(dev), None, dt, pad)
    ref = torch.zeros(E, pad)
    ref[:, :H], ref[:, H:2 * H], ref[:, 2 * H] = h[row.long()], h[col.long()], rh
    assert y.dtype == dt and y.shape == (E, pad)
    assert torch.equal(y.float().cpu(), ref.to(dt).float())


@pytest.mark.parametrize("H,H2", [(64, 0), (128, 0), (128, 64), (512, 64), (20, 0)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_pair_hadamard_products_all_lane_groupings(H, H2, dt):
    """fabind_pair_hadamard (InteractionModule's a_i * b_j, model_utils.py:200-223 / FABind+ model_utils.py:96-146): 16, 32 and 64 lanes
    per pair (4, 2, 1 pairs per wave), with and without the second product block, odd pair counts -- against torch indexing."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H + H2)
    N, P = 333, 4099
    T0 = torch.randn(N, 2 * H, generator=g)
    T1 = torch.randn(N, 2 * max(H2, 4), generator=g)
    rp = torch.randint(0, N, (P,), generator=g).to(torch.int32)
    rc = torch.randint(0, N, (P,), generator=g).to(torch.int32)
    t0, t1 = T0.to(dev), T1.to(dev)
    hd = K.pair_hadamard(t0[:, :H], t0[:, H:], t1[:, :H2], t1[:, max(H2, 4):max(H2, 4) + H2], rp.to(dev), rc.to(dev), dt)
    ref = torch.cat([T0[rp.long(), :H] * T0[rc.long(), H:], T1[rp.long(), :H2] * T1[rc.long(), max(H2, 4):max(H2, 4) + H2]], 1)
    assert hd.shape == ref.shape and torch.equal(hd.float().cpu(), ref.to(dt).float())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_rows_hadamard_adjoint_row_walk(mode):
    """ops.rows_hadamard (the distance-map head's LN(p_i) * LN(c_j) over every pocket residue x ligand atom pair, model.py:355) under
    autograd: forward and the atomics-free adjoint (one wave per row of t walks its pairs) against torch indexing + autograd and
    against the float-atomics kernel; a full protein x ligand grid per complex as in the model."""
    from fabind_amd import config, ops
    dev = _dev()
    config.set_precision(mode)
    try:
        g = torch.Generator().manual_seed(3)
        n_p, n_c, W = [150, 97], [40, 23], 512
        npk, ncs = sum(n_p), sum(n_c)
        t = torch.randn(npk + ncs, W, generator=g)
        ia, ib, po, co = [], [], 0, 0
        for P_, C_ in zip(n_p, n_c):
            ia.append(po + torch.arange(P_).repeat_interleave(C_))
            ib.append(npk + co + torch.arange(C_).repeat(P_))
            po, co = po + P_, co + C_
        ia, ib = torch.cat(ia), torch.cat(ib)
        cot = torch.randn(ia.shape[0], W, generator=g)
        tr = t.clone().requires_grad_(True)
        ref = tr[ia] * tr[ib]
        (ref * cot).sum().backward()
        outs = {}
        for walk in (True, False):
            ops.ROWS_HADAMARD_WALK = walk
            td = t.to(dev).requires_grad_(True)
            y = ops.rows_hadamard(td, ia.to(dev), ib.to(dev))
            (y.float() * cot.to(dev)).sum().backward()
            outs[walk] = (y.detach().float().cpu(), td.grad.cpu())
        tol = 1e-5 if mode == "fp32" else 2e-2
        assert (outs[True][0] - ref.detach()).abs().max() <= tol * float(ref.abs().max())
        assert (outs[True][1] - tr.grad).abs().max() <= tol * float(tr.grad.abs().max())
        assert (outs[True][1] - outs[False][1]).abs().max() <= 1e-4 * float(tr.grad.abs().max())
    finally:
        ops.ROWS_HADAMARD_WALK = True
        config.set_precision("fp32")


@pytest.mark.parametrize("H", [512, 128])
@pytest.mark.parametrize("kind,act", [(0, "silu"), (1, "relu"), (0, "relu"), (1, "silu")])
def test_node_chain_forward_matches_two_linears(H, kind, act):
    """csrc/node_chain.hip (node MLP of MC_E_GCL, egnn.py:99-109, and the Transition blocks, model_utils.py:162-175, as one forward
    kernel with the hidden activation in LDS) against fp32 torch on the same bf16 operands, with the hidden activation rounded to bf16
    as the kernel holds it; ragged last tile; with / without residual and bf16 copy."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(H + kind)
    M = 1000 + 37
    hid = H if kind == 0 else 2 * H
    kin = 2 * H if kind == 0 else H
    X = torch.randn(M, kin, generator=g).bfloat16()
    W1 = (torch.randn(hid, kin, generator=g) / kin ** 0.5).bfloat16()
    W2 = (torch.randn(H, hid, generator=g) / hid ** 0.5).bfloat16()
    b1, b2 = torch.randn(hid, generator=g) * 0.3, torch.randn(H, generator=g) * 0.3
    R = torch.randn(M, H, generator=g)
    fa = torch.nn.functional.silu if act == "silu" else torch.relu
    t = fa(X.float() @ W1.float().T + b1).bfloat16().float()
    ref = t @ W2.float().T + b2 + R
    code = K.ACT_SILU if act == "silu" else K.ACT_RELU
    packs = K.node_chain_pack(W1.to(dev), W2.to(dev), kind)
    Xd = X.to(dev)
    X1, X2 = (Xd[:, :H].contiguous(), Xd[:, H:].contiguous()) if kind == 0 else (Xd, None)
    out, out16 = K.node_chain_fwd(X1, X2, packs, b1.to(dev), b2.to(dev), code, kind, residual=R.to(dev), want16=True)
    err = float((out.cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 5e-3, err
    assert torch.equal(out16.float().cpu(), out.cpu().bfloat16().float())
    out_nr, none16 = K.node_chain_fwd(X1, X2, packs, b1.to(dev), b2.to(dev), code, kind)
    assert none16 is None and float((out_nr.cpu() - (ref - R)).abs().max() / ref.abs().max()) <= 5e-3


@pytest.mark.parametrize("H", [512, 128])
@pytest.mark.parametrize("kind,act", [(0, "silu"), (1, "relu"), (0, "relu"), (1, "silu")])
def test_node_chain_x3_forward_matches_float64(H, kind, act):
    """csrc/node_chain.hip, split-precision form (round 6: the node MLP / Transition of a no-grad pass in the bf16 mode's split-precision
    sites level 3): fp32 rows, fp32 master weights as hi | lo packs, hidden layer never rounded to bf16 -- against float64 of the same
    fp32 operands: fp32-grade (the two-launch bf16 chain above is 5e-3), and equal to the two gemm_x3 launches it replaces to 1e-5."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(7 * H + kind)
    M = 1000 + 37
    hid = H if kind == 0 else 2 * H
    kin = 2 * H if kind == 0 else H
    X = torch.randn(M, kin, generator=g)
    W1 = torch.randn(hid, kin, generator=g) / kin ** 0.5
    W2 = torch.randn(H, hid, generator=g) / hid ** 0.5
    b1, b2 = torch.randn(hid, generator=g) * 0.3, torch.randn(H, generator=g) * 0.3
    R = torch.randn(M, H, generator=g)
    fa = torch.nn.functional.silu if act == "silu" else torch.relu
    ref = (fa(X.double() @ W1.double().T + b1.double()) @ W2.double().T + b2.double() + R.double())
    code = K.ACT_SILU if act == "silu" else K.ACT_RELU
    packs = K.node_chain_x3_pack(W1.to(dev), W2.to(dev), kind)
    assert len(packs) == 8 and (packs[6] is None) == (kind == 0)
    Xd = X.to(dev)
    X1, X2 = (Xd[:, :H].contiguous(), Xd[:, H:].contiguous()) if kind == 0 else (Xd, None)
    out, out16 = K.node_chain_x3_fwd(X1, X2, packs, b1.to(dev), b2.to(dev), code, kind, residual=R.to(dev), want16=True)
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, err
    assert torch.equal(out16.float().cpu(), out.cpu().bfloat16().float())
    # the two split-precision GEMM launches of the same chain
    t, _ = K.gemm(X1, W1.to(dev), bias=b1.to(dev), A2=X2, act_epi=code, force_x3=True)
    two, _ = K.gemm(t, W2.to(dev), bias=b2.to(dev), residual=R.to(dev), force_x3=True)
    assert float((out - two).abs().max() / two.abs().max()) <= 1e-5
    out_nr, none16 = K.node_chain_x3_fwd(X1, X2, packs, b1.to(dev), b2.to(dev), code, kind)
    assert none16 is None and float((out_nr.double().cpu() - (ref - R.double())).abs().max() / ref.abs().max()) <= 2e-5
    assert torch.equal(K.node_chain_x3_fwd(X1, X2, packs, b1.to(dev), b2.to(dev), code, kind)[0], out_nr)          # bit-repeatable


@pytest.mark.parametrize("H", [512, 128])
@pytest.mark.parametrize("kind,act", [(0, "silu"), (1, "relu")])
def test_node_chain_x3_training_forward_saves_the_backward_operands(H, kind, act):
    """The SAVE form of the split-precision chain (ops._MLP2 under autograd at split-precision sites level 3): besides the output it leaves
    bf16(hidden) and, for SiLU, bf16(silu'(pre)) -- the operands of the unchanged bf16 backward -- and the packs of MANY chains come from
    one launch (kernels.node_chain_x3_pack_many == node_chain_x3_pack per chain)."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(11 * H + kind)
    M = 700 + 13
    hid = H if kind == 0 else 2 * H
    kin = 2 * H if kind == 0 else H
    X = torch.randn(M, kin, generator=g)
    W1 = (torch.randn(hid, kin, generator=g) / kin ** 0.5).to(dev)
    W2 = (torch.randn(H, hid, generator=g) / hid ** 0.5).to(dev)
    b1, b2 = torch.randn(hid, generator=g) * 0.3, torch.randn(H, generator=g) * 0.3
    R = torch.randn(M, H, generator=g)
    pre = X.double() @ W1.double().cpu().T + b1.double()
    hidden = torch.nn.functional.silu(pre) if act == "silu" else torch.relu(pre)
    ref = hidden @ W2.double().cpu().T + b2.double() + R.double()
    code = K.ACT_SILU if act == "silu" else K.ACT_RELU
    many = K.node_chain_x3_pack_many([(W1, W2, kind), (W1 * 2.0, W2, kind)])
    one = K.node_chain_x3_pack(W1, W2, kind)
    for a_, b_ in zip(many[0], one):
        assert (a_ is None) == (b_ is None) and (a_ is None or torch.equal(a_.reshape(-1), b_.reshape(-1)))
    Xd = X.to(dev)
    X1, X2 = (Xd[:, :H].contiguous(), Xd[:, H:].contiguous()) if kind == 0 else (Xd, None)
    out, out16, t16, d16 = K.node_chain_x3_fwd(X1, X2, many[0], b1.to(dev), b2.to(dev), code, kind, residual=R.to(dev), want16=True, save=True)
    assert float((out.double().cpu() - ref).abs().max() / ref.abs().max()) <= 2e-5
    assert t16.shape == (M, hid) and float((t16.double().cpu() - hidden).abs().max()) <= 2.0 ** -8 * float(hidden.abs().max())
    if act == "silu":
        sg = torch.sigmoid(pre)
        dref = sg * (1 + pre * (1 - sg))
        assert d16 is not None and float((d16.double().cpu() - dref).abs().max()) <= 6e-3          # bf16 rounding + v_exp / v_rcp
    else:
        assert d16 is None
    plain, _ = K.node_chain_x3_fwd(X1, X2, many[0], b1.to(dev), b2.to(dev), code, kind, residual=R.to(dev))
    assert torch.equal(plain, out)                                                                 # saving does not change the result


def test_gemm_tn_multi_matches_single_launches():
    """fabind_gemm_tn_multi: many weight-gradient contractions as one launch + one reduction (kernels.gemm_tn_queued / tn_flush).  Every
    job against the single-launch result of the same operands (same kernel code path; split counts differ, so fp32 sums are compared
    to 1e-5 of the largest entry) and against float64: ragged shapes, column sums, bf16 outputs, two jobs filling the column blocks of
    one gradient, a column-slice operand."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(9088, 512, 512), (9088, 512, 1024), (2624, 256, 512), (777, 8, 512), (40000, 1024, 576), (85000, 512, 512), (130, 136, 72),
              (9088, 1536, 512), (31, 64, 64)]
    jobs, refs = [], []
    for k, (E, M, N) in enumerate(shapes):
        Y = (torch.randn(E, M, generator=g) * 0.3 + 0.05).bfloat16().to(dev)
        X = torch.randn(E, N, generator=g).bfloat16().to(dev)
        out_dt = torch.bfloat16 if k % 3 == 1 else torch.float32
        with_cs = k % 2 == 0
        out = torch.full((M, N), float("nan"), dtype=out_dt, device=dev)
        tail = torch.full((M,), float("nan"), dtype=torch.float32, device=dev) if with_cs else None
        K.gemm_tn_queued(Y, X, out, tail)
        jobs.append((Y, X, out, tail))
    # two jobs -> the column blocks of one [M, K1 + K2] gradient (the [x | x2] Linear)
    E, M, K1, K2 = 5000, 512, 512, 64
    Yc = torch.randn(E, M, generator=g).bfloat16().to(dev)
    Xa, Xb = torch.randn(E, K1, generator=g).bfloat16().to(dev), torch.randn(E, K2 + 8, generator=g).bfloat16().to(dev)
    both = torch.full((M, K1 + K2), float("nan"), dtype=torch.float32, device=dev)
    K.gemm_tn_queued(Yc, Xa, both[:, :K1], None)
    K.gemm_tn_queued(Yc, Xb[:, 8:], both[:, K1:], None)
    assert K.tn_pending() == len(shapes) + 2
    K.tn_flush()
    assert K.tn_pending() == 0
    for Y, X, out, tail in jobs:
        single = K.gemm_tn(Y, X)
        scale = float(single.abs().max()) + 1e-6
        tol = 1e-5 if out.dtype == torch.float32 else 8e-3
        assert float((out.float() - single).abs().max()) <= tol * scale
        ref = Y.double().T @ X.double()
        assert float((out.double() - ref).abs().max()) <= (2e-5 if out.dtype == torch.float32 else 8e-3) * float(ref.abs().max()) + 1e-6
        if tail is not None:
            cs = Y.double().sum(0)
            assert float((tail.double() - cs).abs().max()) <= 1e-5 * float(Y.double().abs().sum(0).max()) + 1e-6
    ref = torch.cat([Yc.double().T @ Xa.double(), Yc.double().T @ Xb[:, 8:].double()], 1)
    assert float((both.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # a second flush with nothing queued is a no-op; a repeated queue gives bitwise the same result (fixed split order)
    K.tn_flush()
    Y, X, out, tail = jobs[0]
    again = torch.empty_like(out)
    K.gemm_tn_queued(Y, X, again, torch.empty_like(tail))
    K.tn_flush()
    assert torch.equal(again, out)


@pytest.mark.parametrize("case", ["composed_weight", "weight_used_twice", "leaf_weight_accumulates", "x2_and_bias"])
def test_queued_weight_gradients_are_flushed_before_anything_reads_them(case):
    """The contract of the weight-gradient queue under autograd (ops._tn_queue): a queued gradient is written before its first reader --
    a torch op that composed the weight from parameters, autograd's own sum when one weight feeds two Linears, the accumulation into
    an existing .grad, the end of the backward pass.  Each case against the same graph with the queue switched off."""
    from fabind_amd import config, kernels as K, ops
    dev = _dev()
    old = config.get_precision()
    config.set_precision("bf16")
    g = torch.Generator().manual_seed(3)
    n, H = 3000, 128
    x0 = torch.randn(n, H, generator=g).to(dev)
    A0 = (torch.randn(H, H, generator=g) * 0.1).to(dev)
    B0 = (torch.randn(H, H, generator=g) * 0.1).to(dev)
    b0 = (torch.randn(H, generator=g) * 0.1).to(dev)

    def run(defer):
        K.TN_DEFER = defer
        K.TN_STATS.update(queued=0, flushes=0, direct=0)
        A, B, b = (t.clone().requires_grad_(True) for t in (A0, B0, b0))
        x = x0.clone().requires_grad_(True)
        if case == "composed_weight":
            W = (A @ B).to(torch.bfloat16)                       # MmBackward / ToCopyBackward read dW as soon as it is "ready"
            y = ops.linear(x, W, b, act_epi=K.ACT_RELU)
            y = ops.linear(y, B.to(torch.bfloat16), None)
        elif case == "weight_used_twice":
            W = A.to(torch.bfloat16)
            y = ops.linear(ops.linear(x, W, b, act_epi=K.ACT_RELU), W, None) + 0.0 * B.sum()
        elif case == "leaf_weight_accumulates":
            Wl = A0.to(torch.bfloat16).clone().requires_grad_(True)   # a leaf consumed directly; .grad already holds a value
            Wl.grad = torch.ones_like(Wl)
            y = ops.linear(x, Wl, b) + 0.0 * (A.sum() + B.sum())
            (y * y).mean().backward()
            K.TN_DEFER = True
            return [Wl.grad.float().clone(), b.grad.clone(), x.grad.clone()], dict(K.TN_STATS)
        else:
            W = torch.cat([A, B], 1).to(torch.bfloat16)          # [H, 2H]: the [x | x2] Linear, two queued jobs + the bias
            y = ops.linear(x, W, b, x2=torch.tanh(x), act_epi=K.ACT_SILU)
        (y * y).mean().backward()
        K.TN_DEFER = True
        return [A.grad.clone(), B.grad.clone(), b.grad.clone(), x.grad.clone()], dict(K.TN_STATS)

    try:
        got, st = run(True)
        want, st0 = run(False)
    finally:
        K.TN_DEFER = True
        config.set_precision(old)
    assert st["queued"] >= 1 and st["flushes"] >= 1 and st0["queued"] == 0
    assert K.tn_pending() == 0
    for a, b_ in zip(got, want):
        assert torch.isfinite(a).all()
        # (the queued contraction cuts its rows differently from the single launch, and weight gradients are delivered as bf16 here:
        #  agreement to a bf16 rounding; an unflushed buffer would be garbage, not 2^-9 off)
        assert float((a - b_).abs().max()) <= 1e-2 * float(b_.abs().max()) + 1e-7


def test_pack_frag_multi_equals_the_torch_permute():
    """kernels.pack_frag_multi: the fragment packs of many weights (fp32 / bf16, plain and transposed views, several sizes) from one
    launch are bit-equal to kernels.pack_frag's torch permute-copy of each."""
    from fabind_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    ws = []
    for (N, Kd, dt) in [(512, 512, torch.bfloat16), (512, 512, torch.float32), (128, 64, torch.bfloat16), (64, 128, torch.float32), (256, 512, torch.bfloat16)]:
        W = torch.randn(N, Kd, generator=g).to(dt).to(dev)
        ws += [W, W.t()] if N == Kd or True else [W]
    ws.append(torch.randn(96, 512 + 64, generator=g).to(dev)[:, 64:])            # a column slice (row stride != width)
    packs = K.pack_frag_multi(ws)
    for W, p in zip(ws, packs):
        ref = K.pack_frag(W.contiguous())
        assert p.shape == ref.shape and p.dtype == torch.bfloat16
        assert torch.equal(p, ref)


def test_layout_ranges_is_a_lower_bound_per_complex():
    """engine.Layout.ranges on int32 lists (one fabind_lower_bound launch) equals the torch bucketize / histogram / cumsum form it
    replaced -- ragged complexes, complexes without any entry, an empty list."""
    from fabind_amd import engine
    dev = _dev()
    sizes = [(30, 7), (5, 3), (64, 1), (12, 9)]
    bid = torch.cat([torch.full((p + c + 2,), b) for b, (p, c) in enumerate(sizes)]).to(dev)
    seg = torch.cat([torch.cat([torch.zeros(c + 1), torch.ones(p + 1)]) for p, c in sizes]).to(dev)
    lay = engine.Layout(bid, seg)
    g = torch.Generator().manual_seed(1)
    for keep in ([0, 1, 2, 3], [0, 3], [2], []):
        rows = []
        for b in keep:
            lo = int(lay.off[b])
            rows.append(lo + torch.sort(torch.randint(0, int(lay.n[b]), (11 + 3 * b,), generator=g))[0])
        idx = (torch.cat(rows) if rows else torch.zeros(0)).to(torch.int32).to(dev)
        got = lay.ranges(idx)
        want = lay.ranges(idx.long())                     # int64: the torch path
        assert got.dtype == torch.int32 and torch.equal(got, want), (keep, got, want)


@pytest.mark.parametrize("H", [64, 256, 512])
def test_inter_attn_rows_dealt_by_degree_match_one_wave_per_row(H, monkeypatch):
    """csrc/inter_attn_rows.hip (round 5): the inter-edge attention (egnn.py:186-252) with heavy rows split over four waves + log-sum-exp
    combine, light rows one wave each, empty rows pass-through -- against the one-wave-per-row kernels (attn.hip / bwd.hip, themselves
    pinned to the reference by the stack tests) on the same operands: forward outputs, every gradient of ops.inter_attn, bit-identical
    repeats.  A compact complex (ligand atoms with > 64 edges: several 64-edge batches per quarter) next to an ordinary one and a third
    without any inter edge."""
    from fabind_amd import engine, kernels as K, ops, synthetic
    dev = _dev()
    engine.set_precision("fp32")
    inp = synthetic.make_stack_batch([(170, 9), (230, 24), (60, 7)], 32, seed=21, snap=False)
    pr = (inp["segment_id"] > 0.5) & ~inp["is_global"]
    lig = (inp["segment_id"] < 0.5) & ~inp["is_global"]
    g0 = torch.Generator().manual_seed(5)
    sel_l, sel_p = lig & (inp["batch_id"] == 0), pr & (inp["batch_id"] == 0)
    centre = inp["X"][sel_l].mean(0, keepdim=True)
    inp["X"][sel_l] = centre + 0.3 * (torch.rand(int(sel_l.sum()), 1, 3, generator=g0) - 0.5)
    inp["X"][sel_p] = centre + 1.3 * (torch.rand(int(sel_p.sum()), 1, 3, generator=g0) - 0.5) * 2 * 0.75
    inp["X"][lig & (inp["batch_id"] == 2)] += 50.0                       # complex 2: the ligand is far away, no inter edge
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    lay = engine.Layout(t["batch_id"], t["segment_id"])
    br, bc_ = t["compound_edge_index"][0].to(torch.int32), t["compound_edge_index"][1].to(torch.int32)
    g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
    order, n_heavy, n_act = g.int_deal
    deg = (g.rp_int[1:] - g.rp_int[:-1]).cpu()
    assert n_heavy > 0 and n_act > n_heavy and n_act < lay.N and int(deg.max()) > 128
    assert int((deg > 0).sum()) == n_act and int((deg > K.INTER_ATTN_HEAVY).sum()) == n_heavy
    od = deg[order.cpu().long()]
    assert bool((od[:-1] >= od[1:]).all()) and sorted(order.cpu().tolist()) == list(range(lay.N))
    N, E = lay.N, g.col_int.shape[0]
    gen = torch.Generator().manual_seed(H)
    rnd = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
    qkv0, h0, x0 = rnd(N, 3 * H) * 0.3, rnd(N, H), t["X"][:, 0].contiguous().clone()
    Wc0, bc0 = rnd(H, H) / H ** 0.5, rnd(H) * 0.1
    w_rk0, w_rv0, wcr0, w30 = rnd(H) * 0.3, rnd(H) * 0.3, rnd(H) * 0.3, rnd(H) * 0.05
    bias0 = rnd(E // 2, 4) * 0.5
    co_h, co_x = rnd(N, H), rnd(N, 3)

    def run(rows):
        monkeypatch.setattr(K, "INTER_ATTN_ROWS", rows)
        leaf = lambda v: v.clone().requires_grad_(True)
        qkv, h, x, Wc, bc, w_rk, w_rv, w3, bias = map(leaf, (qkv0, h0, x0, Wc0, bc0, w_rk0, w_rv0, w30, bias0))
        wcr = Wc @ w_rv
        d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay, g.int_by_col)
        ho, xo, al = ops.inter_attn(qkv, None, H, h, x, d, rhohat, g, bias, w_rk, w_rv, wcr, w3, 2.0, Wc=Wc, bc=bc)
        ((ho * co_h).sum() + (xo * co_x).sum()).backward()
        return [ho.detach(), xo.detach(), al.detach()] + [v.grad for v in (qkv, h, x, Wc, bc, w_rk, w_rv, w3, bias)]

    new, new2, old = run(True), run(True), run(False)
    names = ("h_out", "x_out", "alpha", "dqkv", "dh", "dx", "dWc", "dbc", "dw_rk", "dw_rv", "dw3", "dbias")
    for nm, a_, b_, c_ in zip(names, new, new2, old):
        assert torch.equal(a_, b_), nm                                   # deterministic
        err = float((a_ - c_).abs().max()) / max(1e-6, float(c_.abs().max()))
        assert err <= 2e-5, (nm, err)
    assert float(new[2].sum()) > 0 and abs(float(new[2].sum()) - n_act) <= 1e-3 * n_act      # softmax rows sum to one


def test_row_gathers_with_copy_and_segment_sum_adjoints_match_index_select():
    """ops.take_unique_rows (repeat-free index: the adjoint is a row copy) and ops.take_rows_few (narrow fp32 rows gathered many times:
    the adjoint is a stable sort + one fixed-order segment sum) against autograd's index_select -- values and gradients; the
    many-to-few gradient is bit-identical run to run (autograd's own is an index_put with contended float atomics)."""
    from fabind_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, 64, generator=g).to(dev)
    idx = torch.randperm(3000, generator=g)[:1700].to(dev)
    cot = torch.randn(1700, 64, generator=g).to(dev)
    for dt in (torch.float32, torch.bfloat16):
        a, b = x.to(dt).clone().requires_grad_(True), x.to(dt).clone().requires_grad_(True)
        ya, yb = ops.take_unique_rows(a, idx), b.index_select(0, idx)
        assert torch.equal(ya, yb)
        (ya.float() * cot).sum().backward()
        (yb.float() * cot).sum().backward()
        assert torch.equal(a.grad, b.grad)
    c = torch.randn(640, 3, generator=g).to(dev)
    many = torch.randint(0, 640, (150000,), generator=g).to(dev)
    many[:5000] = 7                                                  # one row with thousands of readers
    cot3 = torch.randn(150000, 3, generator=g).to(dev)
    grads = []
    for _ in range(2):
        a = c.clone().requires_grad_(True)
        ya = ops.take_rows_few(a, many)
        assert torch.equal(ya, c.index_select(0, many))
        (ya * cot3).sum().backward()
        grads.append(a.grad.clone())
    assert torch.equal(grads[0], grads[1])
    ref = torch.zeros(640, 3, dtype=torch.float64, device=dev).index_add_(0, many, cot3.double())
    assert float((grads[0].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float(grads[0][torch.bincount(many, minlength=640) == 0].abs().max() if (torch.bincount(many, minlength=640) == 0).any() else 0.0) == 0.0


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
@pytest.mark.parametrize("with_res", [True, False])
def test_linear_with_epilogue_dropout_ahead_of_the_residual_under_autograd(prec, with_res):
    """Round 5, train mode: y = r + drop(x W^T + b) (nn.Dropout ahead of `h + ...`: FABind/fabind/models/egnn.py:106, cross_att.py:128)
    as ONE GEMM with the counter-based mask in its fp32 epilogue; the adjoint regenerates the mask from (seed, row, col)
    (fabind_mul_dropmask_colsum).  Checked against the explicit formula with the mask read off the forward output: values, drop rate,
    and the gradients of x, W, b and the residual; the bf16 operand copy of y carries the same values."""
    from fabind_amd import config, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    M, Kd, N, pd = 3001, 256, 512, 0.1
    x0, W0 = torch.randn(M, Kd, generator=g).to(dev), (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
    b0, r0, cot = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    config.set_precision(prec)
    try:
        wd = ops.mm_dtype()
        x, W, b, r = (t.clone().requires_grad_(True) for t in (x0, W0.to(wd), b0, r0))
        torch.manual_seed(11)
        y = ops.linear(x, W, b, residual=r if with_res else None, p_drop=pd, want16=True)
        with torch.no_grad():
            y0 = ops.linear(x0, W0.to(wd), b0)                     # no dropout, no residual
        u = y.detach() - (r0 if with_res else 0.0)
        keep = (u.abs() > 1e-6 * (1.0 + r0.abs() if with_res else 1.0)) | (y0.abs() < 1e-3)
        rate = 1.0 - float(keep.float().mean())
        assert abs(rate - pd) < 5e-3, rate
        ref_u = y0 * keep / (1.0 - pd)
        assert float((u - ref_u).abs().max()) <= 2e-3 * float(y0.abs().max())
        c16 = getattr(y, "_fab_b16", None)
        assert c16 is None or float((c16[1].float() - y.detach()).abs().max()) <= 1e-2 * float(y.detach().abs().max())
        (y * cot).sum().backward()
        # reference: the same mask as a constant, torch autograd in fp32 on the operands the kernels saw
        xr, Wr, br, rr = (t.clone().float().requires_grad_(True) for t in (x0, W0.to(wd).float(), b0, r0))
        xin = xr if prec != "bf16" else xr + (xr.detach().to(torch.bfloat16).float() - xr.detach())
        yr = (xin @ Wr.t() + br) * keep / (1.0 - pd) + (rr if with_res else 0.0)
        (yr * cot).sum().backward()
    finally:
        config.set_precision("fp32")
    tol = 2e-2 if prec == "bf16" else 1e-2                          # (bf16: d pre is rounded to bf16 before the two contractions; bf16x3: its weight gradient contracts bf16 roundings)
    for name, a, b_ in (("x", x.grad, xr.grad), ("W", W.grad.float(), Wr.grad), ("b", b.grad, br.grad)) + ((("r", r.grad, rr.grad),) if with_res else ()):
        err = float((a - b_).norm() / b_.norm())
        assert err <= tol, (name, err)
    if with_res:
        assert torch.equal(r.grad, cot)


def test_mlp2_with_dropout_ahead_of_the_residual_matches_the_two_linear_form():
    """ops.mlp2(..., p_drop): the node MLP of MC_E_GCL in train mode as one autograd node; same seeds -> the same mask as
    linear -> linear(residual, p_drop): values bit-equal, gradients to bf16 accuracy."""
    from fabind_amd import config, ops
    from fabind_amd import kernels as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    M, H, pd = 2050, 512, 0.1
    h0, agg0 = torch.randn(M, H, generator=g).to(dev), torch.randn(M, H, generator=g).to(dev)
    W1, b1 = (torch.randn(H, 2 * H, generator=g) / (2 * H) ** 0.5).to(dev), torch.randn(H, generator=g).to(dev)
    W2, b2 = (torch.randn(H, H, generator=g) / H ** 0.5).to(dev), torch.randn(H, generator=g).to(dev)
    cot = torch.randn(M, H, generator=g).to(dev)
    config.set_precision("bf16")
    try:
        res = []
        for fused in (True, False):
            h, agg = h0.clone().requires_grad_(True), agg0.clone().requires_grad_(True)
            ws = [t.clone().to(torch.bfloat16).requires_grad_(True) if t.dim() == 2 else t.clone().requires_grad_(True) for t in (W1, b1, W2, b2)]
            torch.manual_seed(5)
            if fused:
                y = ops.mlp2(h, ws[0], ws[1], K.ACT_SILU, ws[2], ws[3], residual=h, x2=agg, p_drop=pd)
            else:
                t = ops.linear(h, ws[0], ws[1], x2=agg, act_epi=K.ACT_SILU, out_dtype=torch.bfloat16)
                y = ops.linear(t, ws[2], ws[3], residual=h, p_drop=pd)
            (y * cot).sum().backward()
            res.append((y.detach(), h.grad, agg.grad, [w.grad.float() for w in ws]))
    finally:
        config.set_precision("fp32")
    assert torch.equal(res[0][0], res[1][0])
    for a, b_ in [(res[0][1], res[1][1]), (res[0][2], res[1][2])] + list(zip(res[0][3], res[1][3])):
        assert float((a - b_).norm() / b_.norm()) <= 1e-2


def test_drop_mix_is_h_plus_dropout_of_the_difference():
    """ops.drop_mix: h + nn.Dropout(p)(hn - h) as one pass each way with a regenerated counter-based mask (train mode, egnn.py:236)."""
    from fabind_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    h0, hn0, cot = (torch.randn(1001, 512, generator=g).to(dev) for _ in range(3))
    pd = 0.1
    h, hn = h0.clone().requires_grad_(True), hn0.clone().requires_grad_(True)
    torch.manual_seed(2)
    out = ops.drop_mix(h, hn, pd)
    keep = out.detach() != h0
    assert abs(1.0 - float(keep.float().mean()) - pd) < 5e-3
    pe = int(pd * 65536.0 + 0.5) / 65536.0                       # the kernels' 16-bit threshold
    ref = h0 + (hn0 - h0) * keep / (1.0 - pe)
    assert float((out.detach() - ref).abs().max()) <= 1e-5
    (out * cot).sum().backward()
    dhn = cot * keep / (1.0 - pe)
    assert float((hn.grad - dhn).abs().max()) <= 1e-5 and float((h.grad - (cot - dhn)).abs().max()) <= 1e-5
    torch.manual_seed(2)
    with torch.no_grad():
        assert torch.equal(ops.drop_mix(h0, hn0, pd), out.detach())      # the no-grad form draws the same mask from the same seed
    assert ops.drop_mix(h0, hn0, 0.0) is hn0
