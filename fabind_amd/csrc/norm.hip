// LayerNorm over the last dimension (one wave per row), forward and backward.
// Replaces torch.nn.LayerNorm in Transition_diff_out_dim and the dist-map head
// (reference models/model.py:15,22,29,352-353).
#include <algorithm>
#include "common.h"
#include "fabind_hip.h"

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, int R, int C,
                                                            float* y, float* mean, float* rstd) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* xp = x + (size_t)r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xp[c];
    const float mu = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { float t = xp[c] - mu; v += t * t; }
    const float rs = rsqrtf(wave_sum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[(size_t)r * C + c] = (xp[c] - mu) * rs * w[c] + b[c];
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

extern "C" int fabind_layernorm_fwd(const float* x, const float* w, const float* b, float eps, int R, int C, float* y,
                                    float* mean, float* rstd, hipStream_t stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, x, w, b, eps, R, C, y, mean, rstd);
    FB_CHECK_LAUNCH();
    return 0;
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*w ; per-row partials of dw, db go to part[2][R/4 blocks][C]
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, const float* mean,
                                                            const float* rstd, int R, int C, float* dx, float* dwp,
                                                            float* dbp) {
    extern __shared__ float sh[];  // [4 waves][2][C]: every wave's row contribution, combined in wave order (fixed summation order:
                                   // the LDS float atomics of the first form made the heads' LayerNorm gradients differ by 1 ulp between runs)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    float* mine = sh + (size_t)wv * 2 * C;
    if (r < R) {
        const float mu = mean[r], rs = rstd[r];
        const float* xp = x + (size_t)r * C;
        const float* gp = dy + (size_t)r * C;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float g = gp[c] * w[c], xh = (xp[c] - mu) * rs;
            s1 += g; s2 += g * xh;
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
        for (int c = lane; c < C; c += 64) {
            float xh = (xp[c] - mu) * rs, g = gp[c] * w[c];
            dx[(size_t)r * C + c] = rs * (g - s1 - xh * s2);
            mine[c] = gp[c] * xh;
            mine[C + c] = gp[c];
        }
    } else {
        for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = ((sh[c] + sh[2 * C + c]) + sh[4 * C + c]) + sh[6 * C + c];
        dbp[(size_t)blockIdx.x * C + c] = ((sh[C + c] + sh[3 * C + c]) + sh[5 * C + c]) + sh[7 * C + c];
    }
}

extern "C" int fabind_layernorm_bwd(const float* x, const float* w, const float* dy, const float* mean,
                                    const float* rstd, int R, int C, float* dx, float* dw_part, float* db_part,
                                    hipStream_t stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 8 * C * sizeof(float), stream, x, w, dy, mean,
                       rstd, R, C, dx, dw_part, db_part);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// FABind+ LN-MLPs (FABind_plus/fabind/models/model_utils.py:10-74): LayerNorm in front of every MLP.
// Row LayerNorm with free input / output types and strides; columns [C, pad_to) of the output are zero-filled so that the
// result can feed a GEMM whose contraction dimension is padded to a multiple of 8.  One wave per row.
// ------------------------------------------------------------------------------------------------
// NPL = ceil(C / 64) values per lane are kept in registers: the row is read once (2-byte / 4-byte scalar loads, 64
// consecutive columns per wave instruction), statistics by wave reductions, one write.
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             float eps, int R, int C, void* y, int y_dt, int ldy, int pad_to) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t xo = (size_t)r * ldx, yo = (size_t)r * ldy;
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int c = k * 64 + lane;
        v[k] = c < C ? ld_any(x, x_dt, xo + c) : 0.f;
        s += v[k];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int c = k * 64 + lane;
        const float t = c < C ? v[k] - mu : 0.f;
        q += t * t;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int c = k * 64 + lane;
        if (c < C) st_any(y, y_dt, yo + c, (v[k] - mu) * rs * w[c] + b[c]);
        else if (c < pad_to) st_any(y, y_dt, yo + c, 0.f);
    }
    for (int c = NPL * 64 + lane; c < pad_to; c += 64) st_any(y, y_dt, yo + c, 0.f);
}
// The same with EIGHT consecutive columns per lane (one 16-byte access for bf16, two for fp32, on the input, the output and the
// weight / bias vectors): round 3 -- the scalar form above issued 2-byte loads, 8 per lane at C = 512, and ran at a quarter of the
// rate of a copy (FABind+ training: 18 ms per step in these kernels).  Columns at or beyond C inside the last chunk are masked.
__device__ __forceinline__ void ln_wb8(const float* __restrict__ p, int c0, int C, float (&o)[8]) {
    if (c0 + 8 <= C) {
        const float4 a = *(const float4*)(p + c0), b = *(const float4*)(p + c0 + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (c0 + q < C) ? p[c0 + q] : 0.f;
    }
}
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_rows8_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float eps, int R, int C, void* y, int y_dt, int ldy, int pad_to) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t xo = (size_t)r * ldx, yo = (size_t)r * ldy;
    F8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c0 = (k * 64 + lane) * 8;
        if (c0 < C) v[k] = ld8_any(x, x_dt, xo + c0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (c0 + q >= C) v[k].v[q] = 0.f;
            s += v[k].v[q];
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c0 = (k * 64 + lane) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float t = (c0 + q < C) ? v[k].v[q] - mu : 0.f; qq += t * t; }
    }
    const float rs = rsqrtf(wave_sum(qq) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c0 = (k * 64 + lane) * 8;
        if (c0 >= pad_to) continue;
        F8 o;
        if (c0 < C) {
            float wv[8], bv[8];
            ln_wb8(w, c0, C, wv);
            ln_wb8(b, c0, C, bv);
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = (c0 + q < C) ? (v[k].v[q] - mu) * rs * wv[q] + bv[q] : 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = 0.f;
        }
        st8_any(y, y_dt, yo + c0, o);
    }
    for (int c0 = (NCH * 64 + lane) * 8; c0 < pad_to; c0 += 512) {
        F8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = 0.f;
        st8_any(y, y_dt, yo + c0, o);
    }
}
// Narrow rows (C <= 8 SW, SW = 16 or 32 lanes): 64 / SW rows per wave, reductions over the SW lanes of a row.  With one row per wave a
// 128-wide row -- the pair embedding of the FABind+ pocket model, 3.9 M rows per batch -- kept 16 of 64 lanes busy: 1.2 ms forward and
// 2.2 ms adjoint per launch for 1 GB / 3 GB of traffic.
template <int SW> __device__ __forceinline__ float ln_sub_sum(float v) {
#pragma unroll
    for (int m = 1; m < SW; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <int SW>
__global__ __launch_bounds__(256) void layernorm_rows8s_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float eps, int R, int C, void* y, int y_dt, int ldy, int pad_to) {
    constexpr int RW = 64 / SW;
    const int lane = threadIdx.x & 63, sl = lane % SW, sr = lane / SW;
    const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + sr;
    if (r >= R) return;                                    // (a row's SW lanes leave together; the shuffles stay inside a row)
    const size_t xo = (size_t)r * ldx, yo = (size_t)r * ldy;
    const int c0 = sl * 8;
    F8 v;
    float s = 0.f;
    if (c0 < C) v = ld8_any(x, x_dt, xo + c0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (c0 + q >= C) v.v[q] = 0.f;
        s += v.v[q];
    }
    const float mu = ln_sub_sum<SW>(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float t = (c0 + q < C) ? v.v[q] - mu : 0.f; qq += t * t; }
    const float rs = rsqrtf(ln_sub_sum<SW>(qq) / (float)C + eps);
    for (int cc = c0; cc < pad_to; cc += SW * 8) {
        F8 o;
        if (cc < C) {
            float wv[8], bv[8];
            ln_wb8(w, cc, C, wv);
            ln_wb8(b, cc, C, bv);
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = (cc + q < C) ? (v.v[q] - mu) * rs * wv[q] + bv[q] : 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = 0.f;
        }
        st8_any(y, y_dt, yo + cc, o);
    }
}
// eight-column accesses are possible when rows, row starts and the padded width are multiples of 8 elements and every 8-chunk that
// holds a valid column lies inside the row allocation
static bool ln_vec8_ok(const void* x, int ldx, int C, const void* y, int ldy, int pad_to) {
    const int C8 = (C + 7) / 8 * 8;
    return ldx % 8 == 0 && ldy % 8 == 0 && pad_to % 8 == 0 && ldx >= C8 && pad_to >= C8 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0;
}
extern "C" int fabind_layernorm_rows(const void* x, int x_dt, int ldx, const float* w, const float* b, float eps, int R, int C,
                                     void* y, int y_dt, int ldy, int pad_to, hipStream_t stream) {
    if (R <= 0) return 0;
    FB_REQUIRE(pad_to <= ldy && C <= ldx, "fabind_layernorm_rows: pad_to <= ldy, C <= ldx");
    FB_REQUIRE(C <= 2048, "fabind_layernorm_rows: C <= 2048");
    if (ln_vec8_ok(x, ldx, C, y, ldy, pad_to) && C <= 256) {
        if (C <= 128)
            hipLaunchKernelGGL((layernorm_rows8s_kernel<16>), dim3((R + 15) / 16), dim3(256), 0, stream, x, x_dt, ldx, w, b, eps, R, C, y, y_dt,
                               ldy, pad_to);
        else
            hipLaunchKernelGGL((layernorm_rows8s_kernel<32>), dim3((R + 7) / 8), dim3(256), 0, stream, x, x_dt, ldx, w, b, eps, R, C, y, y_dt,
                               ldy, pad_to);
        FB_CHECK_LAUNCH();
        return 0;
    }
    if (ln_vec8_ok(x, ldx, C, y, ldy, pad_to)) {
#define LNR8_LAUNCH(NCH_) hipLaunchKernelGGL((layernorm_rows8_kernel<NCH_>), dim3((R + 3) / 4), dim3(256), 0, stream, x, x_dt, ldx, \
                                             w, b, eps, R, C, y, y_dt, ldy, pad_to)
        if (C <= 512) LNR8_LAUNCH(1); else if (C <= 1024) LNR8_LAUNCH(2); else if (C <= 1536) LNR8_LAUNCH(3); else LNR8_LAUNCH(4);
#undef LNR8_LAUNCH
        FB_CHECK_LAUNCH();
        return 0;
    }
#define LNR_LAUNCH(NPL_) hipLaunchKernelGGL((layernorm_rows_kernel<NPL_>), dim3((R + 3) / 4), dim3(256), 0, stream, x, x_dt, ldx, \
                                            w, b, eps, R, C, y, y_dt, ldy, pad_to)
    if (C <= 128) LNR_LAUNCH(2);
    else if (C <= 512) LNR_LAUNCH(8);
    else if (C <= 1024) LNR_LAUNCH(16);
    else LNR_LAUNCH(32);
#undef LNR_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

// Row statistics only (mean, rsqrt(centred variance + eps)): the read half of layernorm_rows, for GEMMs that fold the
// normalisation into their epilogue (FabindGemmArgs.row_mu / row_rs).  One wave per row.  bf16 rows with C % 8 == 0 and
// C <= 2048 are read ONCE with 16-B loads and kept in registers for the centred second moment; other shapes take the
// element-wise two-pass loop.
template <int NCH>
__global__ __launch_bounds__(256) void row_stats_bf16_kernel(const bf16_t* __restrict__ x, int ldx, float eps, int R, int C,
                                                             float* __restrict__ mu, float* __restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const bf16_t* xr = x + (size_t)r * ldx;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c0 = (k * 64 + lane) * 8;
        uint4 u = make_uint4(0u, 0u, 0u, 0u);
        if (c0 < C) u = *(const uint4*)(xr + c0);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[k][q] = __uint_as_float((q & 1) ? (w[q >> 1] & 0xffff0000u) : (w[q >> 1] << 16));
            s += v[k][q];
        }
    }
    const float m = wave_sum(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if ((k * 64 + lane) * 8 < C) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float t = v[k][q] - m; qq += t * t; }
        }
    }
    qq = wave_sum(qq);
    if (lane == 0) { mu[r] = m; rs[r] = rsqrtf(qq / (float)C + eps); }
}
__global__ __launch_bounds__(256) void row_stats_kernel(const void* __restrict__ x, int x_dt, int ldx, float eps, int R, int C,
                                                        float* __restrict__ mu, float* __restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t xo = (size_t)r * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ld_any(x, x_dt, xo + c);
    const float m = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float t = ld_any(x, x_dt, xo + c) - m; q += t * t; }
    q = wave_sum(q);
    if (lane == 0) { mu[r] = m; rs[r] = rsqrtf(q / (float)C + eps); }
}
extern "C" int fabind_row_stats(const void* x, int x_dt, int ldx, float eps, int R, int C, float* mu, float* rs,
                                hipStream_t stream) {
    if (R <= 0) return 0;
    if (x_dt == FB_DT_BF16 && C % 8 == 0 && ldx % 8 == 0 && C <= 2048 && (((uintptr_t)x) & 15) == 0) {
        const dim3 grid((R + 3) / 4);
        if (C <= 512) hipLaunchKernelGGL(row_stats_bf16_kernel<1>, grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, eps, R, C, mu, rs);
        else if (C <= 1024) hipLaunchKernelGGL(row_stats_bf16_kernel<2>, grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, eps, R, C, mu, rs);
        else hipLaunchKernelGGL(row_stats_bf16_kernel<4>, grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, eps, R, C, mu, rs);
        FB_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(row_stats_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, x, x_dt, ldx, eps, R, C, mu, rs);
    FB_CHECK_LAUNCH();
    return 0;
}

// LayerNorm of the concatenated edge input of MC_E_GCL, [h[row] | h[col] | rhohat] (2H+1 columns,
// FABind_plus/fabind/models/egnn.py:52-58 + model_utils.py:44-46), evaluated while gathering: the concatenation is
// never written un-normalised.  One wave per edge; output columns [2H+1, pad_to) are zero.
__global__ __launch_bounds__(256) void edge_ln_concat_kernel(const float* __restrict__ h, int ldh, int H,
                                                             const int* __restrict__ row, const int* __restrict__ col,
                                                             const float* __restrict__ rhohat, const float* __restrict__ w,
                                                             const float* __restrict__ b, float eps, int E, void* y, int y_dt,
                                                             int ldy, int pad_to) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float* hr = h + (size_t)row[e] * ldh;
    const float* hc = h + (size_t)col[e] * ldh;
    const float rh = rhohat[e];
    const int C = 2 * H + 1;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += hr[c] + hc[c];
    const float mu = (wave_sum(s) + rh) / (float)C;
    float v = 0.f;
    for (int c = lane; c < H; c += 64) { const float t0 = hr[c] - mu, t1 = hc[c] - mu; v += t0 * t0 + t1 * t1; }
    const float rs = rsqrtf((wave_sum(v) + (rh - mu) * (rh - mu)) / (float)C + eps);
    const size_t yo = (size_t)e * ldy;
    for (int c = lane; c < H; c += 64) {
        st_any(y, y_dt, yo + c, (hr[c] - mu) * rs * w[c] + b[c]);
        st_any(y, y_dt, yo + H + c, (hc[c] - mu) * rs * w[H + c] + b[H + c]);
    }
    if (lane == 0) st_any(y, y_dt, yo + 2 * H, (rh - mu) * rs * w[2 * H] + b[2 * H]);
    for (int c = C + lane; c < pad_to; c += 64) st_any(y, y_dt, yo + c, 0.f);
}
extern "C" int fabind_edge_ln_concat(const float* h, int ldh, int H, const int* row, const int* col, const float* rhohat,
                                     const float* w, const float* b, float eps, int E, void* y, int y_dt, int ldy, int pad_to,
                                     hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(pad_to <= ldy && 2 * H + 1 <= pad_to, "fabind_edge_ln_concat: 2H+1 <= pad_to <= ldy");
    hipLaunchKernelGGL(edge_ln_concat_kernel, dim3((E + 3) / 4), dim3(256), 0, stream, h, ldh, H, row, col, rhohat, w, b, eps, E,
                       y, y_dt, ldy, pad_to);
    FB_CHECK_LAUNCH();
    return 0;
}

// First Linear of FABind+'s LN-MLP edge model with the LayerNorm FOLDED into per-node projections (bf16 inference):
//   relu( W1 LN([h_r | h_c | rho]) + b1 )
//     = relu( rs_e * (A[row] + B[col] + (m_r - mu_e) c_r + (m_c - mu_e) c_c + (rho - mu_e) w_r) + dvec )
// with W1w = W1 diag(ln_w), [A | B] = (h - m 1^T) [W1w_r | W1w_c]^T evaluated once per NODE on the node-CENTRED
// features (m = per-node feature mean: every term is a deviation, so the bf16 projections are never cancelled against
// a large mean), c_r = W1w_r 1, c_c = W1w_c 1, w_r = W1w[:, 2H], dvec = W1 ln_b + b1.  The row statistics of the
// concatenation come from per-node (mean, centred sum of squares) pairs, combined exactly (parallel-variance formula),
// so the [E, 2H+1] LayerNorm input is never formed and the E x (2H+1) x (2H+1) contraction becomes a gather.  One thread per (edge, 8-column chunk); optional dropout with
// the GEMM epilogue's counter-based mask.
__global__ __launch_bounds__(256) void edge_lnfold_kernel(const bf16_t* __restrict__ AB, int ldab, int Kp, int H,
                                                          const int* __restrict__ row, const int* __restrict__ col,
                                                          const float* __restrict__ rho, const float2* __restrict__ stat,
                                                          float eps, const float* __restrict__ w_r,
                                                          const float* __restrict__ c_r, const float* __restrict__ c_c,
                                                          const float* __restrict__ dvec, int E,
                                                          bf16_t* __restrict__ out, uint32_t thr, float dscale, uint32_t seed) {
    const int CH = Kp >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)E * CH) return;
    const int e = (int)(idx / CH), ch = (int)(idx % CH);
    const int r = row[e], c = col[e];
    const float rh = rho[e];
    const float2 sr = stat[r], sc = stat[c];
    const float Cn = (float)(2 * H + 1), Hf = (float)H;
    const float mu = (Hf * (sr.x + sc.x) + rh) / Cn;
    const float dr = sr.x - mu, dc = sc.x - mu, dq = rh - mu;
    const float rs = rsqrtf((sr.y + sc.y + Hf * (dr * dr + dc * dc) + dq * dq) / Cn + eps);
    const uint4 ua = *(const uint4*)(AB + (size_t)r * ldab + ch * 8);
    const uint4 ub = *(const uint4*)(AB + (size_t)c * ldab + Kp + ch * 8);
    const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float a = __uint_as_float((k & 1) ? (wa[k >> 1] & 0xffff0000u) : (wa[k >> 1] << 16));
        const float b = __uint_as_float((k & 1) ? (wb[k >> 1] & 0xffff0000u) : (wb[k >> 1] << 16));
        const int cc = ch * 8 + k;
        float v = fmaxf(rs * (a + b + dr * c_r[cc] + dc * c_c[cc] + dq * w_r[cc]) + dvec[cc], 0.f);
        if (thr) v *= ((fb_hash32(seed + (uint32_t)e * (uint32_t)Kp + (uint32_t)cc) & 0xffffu) >= thr) ? dscale : 0.f;
        o[k] = v;
    }
    uint4 u;
    u.x = pack2_bf16(o[0], o[1]); u.y = pack2_bf16(o[2], o[3]); u.z = pack2_bf16(o[4], o[5]); u.w = pack2_bf16(o[6], o[7]);
    *(uint4*)(out + idx * 8) = u;
}
// H a multiple of 256 (the production width, H = 512): one wave per edge, lane l owns the 8-column chunks l, l + 64, ...
// of the 2H leading columns with their four per-column vectors held in registers across the wave's grid-stride walk over
// the edges (per edge and lane: 2 NPL gathered 16-B loads, NPL 16-B stores); the Kp - 2H tail columns (rho's column and
// the zero padding) go through the first lanes.  The per-edge scalars are wave-uniform.
#ifndef ELF_U1
#define ELF_U1 4   // edges in flight per wave and iteration, hidden <= 256 (one chunk per lane)
#endif
#ifndef ELF_U2
#define ELF_U2 2   // hidden 512 (two chunks per lane)
#endif
// U edges per wave and iteration (round 5): their indices, then their statistics and gathered rows, are all requested before the first
// is consumed -- a wave that walks one edge at a time spends its time in three dependent round trips per edge (0.6 TB/s at H = 128).
template <int NPL, int U, bool TAIL>
__global__ __launch_bounds__(256) void edge_lnfold_wave_kernel(const bf16_t* __restrict__ AB, int ldab, int Kp, int H,
                                                               const int* __restrict__ row, const int* __restrict__ col,
                                                               const float* __restrict__ rho, const float2* __restrict__ stat,
                                                               float eps, const float* __restrict__ w_r,
                                                               const float* __restrict__ c_r, const float* __restrict__ c_c,
                                                               const float* __restrict__ dvec, int E, bf16_t* __restrict__ out,
                                                               uint32_t thr, float dscale, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    float wr[NPL][8], cr[NPL][8], cc[NPL][8], dv[NPL][8];
    // 2H / 8 leading chunks in the lanes' registers, then rho's column + zero padding through the first lanes -- unless ALL Kp / 8 chunks
    // fit the lanes (H <= 128: 40 chunks): then one pass covers the row (a separate 8-lane tail pass cost as many issue cycles as the
    // main pass: the kernel is VALU-bound at H = 128, 1.5 ms for 1.54 M edges)
    const int tail0 = TAIL ? (H >> 2) : (Kp >> 3), n_tail = (Kp >> 3) - tail0;
    bool on[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        on[k] = k * 64 + lane < tail0;
        const int b = on[k] ? (k * 64 + lane) * 8 : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { wr[k][q] = w_r[b + q]; cr[k][q] = c_r[b + q]; cc[k][q] = c_c[b + q]; dv[k][q] = dvec[b + q]; }
    }
    const float Cn = (float)(2 * H + 1), Hf = (float)H;
    const bool tl = lane < n_tail;
    const int tch = tl ? tail0 + lane : 0;
    for (int e0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * U; e0 < E; e0 += nw * U) {
        int r[U], c[U];
        float rh[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u, E - 1);
            r[u] = row[e]; c[u] = col[e]; rh[u] = rho[e];
        }
        float2 sr[U], sc[U];
        uint4 ua[U][NPL], ub[U][NPL], ta[U], tb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sr[u] = stat[r[u]]; sc[u] = stat[c[u]];
            const bf16_t* ar = AB + (size_t)r[u] * ldab;
            const bf16_t* bc = AB + (size_t)c[u] * ldab + Kp;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int o8 = on[k] ? (k * 64 + lane) * 8 : 0;
                ua[u][k] = *(const uint4*)(ar + o8); ub[u][k] = *(const uint4*)(bc + o8);
            }
            if (TAIL) { ta[u] = *(const uint4*)(ar + tch * 8); tb[u] = *(const uint4*)(bc + tch * 8); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u;
            if (e >= E) break;
            const float mu = (Hf * (sr[u].x + sc[u].x) + rh[u]) / Cn;
            const float dr = sr[u].x - mu, dc = sc[u].x - mu, dq = rh[u] - mu;
            const float rs = rsqrtf((sr[u].y + sc[u].y + Hf * (dr * dr + dc * dc) + dq * dq) / Cn + eps);
            bf16_t* oe = out + (size_t)e * Kp;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const uint32_t wa[4] = {ua[u][k].x, ua[u][k].y, ua[u][k].z, ua[u][k].w}, wb[4] = {ub[u][k].x, ub[u][k].y, ub[u][k].z, ub[u][k].w};
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float a = __uint_as_float((q & 1) ? (wa[q >> 1] & 0xffff0000u) : (wa[q >> 1] << 16));
                    const float b = __uint_as_float((q & 1) ? (wb[q >> 1] & 0xffff0000u) : (wb[q >> 1] << 16));
                    float v = fmaxf(rs * (a + b + dr * cr[k][q] + dc * cc[k][q] + dq * wr[k][q]) + dv[k][q], 0.f);
                    if (thr) v *= ((fb_hash32(seed + (uint32_t)e * (uint32_t)Kp + (uint32_t)((k * 64 + lane) * 8 + q)) & 0xffffu) >= thr) ? dscale : 0.f;
                    o[q] = v;
                }
                uint4 w;
                w.x = pack2_bf16(o[0], o[1]); w.y = pack2_bf16(o[2], o[3]); w.z = pack2_bf16(o[4], o[5]); w.w = pack2_bf16(o[6], o[7]);
                if (on[k]) *(uint4*)(oe + (k * 64 + lane) * 8) = w;
            }
            if (TAIL && tl) {
                const uint32_t wa[4] = {ta[u].x, ta[u].y, ta[u].z, ta[u].w}, wb[4] = {tb[u].x, tb[u].y, tb[u].z, tb[u].w};
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int cq = tch * 8 + q;
                    const float a = __uint_as_float((q & 1) ? (wa[q >> 1] & 0xffff0000u) : (wa[q >> 1] << 16));
                    const float b = __uint_as_float((q & 1) ? (wb[q >> 1] & 0xffff0000u) : (wb[q >> 1] << 16));
                    float v = fmaxf(rs * (a + b + dr * c_r[cq] + dc * c_c[cq] + dq * w_r[cq]) + dvec[cq], 0.f);
                    if (thr) v *= ((fb_hash32(seed + (uint32_t)e * (uint32_t)Kp + (uint32_t)cq) & 0xffffu) >= thr) ? dscale : 0.f;
                    o[q] = v;
                }
                uint4 w;
                w.x = pack2_bf16(o[0], o[1]); w.y = pack2_bf16(o[2], o[3]); w.z = pack2_bf16(o[4], o[5]); w.w = pack2_bf16(o[6], o[7]);
                *(uint4*)(oe + tch * 8) = w;
            }
        }
    }
}
extern "C" int fabind_edge_lnfold(const void* AB, int ldab, int Kp, int H, const int* row, const int* col, const float* rho,
                                  const float* stat, float eps, const float* w_r, const float* c_r, const float* c_c,
                                  const float* dvec, int E, void* out, float p_drop, unsigned seed, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(Kp % 8 == 0 && ldab % 8 == 0 && ldab >= 2 * Kp, "fabind_edge_lnfold: Kp % 8, ldab % 8, ldab >= 2 Kp");
    FB_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "fabind_edge_lnfold: p_drop in [0, 1)");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
#ifndef ELF_WAVE_MIN_H
#define ELF_WAVE_MIN_H 0
#endif
    if (H % 4 == 0 && H <= 512 && H >= ELF_WAVE_MIN_H && Kp / 8 - H / 4 <= 64) {
        const int blocks = (int)std::min<size_t>(((size_t)E + 3) / 4, (size_t)256 * 8);
#define ELF_LAUNCH(NN, UU, TT)                                                                                                      \
        hipLaunchKernelGGL((edge_lnfold_wave_kernel<NN, UU, TT>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)AB, ldab, Kp, H, row, \
                           col, rho, (const float2*)stat, eps, w_r, c_r, c_c, dvec, E, (bf16_t*)out, thr, dscale, (uint32_t)seed)
        if (H > 256) ELF_LAUNCH(2, ELF_U2, true);
        else if (Kp / 8 > 64) ELF_LAUNCH(1, ELF_U1, true);
        else ELF_LAUNCH(1, ELF_U1, false);              // every chunk of the row in one pass (H <= 128)
#undef ELF_LAUNCH
        FB_CHECK_LAUNCH();
        return 0;
    }
    const size_t total = (size_t)E * (Kp / 8);
    hipLaunchKernelGGL(edge_lnfold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)AB, ldab, Kp,
                       H, row, col, rho, (const float2*)stat, eps, w_r, c_r, c_c, dvec, E, (bf16_t*)out, thr, dscale, (uint32_t)seed);
    FB_CHECK_LAUNCH();
    return 0;
}

// Adjoint of fabind_edge_lnfold (round 5: the LayerNorm-folded first edge Linear under autograd, FABind_plus egnn.py:52-58).  With
//   u_k = A[r][k] + B[c][k] + dr c_r[k] + dc c_c[k] + dq w_r[k],   out_k = drop(relu(rs u_k + dvec_k)),
//   mu = (H (m_r + m_c) + rho) / Cn,  dr = m_r - mu,  dc = m_c - mu,  dq = rho - mu,  rs = rsqrt((Q_r + Q_c + H (dr^2 + dc^2) + dq^2) / Cn + eps)
// and g_k = dout_k x [out_k != 0] x 1/(1-p) (the zeros of the saved output are the inactive and the dropped positions):
//   du_k = rs g_k  -> written as bf16 [E, Kp]: the caller sums it over the receiving node (d A) and over the sending node (d B);
//   per edge: d rs = sum_k g_k u_k, d dr = sum_k du_k c_r[k], d dc = sum_k du_k c_c[k], d dq = sum_k du_k w_r[k], folded through rs and mu into
//             es[e] = (d m_r, d Q_r, d m_c, d Q_c, 0, 0, 0, 0), drho[e]   (the caller's segment sums over the receiving / sending node);
//   per column: d dvec = sum_e g, d c_r = sum_e du dr, d c_c = sum_e du dc, d w_r = sum_e du dq  -> part[block][4][Kp] partials.
// u is rebuilt from a second gather of the two projection rows (L2-resident).  One wave per edge, lane l owns the 8-column chunks
// l, l + 64, ... (NPL of them), whose four column accumulators it keeps in registers across the wave's walk over the edges.
#ifndef ELB_U1
#define ELB_U1 4   // edges in flight per wave and iteration: one chunk per lane (Kp <= 512)
#endif
#ifndef ELB_U2
#define ELB_U2 2   // two chunks per lane
#endif
#ifndef ELB_U3
#define ELB_U3 1   // three chunks per lane (Kp = 1088, the production stack: registers)
#endif
#ifndef ELB_MINW
#define ELB_MINW 1
#endif
template <int NPL, int U>
__global__ __launch_bounds__(256, ELB_MINW) void edge_lnfold_bwd_kernel(const bf16_t* __restrict__ AB, int ldab, int Kp, int H,
                                                              const int* __restrict__ row, const int* __restrict__ col,
                                                              const float* __restrict__ rho, const float2* __restrict__ stat, float eps,
                                                              const float* __restrict__ w_r, const float* __restrict__ c_r,
                                                              const float* __restrict__ c_c, const bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ dout, int E, float dscale, bf16_t* __restrict__ du,
                                                              float4* __restrict__ es, float* __restrict__ drho,
                                                              float* __restrict__ part) {
    extern __shared__ float sred[];                                   // [4 waves][4 vectors][Kp] at the end
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nw = gridDim.x * 4, CH = Kp >> 3;
    float a_dv[NPL][8], a_cr[NPL][8], a_cc[NPL][8], a_wr[NPL][8];
    bool on[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        on[k] = k * 64 + lane < CH;
#pragma unroll
        for (int q = 0; q < 8; ++q) { a_dv[k][q] = 0.f; a_cr[k][q] = 0.f; a_cc[k][q] = 0.f; a_wr[k][q] = 0.f; }
    }
    const float Cn = (float)(2 * H + 1), Hf = (float)H;
    for (int e0 = (blockIdx.x * 4 + wv) * U; e0 < E; e0 += nw * U) {
        int r[U], c[U];
        float rh[U];
        uint4 uo[U][NPL], ug[U][NPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u, E - 1);
            r[u] = row[e]; c[u] = col[e]; rh[u] = rho[e];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int o8 = on[k] ? (k * 64 + lane) * 8 : 0;
                uo[u][k] = *(const uint4*)(out + (size_t)e * Kp + o8); ug[u][k] = *(const uint4*)(dout + (size_t)e * Kp + o8);
            }
        }
        float2 sr[U], sc[U];
        uint4 ua[U][NPL], ub[U][NPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sr[u] = stat[r[u]]; sc[u] = stat[c[u]];
            const bf16_t* ar = AB + (size_t)r[u] * ldab;
            const bf16_t* bc = AB + (size_t)c[u] * ldab + Kp;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int o8 = on[k] ? (k * 64 + lane) * 8 : 0;
                ua[u][k] = *(const uint4*)(ar + o8); ub[u][k] = *(const uint4*)(bc + o8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u;
            if (e >= E) break;
            const float mu = (Hf * (sr[u].x + sc[u].x) + rh[u]) / Cn;
            const float dr = sr[u].x - mu, dc = sc[u].x - mu, dq = rh[u] - mu;
            const float rs = rsqrtf((sr[u].y + sc[u].y + Hf * (dr * dr + dc * dc) + dq * dq) / Cn + eps);
            float s_gu = 0.f, s_cr = 0.f, s_cc = 0.f, s_wr = 0.f;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if (on[k]) {
                    const int ch = k * 64 + lane;
                    const uint32_t wa[4] = {ua[u][k].x, ua[u][k].y, ua[u][k].z, ua[u][k].w}, wb[4] = {ub[u][k].x, ub[u][k].y, ub[u][k].z, ub[u][k].w};
                    const uint32_t wo[4] = {uo[u][k].x, uo[u][k].y, uo[u][k].z, uo[u][k].w}, wg[4] = {ug[u][k].x, ug[u][k].y, ug[u][k].z, ug[u][k].w};
                    float d8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int cq = ch * 8 + q;
                        const float a = __uint_as_float((q & 1) ? (wa[q >> 1] & 0xffff0000u) : (wa[q >> 1] << 16));
                        const float b = __uint_as_float((q & 1) ? (wb[q >> 1] & 0xffff0000u) : (wb[q >> 1] << 16));
                        const uint32_t ob = (q & 1) ? (wo[q >> 1] & 0xffff0000u) : (wo[q >> 1] << 16);
                        const float gy = __uint_as_float((q & 1) ? (wg[q >> 1] & 0xffff0000u) : (wg[q >> 1] << 16));
                        const float crk = c_r[cq], cck = c_c[cq], wrk = w_r[cq];
                        const float uu = a + b + dr * crk + dc * cck + dq * wrk;
                        const float g = (ob & 0x7fffffffu) ? gy * dscale : 0.f;
                        const float d_ = rs * g;
                        s_gu += g * uu; s_cr += d_ * crk; s_cc += d_ * cck; s_wr += d_ * wrk;
                        a_dv[k][q] += g; a_cr[k][q] += d_ * dr; a_cc[k][q] += d_ * dc; a_wr[k][q] += d_ * dq;
                        d8[q] = d_;
                    }
                    uint4 o;
                    o.x = pack2_bf16(d8[0], d8[1]); o.y = pack2_bf16(d8[2], d8[3]); o.z = pack2_bf16(d8[4], d8[5]); o.w = pack2_bf16(d8[6], d8[7]);
                    *(uint4*)(du + (size_t)e * Kp + ch * 8) = o;
                }
            }
            s_gu = wave_sum(s_gu); s_cr = wave_sum(s_cr); s_cc = wave_sum(s_cc); s_wr = wave_sum(s_wr);
            if (lane == 0) {
                // rs = (V + eps)^(-1/2), V = (Q_r + Q_c + H (dr^2 + dc^2) + dq^2) / Cn
                const float dV = -0.5f * rs * rs * rs * s_gu;
                const float dQ = dV / Cn;
                const float ddr = s_cr + dV * 2.f * Hf * dr / Cn, ddc = s_cc + dV * 2.f * Hf * dc / Cn, ddq = s_wr + dV * 2.f * dq / Cn;
                const float dmu = -(ddr + ddc + ddq);                  // dr, dc, dq = (m_r, m_c, rho) - mu
                es[(size_t)e * 2] = make_float4(ddr + dmu * Hf / Cn, dQ, ddc + dmu * Hf / Cn, dQ);      // (d m_r, d Q_r, d m_c, d Q_c | 0 0 0 0)
                es[(size_t)e * 2 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                drho[e] = ddq + dmu / Cn;
            }
        }
    }
    // block partials of the four column sums: waves 0 .. 3 in order
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int ch = k * 64 + lane;
            if (ch < CH)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    sred[(wv * 4 + v) * Kp + ch * 8 + q] = v == 0 ? a_dv[k][q] : v == 1 ? a_cr[k][q] : v == 2 ? a_cc[k][q] : a_wr[k][q];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * Kp; i += 256)
        part[(size_t)blockIdx.x * 4 * Kp + i] = ((sred[i] + sred[4 * Kp + i]) + sred[8 * Kp + i]) + sred[12 * Kp + i];
}
extern "C" int fabind_edge_lnfold_bwd_blocks(int E) { return (int)std::min<size_t>(((size_t)E + 3) / 4, (size_t)256 * 4); }
extern "C" int fabind_edge_lnfold_bwd(const void* AB, int ldab, int Kp, int H, const int* row, const int* col, const float* rho,
                                      const float* stat, float eps, const float* w_r, const float* c_r, const float* c_c,
                                      const void* out, const void* dout, int E, float p_drop, void* du, float* es,
                                      float* drho, float* part, int n_blocks, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(Kp % 8 == 0 && ldab % 8 == 0 && ldab >= 2 * Kp && Kp <= 8 * 64 * 3, "fabind_edge_lnfold_bwd: Kp % 8, ldab % 8, ldab >= 2 Kp, Kp <= 1536");
    FB_REQUIRE(p_drop >= 0.f && p_drop < 1.f && n_blocks > 0, "fabind_edge_lnfold_bwd: p_drop in [0, 1), n_blocks > 0");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
    const size_t lds = (size_t)16 * Kp * sizeof(float);
    const int npl = (Kp / 8 + 63) / 64;
#define ELB_LAUNCH(NN, UU)                                                                                                             \
    do {                                                                                                                           \
        /* the attribute is per DEVICE and must cover the instantiation's LARGEST request (NN x 512 columns: 16 x NN x 512 x 4 bytes), not the \
           first call's (a later call with a larger Kp, or a second device, launched beyond the allowance) */                  \
        static unsigned long long set_ = 0ull;                                                                                     \
        int dev_ = 0;                                                                                                              \
        (void)hipGetDevice(&dev_);                                                                                                 \
        if (!((set_ >> (dev_ & 63)) & 1ull)) {                                                                                     \
            (void)hipFuncSetAttribute((const void*)edge_lnfold_bwd_kernel<NN, UU>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * NN * 512 * 4); \
            set_ |= 1ull << (dev_ & 63);                                                                                           \
        }                                                                                                                          \
        hipLaunchKernelGGL((edge_lnfold_bwd_kernel<NN, UU>), dim3(n_blocks), dim3(256), lds, stream, (const bf16_t*)AB, ldab, Kp, H, row, col, rho, \
                           (const float2*)stat, eps, w_r, c_r, c_c, (const bf16_t*)out, (const bf16_t*)dout, E, dscale, (bf16_t*)du,  \
                           (float4*)es, drho, part);                                                                                \
    } while (0)
    if (npl <= 1) ELB_LAUNCH(1, ELB_U1); else if (npl == 2) ELB_LAUNCH(2, ELB_U2); else ELB_LAUNCH(3, ELB_U3);
#undef ELB_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

// coord_mlp of FABind+'s inter-edge attention (egnn.py:277-300: LN -> Linear -> relu -> [dropout] -> bias-free Linear to
// one scalar) on v_e = V[col] + rho * w_rv, with the LayerNorm folded into a per-node projection (bf16 inference):
//   s[e] = sum_k w3[k] * drop(relu( rs_e * (P[col[e], k] + rho[e] * u[k]) + d[k] ))
// P = (V - mean(V) 1^T) (W1 diag(ln_w))^T per NODE, u = W1 diag(ln_w) (w_rv - mean(w_rv)), d = W1 ln_b + b1, and
//   rs_e = rsqrt( (q_V[col] + 2 rho (Vc[col] . wc) + rho^2 q_w) / H + eps )      (stat[n] = (q_V[n], Vc[n] . wc))
// so the per-edge [E, H] value rows, their LayerNorm and the E x H x H contraction are never formed.  Wave per edge.
__global__ __launch_bounds__(256) void inter_coord_fold_kernel(const bf16_t* __restrict__ P, int ldp, int H,
                                                               const int* __restrict__ col, const float* __restrict__ rho,
                                                               const float2* __restrict__ stat, float q_w, float eps,
                                                               const float* __restrict__ u, const float* __restrict__ d,
                                                               const float* __restrict__ w3, int E, float* __restrict__ s_out,
                                                               uint32_t thr, float dscale, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    const bool act = lane * 8 < H;                              // H <= 512: one 8-column chunk per lane
    float uu[8], dd[8], ww[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uu[q] = act ? u[lane * 8 + q] : 0.f; dd[q] = act ? d[lane * 8 + q] : 0.f; ww[q] = act ? w3[lane * 8 + q] : 0.f;
    }
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += nw) {
        const int c = col[e];
        const float rh = rho[e];
        const float2 st = stat[c];
        const float rs = rsqrtf(fmaxf(st.x + 2.f * rh * st.y + rh * rh * q_w, 0.f) / (float)H + eps);
        float acc = 0.f;
        if (act) {
            const uint4 up = *(const uint4*)(P + (size_t)c * ldp + lane * 8);
            const uint32_t wp[4] = {up.x, up.y, up.z, up.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float pv = __uint_as_float((q & 1) ? (wp[q >> 1] & 0xffff0000u) : (wp[q >> 1] << 16));
                float v = fmaxf(rs * (pv + rh * uu[q]) + dd[q], 0.f);
                if (thr) v *= ((fb_hash32(seed + (uint32_t)e * (uint32_t)H + (uint32_t)(lane * 8 + q)) & 0xffffu) >= thr) ? dscale : 0.f;
                acc += v * ww[q];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) s_out[e] = acc;
    }
}
extern "C" int fabind_inter_coord_fold(const void* P, int ldp, int H, const int* col, const float* rho, const float* stat,
                                       float q_w, float eps, const float* u, const float* d, const float* w3, int E,
                                       float* s_out, float p_drop, unsigned seed, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(H % 8 == 0 && H <= 512 && ldp % 8 == 0, "fabind_inter_coord_fold: H % 8 == 0, H <= 512, ldp % 8 == 0");
    FB_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "fabind_inter_coord_fold: p_drop in [0, 1)");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
    const int blocks = (int)std::min<size_t>(((size_t)E + 3) / 4, (size_t)256 * 8);
    hipLaunchKernelGGL(inter_coord_fold_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)P, ldp, H, col, rho,
                       (const float2*)stat, q_w, eps, u, d, w3, E, s_out, thr, dscale, (uint32_t)seed);
    FB_CHECK_LAUNCH();
    return 0;
}

// adjoint of layernorm_rows: one wave per row, row values in registers; dw / db partials per work-group through LDS
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                                 const float* __restrict__ w, const void* __restrict__ dy,
                                                                 int dy_dt, int lddy, float eps, int R, int C, void* dx,
                                                                 int dx_dt, int lddx, float* dwp, float* dbp) {
    // One wave per row, rows strided over the grid; the weight / bias gradient of a wave's rows accumulates in REGISTERS (one column set
    // per lane) and the four waves of a work-group are added in wave order through LDS at the end: one [C] partial per work-group and a
    // fixed summation order.  (Round 1 zeroed an LDS accumulator per work-group of FOUR rows, added into it with LDS atomics and wrote a
    // [C] partial per four rows -- as many partial bytes as input bytes, 25 % of the kernel time of a FABind+ training step.)
    extern __shared__ float sh[];                   // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aw[NPL], ab[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { aw[k] = 0.f; ab[k] = 0.f; }
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const size_t xo = (size_t)r * ldx, go = (size_t)r * lddy, o = (size_t)r * lddx;
        float v[NPL], g[NPL];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int c = k * 64 + lane;
            v[k] = c < C ? ld_any(x, x_dt, xo + c) : 0.f;
            g[k] = c < C ? ld_any(dy, dy_dt, go + c) : 0.f;
            s += v[k];
        }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int c = k * 64 + lane;
            const float t = c < C ? v[k] - mu : 0.f;
            q += t * t;
        }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int c = k * 64 + lane;
            if (c < C) {
                const float xh = (v[k] - mu) * rs, gw = g[k] * w[c];
                s1 += gw; s2 += gw * xh;
            }
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int c = k * 64 + lane;
            if (c < C) {
                const float xh = (v[k] - mu) * rs, gw = g[k] * w[c];
                st_any(dx, dx_dt, o + c, rs * (gw - s1 - xh * s2));
                aw[k] += g[k] * xh;
                ab[k] += g[k];
            }
        }
    }
    for (int wv = 0; wv < 4; ++wv) {                // waves add in order: no atomics, fixed order
        if (wave == wv) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int c = k * 64 + lane;
                if (c < C) {
                    sh[c] = (wv ? sh[c] : 0.f) + aw[k];
                    sh[C + c] = (wv ? sh[C + c] : 0.f) + ab[k];
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = sh[c];
        dbp[(size_t)blockIdx.x * C + c] = sh[C + c];
    }
}
// the adjoint with eight consecutive columns per lane (see layernorm_rows8_kernel): same reduction structure and summation order per
// column as layernorm_rows_bwd_kernel (rows strided over the grid, register accumulators, waves added in order)
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_rows_bwd8_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                                  const float* __restrict__ w, const void* __restrict__ dy,
                                                                  int dy_dt, int lddy, float eps, int R, int C, void* dx,
                                                                  int dx_dt, int lddx, float* dwp, float* dbp) {
    extern __shared__ float sh[];                   // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aw[NCH][8], ab[NCH][8], wv[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        ln_wb8(w, (k * 64 + lane) * 8, C, wv[k]);
#pragma unroll
        for (int q = 0; q < 8; ++q) { aw[k][q] = 0.f; ab[k][q] = 0.f; }
    }
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const size_t xo = (size_t)r * ldx, go = (size_t)r * lddy, o = (size_t)r * lddx;
        F8 v[NCH], g[NCH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c0 = (k * 64 + lane) * 8;
            if (c0 < C) { v[k] = ld8_any(x, x_dt, xo + c0); g[k] = ld8_any(dy, dy_dt, go + c0); }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (c0 + q >= C) { v[k].v[q] = 0.f; g[k].v[q] = 0.f; }
                s += v[k].v[q];
            }
        }
        const float mu = wave_sum(s) / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c0 = (k * 64 + lane) * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float t = (c0 + q < C) ? v[k].v[q] - mu : 0.f; qq += t * t; }
        }
        const float rs = rsqrtf(wave_sum(qq) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c0 = (k * 64 + lane) * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (c0 + q < C) {
                    const float xh = (v[k].v[q] - mu) * rs, gw = g[k].v[q] * wv[k][q];
                    s1 += gw; s2 += gw * xh;
                }
            }
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c0 = (k * 64 + lane) * 8;
            if (c0 >= C) continue;
            F8 d;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xh = (v[k].v[q] - mu) * rs, gw = g[k].v[q] * wv[k][q];
                d.v[q] = (c0 + q < C) ? rs * (gw - s1 - xh * s2) : 0.f;
                if (c0 + q < C) { aw[k][q] += g[k].v[q] * xh; ab[k][q] += g[k].v[q]; }
            }
            st8_any(dx, dx_dt, o + c0, d);
        }
    }
    for (int wv_ = 0; wv_ < 4; ++wv_) {             // waves add in order: no atomics, fixed order
        if (wave == wv_) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int c0 = (k * 64 + lane) * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = c0 + q;
                    if (c < C) {
                        sh[c] = (wv_ ? sh[c] : 0.f) + aw[k][q];
                        sh[C + c] = (wv_ ? sh[C + c] : 0.f) + ab[k][q];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = sh[c];
        dbp[(size_t)blockIdx.x * C + c] = sh[C + c];
    }
}
// the adjoint for narrow rows (see layernorm_rows8s_kernel): 64 / SW rows per wave; a wave's rows are combined with shuffles before
// the waves add in order
template <int SW>
__global__ __launch_bounds__(256) void layernorm_rows_bwd8s_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                                   const float* __restrict__ w, const void* __restrict__ dy,
                                                                   int dy_dt, int lddy, float eps, int R, int C, void* dx,
                                                                   int dx_dt, int lddx, float* dwp, float* dbp) {
    extern __shared__ float sh[];                   // [2][C]
    constexpr int RW = 64 / SW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane % SW, sr = lane / SW;
    const int c0 = sl * 8;
    float aw[8], ab[8], wv[8];
    ln_wb8(w, c0, C, wv);
#pragma unroll
    for (int q = 0; q < 8; ++q) { aw[q] = 0.f; ab[q] = 0.f; }
    for (int r = (blockIdx.x * 4 + wave) * RW + sr; r < R; r += gridDim.x * 4 * RW) {
        const size_t xo = (size_t)r * ldx, go = (size_t)r * lddy, o = (size_t)r * lddx;
        F8 v, g;
        float s = 0.f;
        if (c0 < C) { v = ld8_any(x, x_dt, xo + c0); g = ld8_any(dy, dy_dt, go + c0); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (c0 + q >= C) { v.v[q] = 0.f; g.v[q] = 0.f; }
            s += v.v[q];
        }
        const float mu = ln_sub_sum<SW>(s) / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float t = (c0 + q < C) ? v.v[q] - mu : 0.f; qq += t * t; }
        const float rs = rsqrtf(ln_sub_sum<SW>(qq) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (c0 + q < C) {
                const float xh = (v.v[q] - mu) * rs, gw = g.v[q] * wv[q];
                s1 += gw; s2 += gw * xh;
            }
        }
        s1 = ln_sub_sum<SW>(s1) / (float)C; s2 = ln_sub_sum<SW>(s2) / (float)C;
        if (c0 < C) {
            F8 d;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xh = (v.v[q] - mu) * rs, gw = g.v[q] * wv[q];
                d.v[q] = (c0 + q < C) ? rs * (gw - s1 - xh * s2) : 0.f;
                if (c0 + q < C) { aw[q] += g.v[q] * xh; ab[q] += g.v[q]; }
            }
            st8_any(dx, dx_dt, o + c0, d);
        }
    }
#pragma unroll
    for (int m = SW; m < 64; m <<= 1)               // the wave's 64 / SW rows: fixed-order tree
#pragma unroll
        for (int q = 0; q < 8; ++q) { aw[q] += __shfl_xor(aw[q], m, 64); ab[q] += __shfl_xor(ab[q], m, 64); }
    for (int wv_ = 0; wv_ < 4; ++wv_) {             // waves add in order: no atomics, fixed order
        if (wave == wv_ && sr == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = c0 + q;
                if (c < C) {
                    sh[c] = (wv_ ? sh[c] : 0.f) + aw[q];
                    sh[C + c] = (wv_ ? sh[C + c] : 0.f) + ab[q];
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = sh[c];
        dbp[(size_t)blockIdx.x * C + c] = sh[C + c];
    }
}
extern "C" int fabind_layernorm_rows_bwd(const void* x, int x_dt, int ldx, const float* w, const void* dy, int dy_dt, int lddy,
                                         float eps, int R, int C, void* dx, int dx_dt, int lddx, float* dw_part, float* db_part,
                                         int nblk, hipStream_t stream) {
    if (R <= 0) return 0;
    FB_REQUIRE(C <= 2048, "fabind_layernorm_rows_bwd: C <= 2048");
    FB_REQUIRE(nblk >= 1, "fabind_layernorm_rows_bwd: nblk >= 1 (rows of the dw / db partial buffers)");
    const size_t lds = (size_t)2 * C * sizeof(float);
    const int C8 = (C + 7) / 8 * 8;
    if (ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldx >= C8 && lddy >= C8 && lddx >= C8 && (((uintptr_t)x) & 15) == 0 &&
        (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)dx) & 15) == 0 && C <= 1536) {
        if (C <= 256) {
            if (C <= 128)
                hipLaunchKernelGGL((layernorm_rows_bwd8s_kernel<16>), dim3(nblk), dim3(256), lds, stream, x, x_dt, ldx, w, dy, dy_dt, lddy, eps, R,
                                   C, dx, dx_dt, lddx, dw_part, db_part);
            else
                hipLaunchKernelGGL((layernorm_rows_bwd8s_kernel<32>), dim3(nblk), dim3(256), lds, stream, x, x_dt, ldx, w, dy, dy_dt, lddy, eps, R,
                                   C, dx, dx_dt, lddx, dw_part, db_part);
            FB_CHECK_LAUNCH();
            return 0;
        }
#define LNB8_LAUNCH(NCH_) hipLaunchKernelGGL((layernorm_rows_bwd8_kernel<NCH_>), dim3(nblk), dim3(256), lds, stream, x, x_dt, ldx, w, dy, \
                                             dy_dt, lddy, eps, R, C, dx, dx_dt, lddx, dw_part, db_part)
        if (C <= 512) LNB8_LAUNCH(1); else if (C <= 1024) LNB8_LAUNCH(2); else LNB8_LAUNCH(3);
#undef LNB8_LAUNCH
        FB_CHECK_LAUNCH();
        return 0;
    }
#define LNB_LAUNCH(NPL_) hipLaunchKernelGGL((layernorm_rows_bwd_kernel<NPL_>), dim3(nblk), dim3(256), lds, stream, x, x_dt, \
                                            ldx, w, dy, dy_dt, lddy, eps, R, C, dx, dx_dt, lddx, dw_part, db_part)
    if (C <= 128) LNB_LAUNCH(2);
    else if (C <= 512) LNB_LAUNCH(8);
    else if (C <= 1024) LNB_LAUNCH(16);
    else if (C <= 1280) LNB_LAUNCH(20);   /* the FABind+ edge input (2H + 1 = 1025 padded to 1088 at H = 512): 32 column sets per lane left one wave per SIMD */
    else LNB_LAUNCH(32);
#undef LNB_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void edge_concat_kernel(const float* __restrict__ h, int ldh, int H, const int* __restrict__ row,
                                                          const int* __restrict__ col, const float* __restrict__ rhohat, int E,
                                                          void* y, int y_dt, int ldy, int pad_to) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float* hr = h + (size_t)row[e] * ldh;
    const float* hc = h + (size_t)col[e] * ldh;
    const size_t yo = (size_t)e * ldy;
    if (y_dt == FB_DT_BF16 && H % 8 == 0 && ldh % 4 == 0 && ldy % 8 == 0 && (((uintptr_t)h) & 15) == 0 && (((uintptr_t)y) & 15) == 0) {
        // eight columns per lane: two 16-byte fp32 loads -> one 16-byte bf16 store (the scalar loop below moved the 491 MB of the
        // FABind+ edge input at 1.2 TB/s: 398 us per launch, 8 launches per training step)
        bf16_t* yb = (bf16_t*)y + yo;
        for (int c = lane * 8; c < 2 * H; c += 512) {
            const float* src = c < H ? hr + c : hc + (c - H);
            const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
            *(uint4*)(yb + c) = make_uint4(pack2_bf16(a.x, a.y), pack2_bf16(a.z, a.w), pack2_bf16(b.x, b.y), pack2_bf16(b.z, b.w));
        }
        if (lane == 0) yb[2 * H] = f32_to_bf16(rhohat[e]);
        for (int c = 2 * H + 1 + lane; c < pad_to; c += 64) yb[c] = f32_to_bf16(0.f);
        return;
    }
    for (int c = lane; c < H; c += 64) {
        st_any(y, y_dt, yo + c, hr[c]);
        st_any(y, y_dt, yo + H + c, hc[c]);
    }
    if (lane == 0) st_any(y, y_dt, yo + 2 * H, rhohat[e]);
    for (int c = 2 * H + 1 + lane; c < pad_to; c += 64) st_any(y, y_dt, yo + c, 0.f);
}
extern "C" int fabind_edge_concat(const float* h, int ldh, int H, const int* row, const int* col, const float* rhohat, int E,
                                  void* y, int y_dt, int ldy, int pad_to, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(pad_to <= ldy && 2 * H + 1 <= pad_to, "fabind_edge_concat: 2H+1 <= pad_to <= ldy");
    hipLaunchKernelGGL(edge_concat_kernel, dim3((E + 3) / 4), dim3(256), 0, stream, h, ldh, H, row, col, rhohat, E, y, y_dt, ldy,
                       pad_to);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Many small strided 2-D copies with dtype conversion in ONE launch: the differentiable parameter pack of a model call (slices /
// concatenations / zero padding / casts of the nn.Module parameters into the kernels' operand layouts, and its adjoint) was ~450 ATen
// launches of a few microseconds per training step -- a quarter of all launches at the bench shape, half at the pocket shape.
// Segment s: dst[r * dst_sr + c * dst_sc] = convert(src[r * src_sr + c * src_sc]) for r < rows, c < cols; src == NULL writes zeros.
// blockIdx.x = segment, blockIdx.y strides over its elements.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multi_copy_kernel(const FabindCopySeg* __restrict__ segs) {
    const FabindCopySeg g = segs[blockIdx.x];
    const long total = (long)g.rows * g.cols;
    // four consecutive columns per thread when both sides are column-contiguous and aligned for 4-element accesses
    if (g.vec4) {
        const int c4 = g.cols >> 2;
        const long tot4 = (long)g.rows * c4;
        for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < tot4; i += (long)gridDim.y * 256) {
            const long r = i / c4, c = (i % c4) * 4;
            const float4 v = g.src ? ld4_any(g.src, g.src_dt, (size_t)(r * g.src_sr + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
            st4_any(g.dst, g.dst_dt, (size_t)(r * g.dst_sr + c), v);
        }
        return;
    }
    if (g.src && g.src_sr == 1 && g.src_sc > 1 && g.dst_sc == 1 && g.rows >= 32 && g.cols >= 32) {
        // a transposed view of a row-major parameter (the W^T copies the input-gradient GEMMs read): the element-wise loop below reads
        // one 4-byte element per 64-byte line (393 us for the 340 segments of the bench model's pack, a third of it here); 64 x 64
        // tiles through LDS read and write whole lines
        __shared__ float tile[64][65];
        const int tr = (g.rows + 63) >> 6, tc = (g.cols + 63) >> 6;
        const int a = threadIdx.x & 63, b0 = threadIdx.x >> 6;
        for (int t = blockIdx.y; t < tr * tc; t += gridDim.y) {
            const int r0 = (t / tc) << 6, c0 = (t % tc) << 6;
#pragma unroll 4
            for (int b = b0; b < 64; b += 4) {           // source: contiguous along the destination ROW index
                const int r = r0 + a, c = c0 + b;
                tile[b][a] = (r < g.rows && c < g.cols) ? ld_any(g.src, g.src_dt, (size_t)((long)c * g.src_sc + r)) : 0.f;
            }
            __syncthreads();
#pragma unroll 4
            for (int b = b0; b < 64; b += 4) {           // destination: contiguous along its column index
                const int r = r0 + b, c = c0 + a;
                if (r < g.rows && c < g.cols) st_any(g.dst, g.dst_dt, (size_t)((long)r * g.dst_sr + c), tile[a][b]);
            }
            __syncthreads();
        }
        return;
    }
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const long r = i / g.cols, c = i % g.cols;
        const float v = g.src ? ld_any(g.src, g.src_dt, (size_t)(r * g.src_sr + c * g.src_sc)) : 0.f;
        st_any(g.dst, g.dst_dt, (size_t)(r * g.dst_sr + c * g.dst_sc), v);
    }
}

extern "C" int fabind_multi_copy(const FabindCopySeg* segs_dev, int n_segs, int blocks_per_seg, hipStream_t stream) {
    if (n_segs <= 0) return 0;
    FB_REQUIRE(segs_dev != nullptr && blocks_per_seg >= 1 && blocks_per_seg <= 65535, "fabind_multi_copy: segment table / blocks per segment");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(n_segs, blocks_per_seg), dim3(256), 0, stream, segs_dev);
    FB_CHECK_LAUNCH();
    return 0;
}
