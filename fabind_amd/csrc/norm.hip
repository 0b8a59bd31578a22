// LayerNorm over the last dimension (one wave per row), forward and backward.
// Replaces torch.nn.LayerNorm in Transition_diff_out_dim and the dist-map head
// (reference models/model.py:15,22,29,352-353).
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
    extern __shared__ float sh[];  // [2][C] block partials
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 2 * C; c += 256) sh[c] = 0.f;
    __syncthreads();
    const int r = blockIdx.x * 4 + wv;
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
            atomicAdd(&sh[c], gp[c] * xh);          // LDS atomics, 4 waves per block: order-insensitive to 1 ulp
            atomicAdd(&sh[C + c], gp[c]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = sh[c];
        dbp[(size_t)blockIdx.x * C + c] = sh[C + c];
    }
}

extern "C" int fabind_layernorm_bwd(const float* x, const float* w, const float* dy, const float* mean,
                                    const float* rstd, int R, int C, float* dx, float* dw_part, float* db_part,
                                    hipStream_t stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 2 * C * sizeof(float), stream, x, w, dy, mean,
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
extern "C" int fabind_layernorm_rows(const void* x, int x_dt, int ldx, const float* w, const float* b, float eps, int R, int C,
                                     void* y, int y_dt, int ldy, int pad_to, hipStream_t stream) {
    if (R <= 0) return 0;
    FB_REQUIRE(pad_to <= ldy && C <= ldx, "fabind_layernorm_rows: pad_to <= ldy, C <= ldx");
    FB_REQUIRE(C <= 2048, "fabind_layernorm_rows: C <= 2048");
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

// adjoint of layernorm_rows: one wave per row, row values in registers; dw / db partials per work-group through LDS
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const void* __restrict__ x, int x_dt, int ldx,
                                                                 const float* __restrict__ w, const void* __restrict__ dy,
                                                                 int dy_dt, int lddy, float eps, int R, int C, void* dx,
                                                                 int dx_dt, int lddx, float* dwp, float* dbp) {
    extern __shared__ float sh[];                   // [2][C]
    const int lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < 2 * C; c += 256) sh[c] = 0.f;
    __syncthreads();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r < R) {
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
                atomicAdd(&sh[c], g[k] * xh);
                atomicAdd(&sh[C + c], g[k]);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        dwp[(size_t)blockIdx.x * C + c] = sh[c];
        dbp[(size_t)blockIdx.x * C + c] = sh[C + c];
    }
}
extern "C" int fabind_layernorm_rows_bwd(const void* x, int x_dt, int ldx, const float* w, const void* dy, int dy_dt, int lddy,
                                         float eps, int R, int C, void* dx, int dx_dt, int lddx, float* dw_part, float* db_part,
                                         hipStream_t stream) {
    if (R <= 0) return 0;
    FB_REQUIRE(C <= 2048, "fabind_layernorm_rows_bwd: C <= 2048");
    const size_t lds = (size_t)2 * C * sizeof(float);
#define LNB_LAUNCH(NPL_) hipLaunchKernelGGL((layernorm_rows_bwd_kernel<NPL_>), dim3((R + 3) / 4), dim3(256), lds, stream, x, x_dt, \
                                            ldx, w, dy, dy_dt, lddy, eps, R, C, dx, dx_dt, lddx, dw_part, db_part)
    if (C <= 128) LNB_LAUNCH(2);
    else if (C <= 512) LNB_LAUNCH(8);
    else if (C <= 1024) LNB_LAUNCH(16);
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
