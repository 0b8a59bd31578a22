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
