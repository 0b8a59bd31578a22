// Backward kernels of the FABind layer ops (gfx950).  Each kernel is the hand-written adjoint of a
// forward kernel in gcl.hip / attn.hip / graph.hip; contractions in backward passes reuse the GEMM family.
// Reductions over a CSR row, a key tile or a chunk of rows are done in fixed order (no float atomics)
// unless stated in the kernel's comment.
#include "common.h"
#include "fabind_hip.h"

// ------------------------------------------------------------------------------------------------
// out = dy * act'(y)   (elementwise; for ReLU the derivative can be taken at the output)
// ------------------------------------------------------------------------------------------------
__global__ void mul_dact_kernel(const void* dy, int dy_dt, const void* y, int y_dt, int act, void* out, int out_dt, long n, float scale) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {   // 4 elements per thread (8/16-byte accesses)
        float4 g = ld4_any(dy, dy_dt, i), v = ld4_any(y, y_dt, i);
        st4_any(out, out_dt, i, make_float4(scale * g.x * apply_dact(v.x, act), scale * g.y * apply_dact(v.y, act),
                                            scale * g.z * apply_dact(v.z, act), scale * g.w * apply_dact(v.w, act)));
    } else {
        for (; i < n; ++i) st_any(out, out_dt, i, scale * ld_any(dy, dy_dt, i) * apply_dact(ld_any(y, y_dt, i), act));
    }
}
extern "C" int fabind_mul_dact(const void* dy, int dy_dt, const void* y, int y_dt, int act, void* out, int out_dt, long n, float scale,
                               hipStream_t stream) {
    if (n <= 0) return 0;
    FB_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)out % 16 == 0), "fabind_mul_dact: 16-byte alignment");
    long nt = (n + 3) / 4;
    hipLaunchKernelGGL(mul_dact_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, stream, dy, dy_dt, y, y_dt, act,
                       out, out_dt, n, scale);
    FB_CHECK_LAUNCH();
    return 0;
}

// out = dy * act'(y) AND its column sums (the bias gradient) in the same pass over dy: one block = 256 columns x a
// chunk of rows, 4 row-lanes with 4 independent row loads in flight, fixed-order combine (deterministic).
// DROP: out = dy * mask(seed, r, c) / (1 - p) with the counter-based dropout mask of the GEMM epilogues (gemm.hip: C = R + drop(A W^T + b))
// regenerated from its key -- the adjoint of an epilogue dropout that is followed by a residual (the zeros of the saved output do not
// tell the mask there); `scale` = 1 / (1 - p), y is not read.
template <bool DROP>
__global__ __launch_bounds__(256) void mul_dact_colsum_kernel(const void* __restrict__ dy, int dy_dt, const void* __restrict__ y,
                                                              int y_dt, int act, void* __restrict__ out, int out_dt, int R, int C,
                                                              float* scratch, int rows_per, float scale, uint32_t thr, uint32_t seed) {
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {                                     // C % 4 == 0: the lane's 4 columns are all inside
        int r = r0 + q;
        for (; r + 12 < r1; r += 16) {
            float4 g[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = ld4_any(dy, dy_dt, (size_t)(r + 4 * u) * C + c);
                v[u] = y ? ld4_any(y, y_dt, (size_t)(r + 4 * u) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                if (DROP) {
                    const uint32_t k0 = seed + (uint32_t)(r + 4 * u) * (uint32_t)C + (uint32_t)c;
                    o = make_float4(((fb_hash32(k0) & 0xffffu) >= thr) ? scale * g[u].x : 0.f, ((fb_hash32(k0 + 1u) & 0xffffu) >= thr) ? scale * g[u].y : 0.f,
                                    ((fb_hash32(k0 + 2u) & 0xffffu) >= thr) ? scale * g[u].z : 0.f, ((fb_hash32(k0 + 3u) & 0xffffu) >= thr) ? scale * g[u].w : 0.f);
                } else {
                    o = make_float4(scale * g[u].x * apply_dact(v[u].x, act), scale * g[u].y * apply_dact(v[u].y, act),
                                    scale * g[u].z * apply_dact(v[u].z, act), scale * g[u].w * apply_dact(v[u].w, act));
                }
                st4_any(out, out_dt, (size_t)(r + 4 * u) * C + c, o);
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
        }
        for (; r < r1; r += 4) {
            const float4 g = ld4_any(dy, dy_dt, (size_t)r * C + c);
            const float4 v = y ? ld4_any(y, y_dt, (size_t)r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            if (DROP) {
                const uint32_t k0 = seed + (uint32_t)r * (uint32_t)C + (uint32_t)c;
                o = make_float4(((fb_hash32(k0) & 0xffffu) >= thr) ? scale * g.x : 0.f, ((fb_hash32(k0 + 1u) & 0xffffu) >= thr) ? scale * g.y : 0.f,
                                ((fb_hash32(k0 + 2u) & 0xffffu) >= thr) ? scale * g.z : 0.f, ((fb_hash32(k0 + 3u) & 0xffffu) >= thr) ? scale * g.w : 0.f);
            } else {
                o = make_float4(scale * g.x * apply_dact(v.x, act), scale * g.y * apply_dact(v.y, act),
                                scale * g.z * apply_dact(v.z, act), scale * g.w * apply_dact(v.w, act));
            }
            st4_any(out, out_dt, (size_t)r * C + c, o);
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
    }
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        const float4 a = part[0][lane], b = part[1][lane], d_ = part[2][lane], e = part[3][lane];
        float* o = scratch + (size_t)blockIdx.y * C + c;
        o[0] = (a.x + b.x) + (d_.x + e.x); o[1] = (a.y + b.y) + (d_.y + e.y);
        o[2] = (a.z + b.z) + (d_.z + e.z); o[3] = (a.w + b.w) + (d_.w + e.w);
    }
}
#define SUMC_BLOCK 1024
#define SUMC_GRID(C_) dim3(((C_) + 63) / 64)
__global__ void sum_chunks_kernel(const float* __restrict__ scratch, float* out, int C, int nchunk);
extern "C" int fabind_mul_dact_colsum(const void* dy, int dy_dt, const void* y, int y_dt, int act, void* out, int out_dt,
                                      int R, int C, float* colsum, float* scratch, int nchunk, float scale, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    FB_REQUIRE(C % 4 == 0 && nchunk >= 1, "fabind_mul_dact_colsum: C % 4 == 0, nchunk >= 1");
    FB_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "fabind_mul_dact_colsum: 16-byte alignment");
    const int rows_per = (R + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(mul_dact_colsum_kernel<false>, dim3((C + 255) / 256, nchunk), dim3(256), 0, stream, dy, dy_dt, y, y_dt, act, out,
                       out_dt, R, C, scratch, rows_per, scale, 0u, 0u);
    hipLaunchKernelGGL(sum_chunks_kernel, SUMC_GRID(C), dim3(SUMC_BLOCK), 0, stream, scratch, colsum, C, nchunk);
    FB_CHECK_LAUNCH();
    return 0;
}
// out [R, C] = dy * mask / (1 - p) (+ its column sums): the adjoint of the GEMM epilogues' dropout (fabind_gemm with p_drop > 0, key
// seed + r * C + c), for the epilogues whose saved output does not show the mask (dropout ahead of a residual, or with no activation)
extern "C" int fabind_mul_dropmask_colsum(const void* dy, int dy_dt, void* out, int out_dt, int R, int C, float p_drop, unsigned seed,
                                          float* colsum, float* scratch, int nchunk, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    FB_REQUIRE(C % 4 == 0 && nchunk >= 1, "fabind_mul_dropmask_colsum: C % 4 == 0, nchunk >= 1");
    FB_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)out % 16 == 0), "fabind_mul_dropmask_colsum: 16-byte alignment");
    FB_REQUIRE(p_drop > 0.f && p_drop < 1.f, "fabind_mul_dropmask_colsum: p_drop in (0, 1)");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float scale = 1.0f / (1.0f - (float)thr / 65536.0f);
    const int rows_per = (R + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(mul_dact_colsum_kernel<true>, dim3((C + 255) / 256, nchunk), dim3(256), 0, stream, dy, dy_dt, nullptr, 0, FB_ACT_NONE, out,
                       out_dt, R, C, scratch, rows_per, scale, thr, (uint32_t)seed);
    hipLaunchKernelGGL(sum_chunks_kernel, SUMC_GRID(C), dim3(SUMC_BLOCK), 0, stream, scratch, colsum, C, nchunk);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out = h + drop(hn - h) and its adjoint (train mode: nn.Dropout on the aggregated inter-edge attention message ahead of the residual,
// FABind/fabind/models/egnn.py:236) -- one pass each way with the counter-based mask of the GEMM epilogues keyed by (seed, linear index);
// torch ran sub -> native_dropout -> add and their three adjoints over the [N, H] fp32 tensors.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void drop_mix_kernel(const float4* __restrict__ h, const float4* __restrict__ hn, float4* __restrict__ out,
                                                       long n4, uint32_t thr, float scale, uint32_t seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = h[i], b = hn[i];
    const uint32_t k0 = seed + (uint32_t)(i * 4);
    float4 o;
    o.x = a.x + (((fb_hash32(k0) & 0xffffu) >= thr) ? scale * (b.x - a.x) : 0.f);
    o.y = a.y + (((fb_hash32(k0 + 1u) & 0xffffu) >= thr) ? scale * (b.y - a.y) : 0.f);
    o.z = a.z + (((fb_hash32(k0 + 2u) & 0xffffu) >= thr) ? scale * (b.z - a.z) : 0.f);
    o.w = a.w + (((fb_hash32(k0 + 3u) & 0xffffu) >= thr) ? scale * (b.w - a.w) : 0.f);
    out[i] = o;
}
__global__ __launch_bounds__(256) void drop_mix_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dhn, float4* __restrict__ dh,
                                                           long n4, uint32_t thr, float scale, uint32_t seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = g[i];
    const uint32_t k0 = seed + (uint32_t)(i * 4);
    float4 d;
    d.x = ((fb_hash32(k0) & 0xffffu) >= thr) ? scale * a.x : 0.f;
    d.y = ((fb_hash32(k0 + 1u) & 0xffffu) >= thr) ? scale * a.y : 0.f;
    d.z = ((fb_hash32(k0 + 2u) & 0xffffu) >= thr) ? scale * a.z : 0.f;
    d.w = ((fb_hash32(k0 + 3u) & 0xffffu) >= thr) ? scale * a.w : 0.f;
    dhn[i] = d;
    dh[i] = make_float4(a.x - d.x, a.y - d.y, a.z - d.z, a.w - d.w);
}
extern "C" int fabind_drop_mix(const float* h, const float* hn, float* out, long n, float p_drop, unsigned seed, hipStream_t stream) {
    if (n <= 0) return 0;
    FB_REQUIRE(n % 4 == 0 && ((((uintptr_t)h | (uintptr_t)hn | (uintptr_t)out) & 15) == 0), "fabind_drop_mix: n % 4 == 0, 16-byte aligned buffers");
    FB_REQUIRE(p_drop > 0.f && p_drop < 1.f, "fabind_drop_mix: p_drop in (0, 1)");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const long n4 = n / 4;
    hipLaunchKernelGGL(drop_mix_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float4*)h, (const float4*)hn,
                       (float4*)out, n4, thr, 1.0f / (1.0f - (float)thr / 65536.0f), (uint32_t)seed);
    FB_CHECK_LAUNCH();
    return 0;
}
extern "C" int fabind_drop_mix_bwd(const float* g, float* dhn, float* dh, long n, float p_drop, unsigned seed, hipStream_t stream) {
    if (n <= 0) return 0;
    FB_REQUIRE(n % 4 == 0 && ((((uintptr_t)g | (uintptr_t)dhn | (uintptr_t)dh) & 15) == 0), "fabind_drop_mix_bwd: n % 4 == 0, 16-byte aligned buffers");
    FB_REQUIRE(p_drop > 0.f && p_drop < 1.f, "fabind_drop_mix_bwd: p_drop in (0, 1)");
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const long n4 = n / 4;
    hipLaunchKernelGGL(drop_mix_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float4*)g, (float4*)dhn,
                       (float4*)dh, n4, thr, 1.0f / (1.0f - (float)thr / 65536.0f), (uint32_t)seed);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adjoint of the row-dot epilogue:  part[m,t] = sum_{n in tile t} act(z[m,n]) u[n]
//   dz[m,n] = dpart[m, n/128] * u[n] * act'(z[m,n]);   du[n] = sum_m dpart[m, n/128] * act(z[m,n])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const void* z, int z_dt, const float* dpart, int np,
                                                         const float* u, int act, int M, int N, void* dz,
                                                         float* scratch, int rows_per) {
    // 256 columns per block (4 consecutive per lane), 4 row-lanes, fixed-order combine of the du partials
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < N) {
        const float4 uc = *(const float4*)(u + c);
        const int t = c / 128;
        for (int r = r0 + q; r < r1; r += 4) {
            const float4 zv = ld4_any(z, z_dt, (size_t)r * N + c);
            const float g = dpart[(size_t)r * np + t];
            st4_any(dz, z_dt, (size_t)r * N + c,
                    make_float4(g * uc.x * apply_dact(zv.x, act), g * uc.y * apply_dact(zv.y, act),
                                g * uc.z * apply_dact(zv.z, act), g * uc.w * apply_dact(zv.w, act)));
            s.x += g * apply_act(zv.x, act); s.y += g * apply_act(zv.y, act);
            s.z += g * apply_act(zv.z, act); s.w += g * apply_act(zv.w, act);
        }
    }
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && c + 3 < N) {
        const float4 a = part[0][lane], b = part[1][lane], d_ = part[2][lane], e = part[3][lane];
        float* o = scratch + (size_t)blockIdx.y * N + c;
        o[0] = (a.x + b.x) + (d_.x + e.x); o[1] = (a.y + b.y) + (d_.y + e.y);
        o[2] = (a.z + b.z) + (d_.z + e.z); o[3] = (a.w + b.w) + (d_.w + e.w);
    }
}
// out[c] = sum_k scratch[k, c]: 64 columns per 1024-thread block, the chunk rows over 16 row-lanes with 4
// independent loads in flight, fixed-order combine through LDS (deterministic).  Launch with SUMC_GRID / SUMC_BLOCK.
__global__ __launch_bounds__(1024) void sum_chunks_kernel(const float* __restrict__ scratch, float* out, int C, int nchunk) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int k = q;
        for (; k + 48 < nchunk; k += 64) {
            s0 += scratch[(size_t)k * C + c]; s1 += scratch[(size_t)(k + 16) * C + c];
            s2 += scratch[(size_t)(k + 32) * C + c]; s3 += scratch[(size_t)(k + 48) * C + c];
        }
        for (; k < nchunk; k += 16) s0 += scratch[(size_t)k * C + c];
    }
    part[q][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][lane];
        out[c] = t;
    }
}
extern "C" int fabind_rowdot_bwd(const void* z, int z_dt, const float* dpart, int np, const float* u, int act, int M,
                                 int N, void* dz, float* du, float* scratch, int nchunk, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    FB_REQUIRE(N % 4 == 0, "fabind_rowdot_bwd: N % 4");
    int rows_per = (M + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((N + 255) / 256, nchunk), dim3(256), 0, stream, z, z_dt, dpart, np, u, act,
                       M, N, dz, scratch, rows_per);
    hipLaunchKernelGGL(sum_chunks_kernel, SUMC_GRID(N), dim3(SUMC_BLOCK), 0, stream, scratch, du, N, nchunk);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adjoint of edge_geom:  d=x[r]-x[c], rho=|d|^2, rhohat=rho/sqrt(sum_b rho^2)
//   drho_e = drhohat_e/nrm - T_b rho_e / nrm^3,  T_b = sum_e drhohat_e rho_e;   g_e = dd_e + 2 drho_e d_e
//   dx[r] += g_e, dx[c] -= g_e   (float atomics on [N,3]: order-insensitive to rounding only)
// ------------------------------------------------------------------------------------------------
// pass 1: T_b per complex (one 1024-thread work-group per complex, contiguous edge range)
__global__ __launch_bounds__(1024) void edge_geom_bwd_t_kernel(const float* __restrict__ rho, const float* __restrict__ drhohat,
                                                               const int* rowptr, const int* node_off, float* T) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int e0 = rowptr[node_off[b]], e1 = rowptr[node_off[b + 1]];
    float acc = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += 1024) acc += drhohat[e] * rho[e];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k];
        T[b] = t;
    }
}
// pass 2: one thread per edge over the whole batch: g_e -> scratch (no atomics: a global node would take ~3000
// same-address atomics per complex)
__global__ __launch_bounds__(256) void edge_geom_bwd_g_kernel(const float* __restrict__ d, const float* __restrict__ rho,
                                                              const float* __restrict__ norm, const float* __restrict__ dd,
                                                              const float* __restrict__ drhohat, const int* __restrict__ row,
                                                              const int* __restrict__ node_off, const float* __restrict__ T,
                                                              int B, int E, float* __restrict__ g) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int r = row[e];
    int lo = 0, hi = B;                              // complex of node r: last b with node_off[b] <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (node_off[mid] <= r) lo = mid; else hi = mid;
    }
    const float inv = 1.f / norm[lo], inv3 = inv * inv * inv;
    const float drho = drhohat[e] * inv - T[lo] * rho[e] * inv3;
#pragma unroll
    for (int k = 0; k < 3; ++k) g[(size_t)e * 3 + k] = dd[(size_t)e * 3 + k] + 2.f * drho * d[(size_t)e * 3 + k];
}
// pass 3: one wave per node: dx[n] = sum_{e in row n} g_e - sum_{e : col(e) = n} g_e, the second sum walks the edges
// grouped by sending node (colptr, perm)
__global__ __launch_bounds__(256) void edge_geom_bwd_reduce_kernel(const float* __restrict__ g, const int* __restrict__ rowptr,
                                                                   const int* __restrict__ colptr,
                                                                   const int* __restrict__ perm, int n_nodes, float* dx) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_nodes) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int e = rowptr[n] + lane; e < rowptr[n + 1]; e += 64) {
        ax += g[(size_t)e * 3]; ay += g[(size_t)e * 3 + 1]; az += g[(size_t)e * 3 + 2];
    }
    for (int q = colptr[n] + lane; q < colptr[n + 1]; q += 64) {
        const int e = perm[q];
        ax -= g[(size_t)e * 3]; ay -= g[(size_t)e * 3 + 1]; az -= g[(size_t)e * 3 + 2];
    }
    ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
    if (lane == 0) { dx[(size_t)n * 3] = ax; dx[(size_t)n * 3 + 1] = ay; dx[(size_t)n * 3 + 2] = az; }
}
extern "C" int fabind_edge_geom_bwd(const float* d, const float* rho, const float* norm, const float* dd,
                                    const float* drhohat, const int* row, const int* rowptr, const int* colptr,
                                    const int* perm, const int* node_off, int B, int E, int n_nodes, float* dx,
                                    float* g_scratch, float* T_scratch, hipStream_t stream) {
    if (B <= 0 || n_nodes <= 0) return 0;
    if (E > 0) {
        hipLaunchKernelGGL(edge_geom_bwd_t_kernel, dim3(B), dim3(1024), 0, stream, rho, drhohat, rowptr, node_off, T_scratch);
        hipLaunchKernelGGL(edge_geom_bwd_g_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, d, rho, norm, dd, drhohat, row,
                           node_off, T_scratch, B, E, g_scratch);
    }
    hipLaunchKernelGGL(edge_geom_bwd_reduce_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, stream, g_scratch, rowptr, colptr,
                       perm, n_nodes, dx);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adjoint of gcl_pre w.r.t. rhohat and w_r:  drh[e] = dpre[e,:].w_r ;  dw[c] = sum_e rhohat[e] dpre[e,c]
// (dAB comes from two segment_sum launches: by row, and by column through a permutation)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcl_pre_bwd_kernel(const void* dpre, int dt, int H, const float* rhohat,
                                                          const float* w_r, int E, float* drh, float* scratch,
                                                          int edges_per) {
    // register accumulators: lane owns 4 consecutive columns per 256-column slab (H <= 1024)
    __shared__ float4 sh[3][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e0 = blockIdx.x * edges_per, e1 = min(E, e0 + edges_per);
    float4 acc[4], wr[4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        acc[sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = sidx * 256 + lane * 4;
        wr[sidx] = (c < H) ? *(const float4*)(w_r + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int e = e0 + w; e < e1; e += 4) {
        const float rh = rhohat[e];
        float dot = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int c = sidx * 256 + lane * 4;
            if (c < H) {
                const float4 g = ld4_any(dpre, dt, (size_t)e * H + c);
                dot += g.x * wr[sidx].x + g.y * wr[sidx].y + g.z * wr[sidx].z + g.w * wr[sidx].w;
                acc[sidx].x += rh * g.x; acc[sidx].y += rh * g.y; acc[sidx].z += rh * g.z; acc[sidx].w += rh * g.w;
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) drh[e] = dot;
    }
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        if (w > 0) sh[w - 1][sidx * 64 + lane] = acc[sidx];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int c = sidx * 256 + lane * 4;
            if (c < H) {
                float4 a = acc[sidx];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 o = sh[k][sidx * 64 + lane];
                    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                }
                *(float4*)(scratch + (size_t)blockIdx.x * H + c) = a;
            }
        }
    }
}
extern "C" int fabind_gcl_pre_bwd(const void* dpre, int dt, int H, const float* rhohat, const float* w_r, int E,
                                  float* drh, float* dw, float* scratch, int nchunk, hipStream_t stream) {
    if (E <= 0) return 0;
    int edges_per = (E + nchunk - 1) / nchunk;
    FB_REQUIRE(H % 4 == 0 && H <= 1024, "fabind_gcl_pre_bwd: H % 4, H <= 1024");
    hipLaunchKernelGGL(gcl_pre_bwd_kernel, dim3(nchunk), dim3(256), 0, stream, dpre, dt, H, rhohat,
                       w_r, E, drh, scratch, edges_per);
    hipLaunchKernelGGL(sum_chunks_kernel, SUMC_GRID(H), dim3(SUMC_BLOCK), 0, stream, scratch, dw, H, nchunk);
    FB_CHECK_LAUNCH();
    return 0;
}

// adjoint of segment_sum: dZ[e,:] = dout[row[e],:] * act'(Z[e,:])     (one wave per edge)
__global__ __launch_bounds__(256) void gather_dact_kernel(const float* dout, int ldo, const int* row, const void* Z,
                                                          int z_dt, int act, void* dZ, int dz_dt, int E, int H) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float* g = dout + (size_t)row[e] * ldo;
    for (int c = lane * 4; c < H; c += 256) {
        float4 gv = *(const float4*)(g + c), zv = ld4_any(Z, z_dt, (size_t)e * H + c);
        st4_any(dZ, dz_dt, (size_t)e * H + c,
                make_float4(gv.x * apply_dact(zv.x, act), gv.y * apply_dact(zv.y, act), gv.z * apply_dact(zv.z, act),
                            gv.w * apply_dact(zv.w, act)));
    }
}
extern "C" int fabind_gather_dact(const float* dout, int ldo, const int* row, const void* Z, int z_dt, int act, void* dZ,
                                  int dz_dt, int E, int H, hipStream_t stream) {
    if (E <= 0) return 0;
    hipLaunchKernelGGL(gather_dact_kernel, dim3((E + 3) / 4), dim3(256), 0, stream, dout, ldo, row, Z, z_dt, act, dZ,
                       dz_dt, E, H);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adjoint of coord_update (one wave per row): t = sum_e d_e s_e / den; x_out = x + clamp(t)
//   g = dxo * [|t| <= clampv];  dd_e = g s_e / den;  ds_e = (g . d_e) / den
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coord_update_bwd_kernel(const float* d, const float* s, const int* rowptr,
                                                               int n_rows, int mean, float clampv, const float* dxo,
                                                               float* dd, float* ds) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) {
        const float sv = s[e];
        ax += d[(size_t)e * 3] * sv; ay += d[(size_t)e * 3 + 1] * sv; az += d[(size_t)e * 3 + 2] * sv;
    }
    ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
    const int deg = e1 - e0;
    const float den = mean ? (float)(deg < 1 ? 1 : deg) : 1.f;
    const float gx = (fabsf(ax / den) <= clampv ? dxo[(size_t)r * 3] : 0.f) / den;
    const float gy = (fabsf(ay / den) <= clampv ? dxo[(size_t)r * 3 + 1] : 0.f) / den;
    const float gz = (fabsf(az / den) <= clampv ? dxo[(size_t)r * 3 + 2] : 0.f) / den;
    for (int e = e0 + lane; e < e1; e += 64) {
        const float sv = s[e];
        dd[(size_t)e * 3] = gx * sv; dd[(size_t)e * 3 + 1] = gy * sv; dd[(size_t)e * 3 + 2] = gz * sv;
        ds[e] = gx * d[(size_t)e * 3] + gy * d[(size_t)e * 3 + 1] + gz * d[(size_t)e * 3 + 2];
    }
}
extern "C" int fabind_coord_update_bwd(const float* d, const float* s, const int* rowptr, int n_rows, int mean,
                                       float clampv, const float* dxo, float* dd, float* ds, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(coord_update_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, d, s, rowptr, n_rows, mean,
                       clampv, dxo, dd, ds);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// cross attention backward.  Pass Q (thread = query x head): dq, dgpre, dbias(lin,gate), and the per-query
// helpers dO = dout*sigmoid(gpre) and D = dout.out for pass KV.  Pass KV (thread = key x head): dk, dv.
// Probabilities are recomputed from the saved log-sum-exp.
// ------------------------------------------------------------------------------------------------
#define CB_KT 32
// LDS tile rows: the four heads' 32-float segments sit 36 floats apart (row stride 144), so that the four distinct addresses a
// wave reads per instruction (one per head, each broadcast to 16 lanes) fall in different banks; at 32 apart heads 0/2 and 1/3
// share banks (64 banks x 4 B) and every inner-loop read is a 2-way conflict
#define CB_LD 144
#define CB_H(h) ((h) * 36)
#define CB_COL(c4) (((c4) >> 5) * 36 + ((c4) & 31))      /* column c4 of a [.,128] row -> padded offset */
__global__ __launch_bounds__(256) void cross_attn_bwd_q_kernel(const float* __restrict__ qg, int ldq,
                                                               const float* __restrict__ kv, int ldkv,
                                                               const float* __restrict__ bias, int bias_ld, int lin_col,
                                                               int gate_col, const int* desc, float scale,
                                                               const float* __restrict__ out, const float* lse,
                                                               const float* __restrict__ dout, float* dqg, float* dbias,
                                                               float* dO, float* Dv, int ksplit, float* part, int part_rows) {
    __shared__ __attribute__((aligned(16))) float sK[CB_KT * CB_LD];
    __shared__ __attribute__((aligned(16))) float sV[CB_KT * CB_LD];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nq) return;
    const int tid = threadIdx.x, ql = tid >> 2, h = tid & 3;
    const int qi = blockIdx.x * 64 + ql;
    const bool valid = qi < nq;
    float qr[32], dor[32], dq[32];
    float D = 0.f, L = 0.f;
#pragma unroll
    for (int dd_ = 0; dd_ < 32; ++dd_) { qr[dd_] = 0.f; dor[dd_] = 0.f; dq[dd_] = 0.f; }
    if (valid) {
        const size_t ro = (size_t)(q_off + qi);
        const float* qp = qg + ro * ldq + h * 32;
        const float* gp = qg + ro * ldq + 128 + h * 32;
        const float* op = out + ro * 128 + h * 32;
        const float* dp = dout + ro * 128 + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            qr[c] = qp[c] * scale;
            const float sg = sigmoid_f(gp[c]);
            dor[c] = dp[c] * sg;
            D += dp[c] * op[c];
            dqg[ro * ldq + 128 + h * 32 + c] = dp[c] * op[c] * (1.f - sg);   // d gpre
            dO[ro * 128 + h * 32 + c] = dor[c];
        }
        L = lse[ro * 4 + h];
        Dv[ro * 4 + h] = D;
    }
    int kb = 0, ke = nk;
    if (ksplit > 1) {
        const int per = ((nk + ksplit - 1) / ksplit + CB_KT - 1) / CB_KT * CB_KT;
        kb = blockIdx.z * per; ke = min(nk, kb + per);
    }
    for (int j0 = kb; j0 < ke; j0 += CB_KT) {
        __syncthreads();
        for (int i = tid; i < CB_KT * 32; i += 256) {
            int jr = i >> 5, c4 = (i & 31) * 4;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (j0 + jr < ke) {
                kk = *(const float4*)(kv + (size_t)(k_off + j0 + jr) * ldkv + c4);
                vv = *(const float4*)(kv + (size_t)(k_off + j0 + jr) * ldkv + 128 + c4);
            }
            *(float4*)(&sK[jr * CB_LD + CB_COL(c4)]) = kk;
            *(float4*)(&sV[jr * CB_LD + CB_COL(c4)]) = vv;
        }
        __syncthreads();
        if (!valid) continue;
        const int jn = min(CB_KT, ke - j0);
        // pair-bias loads (the only HBM traffic of the loop) are issued 8 keys ahead of their use
        for (int jg = 0; jg < jn; jg += 8) {
            float lin8[8], gate8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = min(jg + u, jn - 1);
                const size_t bi = (size_t)(pair_off + (long)qi * sq + (long)(j0 + jj) * sk) * bias_ld;
                lin8[u] = bias[bi + lin_col + h];
                gate8[u] = bias[bi + gate_col + h];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jg + u;
                if (j < jn) {
                    const float* kp = &sK[j * CB_LD + CB_H(h)];
                    const float* vp = &sV[j * CB_LD + CB_H(h)];
                    float s = 0.f, dp = 0.f;
#pragma unroll
                    for (int c = 0; c < 32; ++c) { s += qr[c] * kp[c]; dp += dor[c] * vp[c]; }
                    const size_t bi = (size_t)(pair_off + (long)qi * sq + (long)(j0 + j) * sk) * bias_ld;
                    const float lin = lin8[u], sg = sigmoid_f(gate8[u]);
                    s += lin * sg;
                    const float p = __expf(s - L);
                    const float dsj = p * (dp - D);
                    dbias[bi + lin_col + h] = dsj * sg;
                    dbias[bi + gate_col + h] = dsj * lin * sg * (1.f - sg);
#pragma unroll
                    for (int c = 0; c < 32; ++c) dq[c] += dsj * kp[c];
                }
            }
        }
    }
    if (!valid) return;
    float* dqp = dqg + (size_t)(q_off + qi) * ldq + h * 32;
    if (ksplit > 1) {      // partial over this key range -> part[split][complex][local query][128]; split_reduce_kernel sums them
        float* pp = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * part_rows + qi) * 128 + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) pp[c] = dq[c] * scale;
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) dqp[c] = dq[c] * scale;
    }
}

__global__ __launch_bounds__(256) void cross_attn_bwd_kv_kernel(const float* __restrict__ qg, int ldq,
                                                                const float* __restrict__ kv, int ldkv,
                                                                const float* __restrict__ bias, int bias_ld, int lin_col,
                                                                int gate_col, const int* desc, float scale,
                                                                const float* lse, const float* __restrict__ dO,
                                                                const float* Dv, float* dkv, int qsplit, float* part, int part_rows) {
    __shared__ __attribute__((aligned(16))) float sQ[CB_KT * CB_LD];
    __shared__ __attribute__((aligned(16))) float sDO[CB_KT * CB_LD];
    __shared__ float sL[CB_KT * 4], sD[CB_KT * 4];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nk) return;
    const int tid = threadIdx.x, kl = tid >> 2, h = tid & 3;
    const int kj = blockIdx.x * 64 + kl;
    const bool valid = kj < nk;
    float kr[32], vr[32], dk[32], dv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { kr[c] = 0.f; vr[c] = 0.f; dk[c] = 0.f; dv[c] = 0.f; }
    if (valid) {
        const float* kp = kv + (size_t)(k_off + kj) * ldkv + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) { kr[c] = kp[c]; vr[c] = kp[128 + c]; }
    }
    int qb = 0, qe = nq;
    if (qsplit > 1) {
        const int per = ((nq + qsplit - 1) / qsplit + CB_KT - 1) / CB_KT * CB_KT;
        qb = blockIdx.z * per; qe = min(nq, qb + per);
    }
    for (int i0 = qb; i0 < qe; i0 += CB_KT) {
        __syncthreads();
        for (int i = tid; i < CB_KT * 32; i += 256) {
            int ir = i >> 5, c4 = (i & 31) * 4;
            float4 qq = make_float4(0.f, 0.f, 0.f, 0.f), dd_ = qq;
            if (i0 + ir < qe) {
                qq = *(const float4*)(qg + (size_t)(q_off + i0 + ir) * ldq + c4);
                dd_ = *(const float4*)(dO + (size_t)(q_off + i0 + ir) * 128 + c4);
            }
            *(float4*)(&sQ[ir * CB_LD + CB_COL(c4)]) = qq;
            *(float4*)(&sDO[ir * CB_LD + CB_COL(c4)]) = dd_;
        }
        if (tid < CB_KT * 4) {
            int ir = tid >> 2;
            sL[tid] = (i0 + ir < qe) ? lse[(size_t)(q_off + i0 + ir) * 4 + (tid & 3)] : 0.f;
            sD[tid] = (i0 + ir < qe) ? Dv[(size_t)(q_off + i0 + ir) * 4 + (tid & 3)] : 0.f;
        }
        __syncthreads();
        if (!valid) continue;
        const int in_ = min(CB_KT, qe - i0);
        for (int ig = 0; ig < in_; ig += 8) {
            float lin8[8], gate8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ii = min(ig + u, in_ - 1);
                const size_t bi = (size_t)(pair_off + (long)(i0 + ii) * sq + (long)kj * sk) * bias_ld;
                lin8[u] = bias[bi + lin_col + h];
                gate8[u] = bias[bi + gate_col + h];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = ig + u;
                if (i < in_) {
                    const float* qp = &sQ[i * CB_LD + CB_H(h)];
                    const float* dp_ = &sDO[i * CB_LD + CB_H(h)];
                    float s = 0.f, dp = 0.f;
#pragma unroll
                    for (int c = 0; c < 32; ++c) { s += qp[c] * kr[c]; dp += dp_[c] * vr[c]; }
                    s = s * scale + lin8[u] * sigmoid_f(gate8[u]);
                    const float p = __expf(s - sL[i * 4 + h]);
                    const float dsj = p * (dp - sD[i * 4 + h]) * scale;
#pragma unroll
                    for (int c = 0; c < 32; ++c) { dk[c] += dsj * qp[c]; dv[c] += p * dp_[c]; }
                }
            }
        }
    }
    if (!valid) return;
    float* o = dkv + (size_t)(k_off + kj) * ldkv + h * 32;
    if (qsplit > 1) {      // partial over this query range -> part[split][complex][local key][256] (dk | dv)
        float* pp = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * part_rows + kj) * 256 + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) { pp[c] = dk[c]; pp[128 + c] = dv[c]; }
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) { o[c] = dk[c]; o[128 + c] = dv[c]; }
    }
}

// Sum of the per-split partials of the split direction (<= 256 rows per complex): out[(off_b + r) * ldo + c] = sum_z part[z][b][r][c].
// Replaces float atomics into dq / dkv (11-way contended, one cache line per lane: ~0.4 ms per launch at the bench shape).
__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ part, int nsplit, int part_rows, int cols,
                                                           const int* __restrict__ desc, int off_idx, int cnt_idx,
                                                           float* __restrict__ out, int ldo) {
    const int* ds = desc + blockIdx.y * 8;
    const int off = ds[off_idx], n = ds[cnt_idx];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int r = idx / cols, c = idx % cols;
    if (r >= n) return;
    float acc = 0.f;
    for (int z = 0; z < nsplit; ++z) acc += part[(((size_t)z * gridDim.y + blockIdx.y) * part_rows + r) * cols + c];
    out[(size_t)(off + r) * ldo + c] = acc;
}

// floats of scratch fabind_cross_attn_bwd needs for (B, max_nq, max_nk); 0 when neither pass is split
static void ca_splits(int max_nq, int max_nk, int* ksplit, int* qsplit) {
    // few queries / many keys -> split the keys in pass Q; few keys / many queries -> split the queries in pass KV
    *ksplit = (max_nq <= 256 && max_nk >= 512) ? max_nk / 128 : 1;
    *qsplit = (max_nk <= 256 && max_nq >= 512) ? max_nq / 128 : 1;
    if (*ksplit > 32) *ksplit = 32;
    if (*qsplit > 32) *qsplit = 32;
}
extern "C" long fabind_cross_attn_bwd_scratch(int B, int max_nq, int max_nk) {
    int ks, qs;
    ca_splits(max_nq, max_nk, &ks, &qs);
    if (ks > 1) return (long)ks * B * max_nq * 128;
    if (qs > 1) return (long)qs * B * max_nk * 256;
    return 0;
}

extern "C" int fabind_cross_attn_bwd(const float* qg, int ldq, const float* kv, int ldkv, const float* bias, int bias_ld,
                                     int lin_col, int gate_col, const int* desc, int B, int max_nq, int max_nk,
                                     float scale, const float* out, const float* lse, const float* dout, float* dqg,
                                     float* dkv, float* dbias, float* dO, float* Dv, float* scratch, hipStream_t stream) {
    if (B <= 0) return 0;
    int ksplit, qsplit;
    ca_splits(max_nq, max_nk, &ksplit, &qsplit);
    FB_REQUIRE((ksplit == 1 && qsplit == 1) || scratch != nullptr,
               "fabind_cross_attn_bwd: this shape splits a pass and needs fabind_cross_attn_bwd_scratch() floats of scratch");
    hipLaunchKernelGGL(cross_attn_bwd_q_kernel, dim3((max_nq + 63) / 64, B, ksplit), dim3(256), 0, stream, qg, ldq, kv,
                       ldkv, bias, bias_ld, lin_col, gate_col, desc, scale, out, lse, dout, dqg, dbias, dO, Dv, ksplit, scratch,
                       max_nq);
    if (ksplit > 1)
        hipLaunchKernelGGL(split_reduce_kernel, dim3((max_nq * 128 + 255) / 256, B), dim3(256), 0, stream, scratch, ksplit, max_nq,
                           128, desc, 0, 1, dqg, ldq);
    hipLaunchKernelGGL(cross_attn_bwd_kv_kernel, dim3((max_nk + 63) / 64, B, qsplit), dim3(256), 0, stream, qg, ldq, kv,
                       ldkv, bias, bias_ld, lin_col, gate_col, desc, scale, lse, dO, Dv, dkv, qsplit, scratch, max_nk);
    if (qsplit > 1)
        hipLaunchKernelGGL(split_reduce_kernel, dim3((max_nk * 256 + 255) / 256, B), dim3(256), 0, stream, scratch, qsplit, max_nk,
                           256, desc, 2, 3, dkv, ldkv);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Operands of the pair-bias adjoint with all attention blocks concatenated along K (bf16 training path).
//   Acat [sum_b P_b, nblk*Kp]:  Acat[poff_b + i, k*Kp + j*8 + o] = dout_k[pair(b,i,j), o]      (0 for j >= C_b; Kp = 8*max_C -> %32)
//   BTcat[B*H,       nblk*Kp]:  BTcat[b*H + h,   k*Kp + j*8 + o] = b0[lig_j, h] * wcomp[k, o, h]
// so that  d a0[protein rows of b] += Acat_b . BTcat_b^T  is ONE plain-group GEMM (one accumulating epilogue pass over the
// strided fp32 gradient instead of one per block), and the per-block  T_k = D_k^T a0  contractions read the column slice k of
// Acat (ldy = nblk*Kp).  A row of dout_k for a (protein i, all ligand j, o) is contiguous, so both kernels move 16-B chunks.
// ------------------------------------------------------------------------------------------------
struct PbCatPtrs { const float* d[16]; };
// Thread c owns column chunk c = (block k, ligand atom j) for a strip of PB_ROWS protein rows: 32-B loads and 16-B stores are
// contiguous across the threads of a row, and the eight column sums of the fp32 values it reads (the gradient of the pair-bias
// constants is the column sum of every dout_k) accumulate in registers -- one plain LDS tree per work-group, no atomics.
// colpart (optional): [gridDim.y * gridDim.x][nblk * 8] per-work-group partial sums, added up by the caller in a fixed order.
#define PB_ROWS 32
__global__ __launch_bounds__(384) void pair_bias_dcat_kernel(PbCatPtrs ptrs, int nblk, const int* __restrict__ desc_p, int max_P,
                                                             int Kp, bf16_t* __restrict__ Acat, int lda, float* __restrict__ colpart) {
    extern __shared__ float sS[];                                   // [nblk * cpr][8]
    const int* ds = desc_p + blockIdx.y * 8;
    const int poff = ds[0], P = ds[1], C = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int cpr = Kp >> 3, ncol = nblk * cpr;
    const int i0 = blockIdx.x * PB_ROWS, i1 = min(P, i0 + PB_ROWS);
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
        const int k = c / cpr, j = c % cpr;
        const float* d = ptrs.d[k];
        const bool live = j < C && d != nullptr;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = i0; i < i1; ++i) {
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            if (live) {
                const float* src = d + ((size_t)(pair_off + (long)i * C + j)) * 8;
                const float4 x = *(const float4*)src, y = *(const float4*)(src + 4);
                u.x = pack2_bf16(x.x, x.y); u.y = pack2_bf16(x.z, x.w); u.z = pack2_bf16(y.x, y.y); u.w = pack2_bf16(y.z, y.w);
                acc[0] += x.x; acc[1] += x.y; acc[2] += x.z; acc[3] += x.w;
                acc[4] += y.x; acc[5] += y.y; acc[6] += y.z; acc[7] += y.w;
            }
            *(uint4*)(Acat + (size_t)(poff + i) * lda + (size_t)c * 8) = u;
        }
        if (colpart != nullptr) {
#pragma unroll
            for (int o = 0; o < 8; ++o) sS[c * 8 + o] = acc[o];
        }
    }
    if (colpart != nullptr) {
        __syncthreads();
        for (int t = threadIdx.x; t < nblk * 8; t += blockDim.x) {
            const int k = t >> 3, o = t & 7;
            float tot = 0.f;
            for (int j = 0; j < cpr; ++j) tot += sS[(k * cpr + j) * 8 + o];
            colpart[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (nblk * 8) + t] = tot;
        }
    }
}
__global__ __launch_bounds__(256) void pair_bias_btcat_kernel(const float* __restrict__ b0, int ld_ab, const int* __restrict__ c_index,
                                                              const int* __restrict__ desc_p, const float* __restrict__ wcomp,
                                                              int nblk, int H, int Kp, bf16_t* __restrict__ BTcat, int ldb) {
    const int* ds = desc_p + blockIdx.y * 8;
    const int coff = ds[2], C = ds[3];
    const int cpr = Kp >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_h = (long)nblk * cpr;
    const int h = (int)(idx / per_h);
    if (h >= H) return;
    const int rem = (int)(idx % per_h), k = rem / cpr, j = rem % cpr;
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (j < C) {
        const float bv = b0[(size_t)c_index[coff + j] * ld_ab + h];
        const float* w = wcomp + ((size_t)k * 8) * H + h;            // wcomp[k][o][h]
        float v[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) v[o] = bv * w[(size_t)o * H];
        u.x = pack2_bf16(v[0], v[1]); u.y = pack2_bf16(v[2], v[3]); u.z = pack2_bf16(v[4], v[5]); u.w = pack2_bf16(v[6], v[7]);
    }
    *(uint4*)(BTcat + ((size_t)blockIdx.y * H + h) * ldb + k * Kp + j * 8) = u;
}
// Batched tiled transpose, bf16, with zero padding of the new inner dimension: for z = (b, k) (k < nsub)
//   out[(k*B + b)*cols + m][i] = in[row0_b + i][k*cols + m]   for i < P_b,   0 for P_b <= i < Pp
// (row0_b, P_b) = desc[b][0..1].  Feeds the per-block T = D^T a0 contractions of the pair-bias adjoint to the pipelined NT GEMM:
// K becomes the (padded) protein length, uniform over the complexes, so plain groups apply.  64 x 32 tiles through LDS, 16-byte
// loads along the input rows and 16-byte stores along the output rows.
__global__ __launch_bounds__(256) void batched_transpose_pad_kernel(const bf16_t* __restrict__ in, int ld_in, const int* __restrict__ desc,
                                                                    int B, int nsub, int cols, int Pp, bf16_t* __restrict__ out) {
    __shared__ bf16_t tile[64][40];                                  // 32 columns + padding (rows stay 16-byte aligned: 80 B)
    const int b = blockIdx.z / nsub, k = blockIdx.z % nsub;
    const int* ds = desc + b * 8;
    const int row0 = ds[0], P = ds[1];
    const int i0 = blockIdx.x * 64, m0 = blockIdx.y * 32;
    const int t = threadIdx.x;
    {
        const int r = t >> 2, c8 = (t & 3) * 8;
        uint4 u = make_uint4(0u, 0u, 0u, 0u);
        if (i0 + r < P) u = *(const uint4*)(in + (size_t)(row0 + i0 + r) * ld_in + k * cols + m0 + c8);
        *(uint4*)&tile[r][c8] = u;
    }
    __syncthreads();
    {
        const int m = t >> 3, i8 = (t & 7) * 8;
        bf16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = tile[i8 + q][m];
        if (i0 + i8 < Pp)
            *(uint4*)(out + ((size_t)(k * B + b) * cols + m0 + m) * Pp + i0 + i8) = *(const uint4*)v;
    }
}
extern "C" int fabind_batched_transpose_pad(const void* in, int ld_in, const int* desc, int B, int nsub, int cols, int Pp, void* out,
                                            hipStream_t stream) {
    if (B <= 0 || nsub <= 0) return 0;
    FB_REQUIRE(cols % 32 == 0 && Pp % 8 == 0 && ld_in % 8 == 0, "fabind_batched_transpose_pad: cols % 32, Pp % 8, ld_in % 8");
    hipLaunchKernelGGL(batched_transpose_pad_kernel, dim3((Pp + 63) / 64, cols / 32, B * nsub), dim3(256), 0, stream, (const bf16_t*)in,
                       ld_in, desc, B, nsub, cols, Pp, (bf16_t*)out);
    FB_CHECK_LAUNCH();
    return 0;
}

// rows of the colpart array fabind_pair_bias_cat writes (one per work-group of the operand-building kernel)
extern "C" long fabind_pair_bias_cat_parts(int B, int max_P) { return (long)B * ((max_P + PB_ROWS - 1) / PB_ROWS); }
extern "C" int fabind_pair_bias_cat(const void* const* douts, int nblk, const int* desc_p, int B, int max_P, int Kp, void* Acat,
                                    int lda, const float* b0, int ld_ab, const int* c_index, const float* wcomp, int H,
                                    void* BTcat, int ldb, float* colpart, hipStream_t stream) {
    if (B <= 0 || nblk <= 0) return 0;
    FB_REQUIRE(nblk <= 16 && Kp % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= nblk * Kp && ldb >= nblk * Kp,
               "fabind_pair_bias_cat: nblk <= 16, Kp % 32 == 0, lda / ldb % 8 == 0 and >= nblk * Kp");
    PbCatPtrs ptrs;
    for (int k = 0; k < 16; ++k) ptrs.d[k] = k < nblk ? (const float*)douts[k] : nullptr;
    const long cpr = Kp / 8;
    FB_REQUIRE(nblk * cpr * 8 * sizeof(float) <= 64 * 1024, "fabind_pair_bias_cat: nblk * Kp too large for the column-sum tile");
    hipLaunchKernelGGL(pair_bias_dcat_kernel, dim3((max_P + PB_ROWS - 1) / PB_ROWS, B), dim3(384), (size_t)nblk * cpr * 8 * sizeof(float),
                       stream, ptrs, nblk, desc_p, max_P, Kp, (bf16_t*)Acat, lda, colpart);
    hipLaunchKernelGGL(pair_bias_btcat_kernel, dim3((unsigned)(((long)H * nblk * cpr + 255) / 256), B), dim3(256), 0, stream, b0, ld_ab,
                       c_index, desc_p, wcomp, nblk, H, Kp, (bf16_t*)BTcat, ldb);
    FB_CHECK_LAUNCH();
    return 0;
}

// The ligand-side operand alone (the fused attention backward writes the gradient rows `Acat` itself): BTcat of fabind_pair_bias_cat
extern "C" int fabind_pair_bias_btcat(const float* b0, int ld_ab, const int* c_index, const int* desc_p, const float* wcomp, int nblk, int H,
                                      int Kp, void* BTcat, int ldb, int B, hipStream_t stream) {
    if (B <= 0 || nblk <= 0) return 0;
    FB_REQUIRE(nblk <= 16 && Kp % 32 == 0 && ldb % 8 == 0 && ldb >= nblk * Kp, "fabind_pair_bias_btcat: nblk <= 16, Kp % 32 == 0, ldb % 8 == 0 and >= nblk * Kp");
    const long cpr = Kp / 8;
    hipLaunchKernelGGL(pair_bias_btcat_kernel, dim3((unsigned)(((long)H * nblk * cpr + 255) / 256), B), dim3(256), 0, stream, b0, ld_ab,
                       c_index, desc_p, wcomp, nblk, H, Kp, (bf16_t*)BTcat, ldb);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adjoint of pair_hadamard: hd[e, 0:H] = a0[p]*b0[c], hd[e, H:H+H2] = a1[p]*b1[c]
// (float atomics: every node collects the few pairs it belongs to)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_hadamard_bwd_kernel(const void* dhd, int dt, int ldh, const float* a0,
                                                                const float* b0, int ld0, int H, const float* a1,
                                                                const float* b1, int ld1, int H2, const int* red_p,
                                                                const int* red_c, int n, float* da0, float* db0,
                                                                int ldd0, float* da1, float* db1, int ldd1) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= n) return;
    const int pn = red_p[e], cn = red_c[e];
    for (int c = lane; c < H; c += 64) {
        const float g = ld_any(dhd, dt, (size_t)e * ldh + c);
        atomicAdd(&da0[(size_t)pn * ldd0 + c], g * b0[(size_t)cn * ld0 + c]);
        atomicAdd(&db0[(size_t)cn * ldd0 + c], g * a0[(size_t)pn * ld0 + c]);
    }
    for (int c = lane; c < H2; c += 64) {
        const float g = ld_any(dhd, dt, (size_t)e * ldh + H + c);
        atomicAdd(&da1[(size_t)pn * ldd1 + c], g * b1[(size_t)cn * ld1 + c]);
        atomicAdd(&db1[(size_t)cn * ldd1 + c], g * a1[(size_t)pn * ld1 + c]);
    }
}
// The same adjoint over the inter graph's reduced pairs WITHOUT float atomics: the inter graph is symmetric, so the pairs of a node are
// the inter edges of its own row (pair index = red_idx[e], partner = col[e]).  One wave per node walks that row; every output element
// has one writer and a fixed summation order (the atomic form adds in whatever order the work-groups arrive).  T = (a | b) per block:
// a ligand node receives d b = sum g * a[partner], a protein node d a = sum g * b[partner]; the sums are ADDED to d0 / d1 (d0 may be a
// shared gradient buffer that already holds other consumers' contributions).
__global__ __launch_bounds__(256) void pair_hadamard_bwd_rows_kernel(const void* dhd, int dt, int ldh, const float* __restrict__ t0,
                                                                    int ld0, int H, const float* __restrict__ t1, int ld1, int H2,
                                                                    const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                    const int* __restrict__ red_idx, const int* __restrict__ red_c,
                                                                    int n_rows, float* d0, int ldd0, float* d1, int ldd1) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rows) return;
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    if (e0 == e1) return;
    const bool lig = red_c[red_idx[e0]] == n;                 // which side of its pairs this node is
    // Edge loop outside, column chunks inside; the two index loads of an edge (its pair's row of dhd, its partner node) are made for
    // up to 64 edges at once, one edge per lane, and handed out with v_readlane -- inside the loop each was the head of a dependent
    // load chain per edge and column chunk (see inter_attn_fwd_kernel).  H <= 1024, H2 <= 256 (checked by the entry point).
    const int src0 = lig ? 0 : H, dst0 = lig ? H : 0;         // partner's factor / this node's gradient slot inside (a | b)
    const int src1 = lig ? 0 : H2, dst1 = lig ? H2 : 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc0[4] = {z4, z4, z4, z4}, acc1 = z4;
    for (int eb = e0; eb < e1; eb += 64) {
        const int ne = min(64, e1 - eb);
        int ri_l = 0, cn_l = 0;
        if (lane < ne) { ri_l = red_idx[eb + lane]; cn_l = col[eb + lane]; }
#pragma unroll 2
        for (int j = 0; j < ne; ++j) {
            const size_t ri = (size_t)__shfl(ri_l, j, 64), cn = (size_t)__shfl(cn_l, j, 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane * 4 + 256 * k;
                if (c < H) {
                    const float4 g = ld4_any(dhd, dt, ri * ldh + c);
                    const float4 v = *(const float4*)(t0 + cn * ld0 + src0 + c);
                    acc0[k].x += g.x * v.x; acc0[k].y += g.y * v.y; acc0[k].z += g.z * v.z; acc0[k].w += g.w * v.w;
                }
            }
            if (lane * 4 < H2) {
                const int c = lane * 4;
                const float4 g = ld4_any(dhd, dt, ri * ldh + H + c);
                const float4 v = *(const float4*)(t1 + cn * ld1 + src1 + c);
                acc1.x += g.x * v.x; acc1.y += g.y * v.y; acc1.z += g.z * v.z; acc1.w += g.w * v.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < H) {
            float4* o = (float4*)(d0 + (size_t)n * ldd0 + dst0 + c);
            float4 q = *o;
            q.x += acc0[k].x; q.y += acc0[k].y; q.z += acc0[k].z; q.w += acc0[k].w;
            *o = q;
        }
    }
    if (lane * 4 < H2) {
        float4* o = (float4*)(d1 + (size_t)n * ldd1 + dst1 + lane * 4);
        float4 q = *o;
        q.x += acc1.x; q.y += acc1.y; q.z += acc1.z; q.w += acc1.w;
        *o = q;
    }
}
extern "C" int fabind_pair_hadamard_bwd_rows(const void* dhd, int dt, int ldh, const float* t0, int ld0, int H, const float* t1,
                                             int ld1, int H2, const int* rowptr, const int* col, const int* red_idx,
                                             const int* red_c, int n_rows, float* d0, int ldd0, float* d1, int ldd1,
                                             hipStream_t stream) {
    if (n_rows <= 0) return 0;
    FB_REQUIRE(H % 4 == 0 && H2 % 4 == 0 && ldh % 4 == 0 && ld0 % 4 == 0 && ldd0 % 4 == 0 && (H2 == 0 || (ld1 % 4 == 0 && ldd1 % 4 == 0)),
               "fabind_pair_hadamard_bwd_rows: widths and leading dimensions must be multiples of 4");
    FB_REQUIRE((((uintptr_t)t0 | (uintptr_t)d0 | (uintptr_t)dhd) & 15) == 0 && (H2 == 0 || (((uintptr_t)t1 | (uintptr_t)d1) & 15) == 0),
               "fabind_pair_hadamard_bwd_rows: 16-byte aligned buffers");
    FB_REQUIRE(H <= 1024 && H2 <= 256, "fabind_pair_hadamard_bwd_rows: H <= 1024, H2 <= 256");
    hipLaunchKernelGGL(pair_hadamard_bwd_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, dhd, dt, ldh, t0, ld0, H, t1, ld1, H2,
                       rowptr, col, red_idx, red_c, n_rows, d0, ldd0, d1, ldd1);
    FB_CHECK_LAUNCH();
    return 0;
}

// Adjoint of out[e, :] = t[ia[e], :] * t[ib[e], :] as a row walk (see the header): the float-atomics kernel below took 2.2 ms per call
// in the full model's distance-map head (every ligand atom's ~150 pairs add 512-float rows into the same row of d t).
__global__ __launch_bounds__(256) void rows_hadamard_bwd_kernel(const void* __restrict__ dout, int dt, int ldo, const float* __restrict__ t,
                                                               int ldt, int W, const int* __restrict__ rowptr,
                                                               const int* __restrict__ pair_idx, const int* __restrict__ partner,
                                                               int n_rows, float* dT, int lddt) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rows) return;
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[4] = {z4, z4, z4, z4};
    for (int eb = e0; eb < e1; eb += 64) {               // the two indices of up to 64 pairs, one pair per lane, ahead of the walk
        const int ne = min(64, e1 - eb);
        int pi_l = 0, pn_l = 0;
        if (lane < ne) { pi_l = pair_idx[eb + lane]; pn_l = partner[eb + lane]; }
#pragma unroll 2
        for (int j = 0; j < ne; ++j) {
            const size_t pi = (size_t)__shfl(pi_l, j, 64), pn = (size_t)__shfl(pn_l, j, 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane * 4 + 256 * k;
                if (c < W) {
                    const float4 g = ld4_any(dout, dt, pi * ldo + c);
                    const float4 v = *(const float4*)(t + pn * ldt + c);
                    acc[k].x += g.x * v.x; acc[k].y += g.y * v.y; acc[k].z += g.z * v.z; acc[k].w += g.w * v.w;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        if (c < W) *(float4*)(dT + (size_t)n * lddt + c) = acc[k];
    }
}
extern "C" int fabind_rows_hadamard_bwd(const void* dout, int dt, int ldo, const float* t, int ldt, int W, const int* rowptr,
                                        const int* pair_idx, const int* partner, int n_rows, float* dT, int lddt, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    FB_REQUIRE(W % 4 == 0 && W >= 4 && W <= 1024 && ldo % 4 == 0 && ldt % 4 == 0 && lddt % 4 == 0,
               "fabind_rows_hadamard_bwd: W a multiple of 4, <= 1024; strides % 4");
    FB_REQUIRE((((uintptr_t)t | (uintptr_t)dT | (uintptr_t)dout) & 15) == 0, "fabind_rows_hadamard_bwd: 16-byte aligned buffers");
    hipLaunchKernelGGL(rows_hadamard_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, dout, dt, ldo, t, ldt, W, rowptr, pair_idx,
                       partner, n_rows, dT, lddt);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_pair_hadamard_bwd(const void* dhd, int dt, int ldh, const float* a0, const float* b0, int ld0,
                                        int H, const float* a1, const float* b1, int ld1, int H2, const int* red_p,
                                        const int* red_c, int n, float* da0, float* db0, int ldd0, float* da1,
                                        float* db1, int ldd1, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pair_hadamard_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, dhd, dt, ldh, a0, b0, ld0, H, a1,
                       b1, ld1, H2, red_p, red_c, n, da0, db0, ldd0, da1, db1, ldd1);
    FB_CHECK_LAUNCH();
    return 0;
}

// The same adjoint for the ALL-PAIRS list of a batch (FABind+ threads the pair embedding over every protein x ligand pair: pair
// (b, i, j) sits at pair_off[b] + i * C_b + j, fabind_amd/plus/engine.py: PairList) without float atomics: one wave per node; a protein
// node i sums its C_b CONTIGUOUS rows against the ligand factors b[j], a ligand node j its P_b rows (stride C_b) against the protein
// factors a[i].  Every output element has one writer and a fixed summation order; the atomics form took 1.4 ms per launch on average
// (4.3 ms at W = 512) in the FABind+ training step because a ligand node's ~250 pairs all hit the same W addresses.
// T = (a | b) rows of width ld; d T gets d a in [0, W) of protein rows and d b in [W, 2W) of ligand rows (the other halves untouched).
__global__ __launch_bounds__(256) void pair_hadamard_bwd_grid_kernel(const void* __restrict__ dhd, int dt, int ldh,
                                                                    const float* __restrict__ T, int ldt, int W,
                                                                    const int* __restrict__ node_off, const int* __restrict__ c_cnt,
                                                                    const int* __restrict__ node_b, const int* __restrict__ desc_p,
                                                                    int n_nodes, float* dT, int lddt) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_nodes) return;
    const int b = node_b[n];
    const int off = node_off[b], C = c_cnt[b], P = node_off[b + 1] - off - C;
    const long po = (long)(unsigned)desc_p[b * 8 + 4] | ((long)desc_p[b * 8 + 5] << 32);
    const int loc = n - off;
    const bool lig = loc < C;
    // rows of this node: ligand j = loc: pairs po + i * C + j (i < P), partner protein node off + C + i, its factor a = T[.][0, W)
    //                    protein i = loc - C: pairs po + i * C + j (j < C), partner ligand node off + j, its factor b = T[.][W, 2W)
    const int cnt = lig ? P : C;
    const long first = lig ? po + loc : po + (long)(loc - C) * C;
    const int pstep = lig ? C : 1;
    const int partner0 = lig ? off + C : off;
    const int src = lig ? 0 : W, dst = lig ? W : 0;
    const int LPR = W / 4;                                    // lanes per row (float4 each)
    if (LPR >= 64) {
        for (int c = lane * 4; c < W; c += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int k = 0; k < cnt; ++k) {
                const float4 g = ld4_any(dhd, dt, (size_t)(first + (long)k * pstep) * ldh + c);
                const float4 v = *(const float4*)(T + (size_t)(partner0 + k) * ldt + src + c);
                acc.x += g.x * v.x; acc.y += g.y * v.y; acc.z += g.z * v.z; acc.w += g.w * v.w;
            }
            *(float4*)(dT + (size_t)n * lddt + dst + c) = acc;
        }
    } else {                                                  // narrow rows: 64 / LPR rows of the node per wave iteration, combined at the end
        const int G = 64 / LPR, grp = lane / LPR, c = (lane % LPR) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int k = grp; k < cnt; k += G) {             // (a ligand node of the pocket model walks 1,500 pairs: eight row pairs in flight)
            const float4 g = ld4_any(dhd, dt, (size_t)(first + (long)k * pstep) * ldh + c);
            const float4 v = *(const float4*)(T + (size_t)(partner0 + k) * ldt + src + c);
            acc.x += g.x * v.x; acc.y += g.y * v.y; acc.z += g.z * v.z; acc.w += g.w * v.w;
        }
        for (int o = LPR; o < 64; o <<= 1) {                  // fixed combination order of the G partial sums
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (grp == 0) *(float4*)(dT + (size_t)n * lddt + dst + c) = acc;
    }
}
extern "C" int fabind_pair_hadamard_bwd_grid(const void* dhd, int dt, int ldh, const float* T, int ldt, int W, const int* node_off,
                                             const int* c_cnt, const int* node_b, const int* desc_p, int n_nodes, float* dT, int lddt,
                                             hipStream_t stream) {
    if (n_nodes <= 0) return 0;
    FB_REQUIRE(W % 4 == 0 && W >= 4 && (W >= 256 || 256 % W == 0) && ldt % 4 == 0 && lddt % 4 == 0 && ldh % 4 == 0,
               "fabind_pair_hadamard_bwd_grid: W a multiple of 4 that divides 256 (or >= 256), strides % 4");
    hipLaunchKernelGGL(pair_hadamard_bwd_grid_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, stream, dhd, dt, ldh, T, ldt, W, node_off,
                       c_cnt, node_b, desc_p, n_nodes, dT, lddt);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// inter-graph attention backward.
// Pass A (waves stride over rows): softmax adjoint per row -> dlogit[e], dcp[e], dd[e], drh[e], dq[r],
//   per-block partials of dw_rk, dw_rv, dwcr, dw3.
// Pass B (one wave per node n): the inter graph is symmetric, so "scatter by sending node" is a gather
//   over the node's own row through mirror[]: dk[n], dv[n], dcv[n]; dbias_red[pair] = dlogit[e]+dlogit[mirror].
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void inter_attn_bwd_a_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* __restrict__ d,
    const float* __restrict__ rhohat, const int* rowptr, const int* col, const float* __restrict__ w_rk,
    const float* __restrict__ w_rv, const float* __restrict__ wcr, const float* __restrict__ w3, const float* alpha,
    const float* cvs, float clampv, int n_rows, const float* __restrict__ dh_out, const float* __restrict__ dx_out,
    float* dqkv, float* dd, float* drh, float* dlogit, float* dcp, float* wpart) {
    extern __shared__ float4 shp[];   // [4 quantities][4 waves][NS*64] float4
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_wrk[NS], a_wrv[NS], a_wcr[NS], a_w3[NS], wk[NS], wvv[NS], wc[NS], w3v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
        a_wrk[s] = z4; a_wrv[s] = z4; a_wcr[s] = z4; a_w3[s] = z4; wk[s] = z4; wvv[s] = z4; wc[s] = z4; w3v[s] = z4;
        if (c < H) {
            wk[s] = *(const float4*)(w_rk + c); wvv[s] = *(const float4*)(w_rv + c);
            wc[s] = *(const float4*)(wcr + c); w3v[s] = *(const float4*)(w3 + c);
        }
    }
    for (int r = blockIdx.x * 4 + wv; r < n_rows; r += gridDim.x * 4) {
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        if (e1 == e0) {                                  // no inter edges: d q = 0 (the caller does not pre-zero dqkv)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                if (c < H) *(float4*)(dqkv + (size_t)r * ldqkv + c) = z4;
            }
            continue;
        }
        float4 qv[NS], gh[NS], dq[NS];
        float qw = 0.f, gwv = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            qv[s] = z4; gh[s] = z4; dq[s] = z4;
            if (c < H) {
                qv[s] = *(const float4*)(qkv + (size_t)r * ldqkv + c);
                gh[s] = *(const float4*)(dh_out + (size_t)r * H + c);
                qw += qv[s].x * wk[s].x + qv[s].y * wk[s].y + qv[s].z * wk[s].z + qv[s].w * wk[s].w;
                gwv += gh[s].x * wvv[s].x + gh[s].y * wvv[s].y + gh[s].z * wvv[s].z + gh[s].w * wvv[s].w;
            }
        }
        qw = wave_sum(qw); gwv = wave_sum(gwv);
        // recompute t_r for the clamp mask
        float tx = 0.f, ty = 0.f, tz = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) {
            const float w = alpha[e] * cvs[e];
            tx += w * d[(size_t)e * 3]; ty += w * d[(size_t)e * 3 + 1]; tz += w * d[(size_t)e * 3 + 2];
        }
        tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
        const float gx = fabsf(tx) <= clampv ? dx_out[(size_t)r * 3] : 0.f;
        const float gy = fabsf(ty) <= clampv ? dx_out[(size_t)r * 3 + 1] : 0.f;
        const float gz = fabsf(tz) <= clampv ? dx_out[(size_t)r * 3 + 2] : 0.f;
        // pass 1: dalpha_e, stored in dlogit[]; S = sum alpha dalpha
        // Both edge loops are latency-bound on the gathered rows: the rows of edge j + 1 are requested before edge j is reduced; every
        // per-edge scalar (sending node, rhohat, alpha, the coordinate scalar, d) is loaded for up to 64 edges at once, one edge per
        // lane, and handed out with v_readlane; per-edge results collect in the lane of their edge and leave as coalesced stores
        // (see inter_attn_fwd_kernel: inside the loop those loads were ~2 us of exposed latency per edge of a ~27-edge chain).
        float S = 0.f;
        float4 pa[NS], pb[NS];
        auto fetch_v = [&](int cn) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                pa[s] = z4;
                if (c < H) pa[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + 2 * H + c);
            }
        };
        for (int eb = e0; eb < e1; eb += 64) {
            const int ne = min(64, e1 - eb);
            int my_col = 0;
            float my_rh = 0.f, my_al = 0.f, my_cvs = 0.f, my_gd = 0.f, my_da = 0.f;
            if (lane < ne) {
                const int e = eb + lane;
                my_col = col[e]; my_rh = rhohat[e]; my_al = alpha[e]; my_cvs = cvs[e];
                my_gd = gx * d[(size_t)e * 3] + gy * d[(size_t)e * 3 + 1] + gz * d[(size_t)e * 3 + 2];
            }
            fetch_v(__shfl(my_col, 0, 64));
            for (int j = 0; j < ne; ++j) {
                const float rh = __shfl(my_rh, j, 64);
                float hp = 0.f;
                float4 vv[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) vv[s] = pa[s];
                if (j + 1 < ne) fetch_v(__shfl(my_col, j + 1, 64));
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    hp += gh[s].x * vv[s].x + gh[s].y * vv[s].y + gh[s].z * vv[s].z + gh[s].w * vv[s].w;
                hp = wave_sum(hp) + rh * gwv;
                const float da = hp + __shfl(my_cvs, j, 64) * __shfl(my_gd, j, 64);
                S += __shfl(my_al, j, 64) * da;
                if (lane == j) my_da = da;
            }
            if (lane < ne) dlogit[eb + lane] = my_da;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // pass 2
        float s_lrh = 0.f, s_arh = 0.f;
        auto fetch_kc = [&](int cn) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                pa[s] = z4; pb[s] = z4;
                if (c < H) {
                    pa[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + H + c);
                    pb[s] = *(const float4*)(cv + (size_t)cn * ldcv + c);
                }
            }
        };
        for (int eb = e0; eb < e1; eb += 64) {
            const int ne = min(64, e1 - eb);
            int my_col = 0;
            float my_rh = 0.f, my_al = 0.f, my_cvs = 0.f, my_gd = 0.f, my_da = 0.f, o_dl = 0.f, o_dc = 0.f, o_drh = 0.f;
            if (lane < ne) {
                const int e = eb + lane;
                my_col = col[e]; my_rh = rhohat[e]; my_al = alpha[e]; my_cvs = cvs[e]; my_da = dlogit[e];
                my_gd = gx * d[(size_t)e * 3] + gy * d[(size_t)e * 3 + 1] + gz * d[(size_t)e * 3 + 2];
            }
            fetch_kc(__shfl(my_col, 0, 64));
            for (int j = 0; j < ne; ++j) {
                const float rh = __shfl(my_rh, j, 64), al = __shfl(my_al, j, 64);
                float4 kks[NS], ccs[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) { kks[s] = pa[s]; ccs[s] = pb[s]; }
                if (j + 1 < ne) fetch_kc(__shfl(my_col, j + 1, 64));
                const float dl = al * (__shfl(my_da, j, 64) - S);
                const float dc = al * __shfl(my_gd, j, 64);
                float dsum = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = s * 256 + lane * 4;
                    if (c < H) {
                        const float4 kk = kks[s], cc = ccs[s];
                        dq[s].x += dl * (kk.x + rh * wk[s].x); dq[s].y += dl * (kk.y + rh * wk[s].y);
                        dq[s].z += dl * (kk.z + rh * wk[s].z); dq[s].w += dl * (kk.w + rh * wk[s].w);
                        const float ux = cc.x + rh * wc[s].x, uy = cc.y + rh * wc[s].y, uz = cc.z + rh * wc[s].z, uw = cc.w + rh * wc[s].w;
                        const float px = w3v[s].x * dsilu_f(ux), py = w3v[s].y * dsilu_f(uy), pz = w3v[s].z * dsilu_f(uz), pw = w3v[s].w * dsilu_f(uw);
                        dsum += px * wc[s].x + py * wc[s].y + pz * wc[s].z + pw * wc[s].w;
                        a_wcr[s].x += dc * rh * px; a_wcr[s].y += dc * rh * py; a_wcr[s].z += dc * rh * pz; a_wcr[s].w += dc * rh * pw;
                        a_w3[s].x += dc * silu_f(ux); a_w3[s].y += dc * silu_f(uy); a_w3[s].z += dc * silu_f(uz); a_w3[s].w += dc * silu_f(uw);
                    }
                }
                dsum = wave_sum(dsum);
                s_lrh += dl * rh; s_arh += al * rh;
                if (lane == j) { o_dl = dl; o_dc = dc; o_drh = dl * qw + al * gwv + dc * dsum; }
            }
            if (lane < ne) {
                const int e = eb + lane;
                dlogit[e] = o_dl;
                dcp[e] = o_dc;
                drh[e] = o_drh;
                const float w = my_al * my_cvs;
                dd[(size_t)e * 3] = w * gx; dd[(size_t)e * 3 + 1] = w * gy; dd[(size_t)e * 3 + 2] = w * gz;
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            if (c < H) {
                *(float4*)(dqkv + (size_t)r * ldqkv + c) = dq[s];
                a_wrk[s].x += s_lrh * qv[s].x; a_wrk[s].y += s_lrh * qv[s].y; a_wrk[s].z += s_lrh * qv[s].z; a_wrk[s].w += s_lrh * qv[s].w;
                a_wrv[s].x += s_arh * gh[s].x; a_wrv[s].y += s_arh * gh[s].y; a_wrv[s].z += s_arh * gh[s].z; a_wrv[s].w += s_arh * gh[s].w;
            }
        }
    }
    // block partials of the four vector gradients: [gridDim.x][4][H]
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        shp[(0 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wrk[s];
        shp[(1 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wrv[s];
        shp[(2 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wcr[s];
        shp[(3 * 4 + wv) * NS * 64 + s * 64 + lane] = a_w3[s];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * NS * 64; idx += 256) {
        const int qn = idx / (NS * 64), rem = idx % (NS * 64);
        const int c = (rem / 64) * 256 + (rem % 64) * 4;
        if (c >= H) continue;
        float4 t = shp[(qn * 4 + 0) * NS * 64 + rem];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const float4 o = shp[(qn * 4 + w2) * NS * 64 + rem];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *(float4*)(wpart + ((size_t)blockIdx.x * 4 + qn) * H + c) = t;      // [block][4][H]: ONE column sum over [nblk, 4H] finishes all four
    }
}

template <int NS>
__global__ __launch_bounds__(256) void inter_attn_bwd_b_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* rhohat,
    const int* rowptr, const int* col, const int* mirror, const int* red_idx, const float* __restrict__ wcr,
    const float* __restrict__ w3, const float* alpha, const float* dlogit, const float* dcp, int n_rows, int n_lig_flag,
    const float* __restrict__ dh_out, float* dqkv, float* dcv, float* dbias_red, const int* node_is_lig) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rows) return;
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dk[NS], dv[NS], dc[NS], cvn[NS], wc[NS], w3v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
        dk[s] = z4; dv[s] = z4; dc[s] = z4; cvn[s] = z4; wc[s] = z4; w3v[s] = z4;
        if (c < H) {
            cvn[s] = *(const float4*)(cv + (size_t)n * ldcv + c);
            wc[s] = *(const float4*)(wcr + c); w3v[s] = *(const float4*)(w3 + c);
        }
    }
    for (int eb = e0; eb < e1; eb += 64) {             // per-edge scalars (through mirror[]: dependent loads) one edge per lane, ahead of the loop
        const int ne = min(64, e1 - eb);
        int m_l = 0;
        float dl_l = 0.f, al_l = 0.f, dcm_l = 0.f, rh_l = 0.f;
        if (lane < ne) {
            const int e = eb + lane, me = mirror[e];   // edge me = (m <- n)
            m_l = col[e];
            dl_l = dlogit[me]; al_l = alpha[me]; dcm_l = dcp[me]; rh_l = rhohat[me];
            if (n < m_l) dbias_red[red_idx[e]] = dlogit[e] + dl_l;   // ligand rows precede their proteins
        }
#pragma unroll 2
        for (int j = 0; j < ne; ++j) {                 // independent gathers: two edges in flight
            const int m = __shfl(m_l, j, 64);
            const float dl = __shfl(dl_l, j, 64), al = __shfl(al_l, j, 64), dcm = __shfl(dcm_l, j, 64), rh = __shfl(rh_l, j, 64);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                if (c < H) {
                    const float4 qm = *(const float4*)(qkv + (size_t)m * ldqkv + c);
                    const float4 gm = *(const float4*)(dh_out + (size_t)m * H + c);
                    dk[s].x += dl * qm.x; dk[s].y += dl * qm.y; dk[s].z += dl * qm.z; dk[s].w += dl * qm.w;
                    dv[s].x += al * gm.x; dv[s].y += al * gm.y; dv[s].z += al * gm.z; dv[s].w += al * gm.w;
                    dc[s].x += dcm * w3v[s].x * dsilu_f(cvn[s].x + rh * wc[s].x);
                    dc[s].y += dcm * w3v[s].y * dsilu_f(cvn[s].y + rh * wc[s].y);
                    dc[s].z += dcm * w3v[s].z * dsilu_f(cvn[s].z + rh * wc[s].z);
                    dc[s].w += dcm * w3v[s].w * dsilu_f(cvn[s].w + rh * wc[s].w);
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
        if (c < H) {
            *(float4*)(dqkv + (size_t)n * ldqkv + H + c) = dk[s];
            *(float4*)(dqkv + (size_t)n * ldqkv + 2 * H + c) = dv[s];
            *(float4*)(dcv + (size_t)n * ldcv + c) = dc[s];
        }
    }
}

extern "C" int fabind_inter_attn_bwd(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* d,
                                     const float* rhohat, const int* rowptr, const int* col, const int* mirror,
                                     const int* red_idx, const float* w_rk, const float* w_rv, const float* wcr,
                                     const float* w3, const float* alpha, const float* cvs, float clampv, int n_rows,
                                     const float* dh_out, const float* dx_out, float* dqkv, float* dcv, float* dd,
                                     float* drh, float* dbias_red, float* dlogit, float* dcp, float* wpart, int nblk,
                                     hipStream_t stream) {
    if (n_rows <= 0) return 0;
    FB_REQUIRE(H % 4 == 0 && H <= 1024, "fabind_inter_attn_bwd: H % 4, H <= 1024");
    dim3 gb((n_rows + 3) / 4);
#define LA(NS) do {                                                                                                    \
        size_t lds = (size_t)16 * NS * 64 * sizeof(float4);                                                            \
        hipLaunchKernelGGL((inter_attn_bwd_a_kernel<NS>), dim3(nblk), dim3(256), lds, stream, qkv, ldqkv, cv, ldcv, H, d, \
                           rhohat, rowptr, col, w_rk, w_rv, wcr, w3, alpha, cvs, clampv, n_rows, dh_out, dx_out, dqkv, dd, \
                           drh, dlogit, dcp, wpart);                                                                    \
        hipLaunchKernelGGL((inter_attn_bwd_b_kernel<NS>), gb, dim3(256), 0, stream, qkv, ldqkv, cv, ldcv, H, rhohat, rowptr, \
                           col, mirror, red_idx, wcr, w3, alpha, dlogit, dcp, n_rows, 0, dh_out, dqkv, dcv, dbias_red,   \
                           (const int*)nullptr);                                                                        \
    } while (0)
    if (H <= 256) LA(1); else if (H <= 512) LA(2); else LA(4);
#undef LA
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LAS step backward: x_out_j = x_j + clamp(step F_j), F_j = sum_{(i,j)} 4(|d|^2 - |d0|^2) d, d = x_i - x_j
// ------------------------------------------------------------------------------------------------
// (one wave per ligand atom, lanes over the complex's LAS edges, fixed-order wave sums; other nodes: a copy, one thread each -- the
//  thread-per-atom form walked all 154 edges serially: 76 us per call at 2,560 atoms)
__global__ __launch_bounds__(256) FB_NO_PACKED_F32 void las_step_bwd_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ xo, const int* las_i,
                                                           const int* las_j, const int* las_off, const int* node_off,
                                                           const int* c_cnt, float step, float clampv,
                                                           const float* __restrict__ dout, float* dx, int nb_copy) {
    const int b = blockIdx.y;
    const int off = node_off[b], n = node_off[b + 1] - off, C = c_cnt[b];
    if ((int)blockIdx.x < nb_copy) {
        const int u = blockIdx.x * 256 + threadIdx.x;
        if (u >= n || (u >= 1 && u < C)) return;
        const int gu = off + u;
        dx[(size_t)gu * 3] = dout[(size_t)gu * 3]; dx[(size_t)gu * 3 + 1] = dout[(size_t)gu * 3 + 1]; dx[(size_t)gu * 3 + 2] = dout[(size_t)gu * 3 + 2];
        return;
    }
    const int lane = threadIdx.x & 63;
    const int u = 1 + ((int)blockIdx.x - nb_copy) * 4 + (threadIdx.x >> 6);
    if (u >= C || u >= n) return;
    const int gu = off + u;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int e = las_off[b] + lane; e < las_off[b + 1]; e += 64) {
        const int i = las_i[e], j = las_j[e];
        if (i != gu && j != gu) continue;
        // dF_j = step * dout_j * [clamp inactive]
        float gF[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float delta = xo[(size_t)j * 3 + k] - x[(size_t)j * 3 + k];
            gF[k] = fabsf(delta) < clampv ? step * dout[(size_t)j * 3 + k] : 0.f;
        }
        const float dx_ = x[i * 3] - x[j * 3], dy_ = x[i * 3 + 1] - x[j * 3 + 1], dz_ = x[i * 3 + 2] - x[j * 3 + 2];
        const float ex = x0[i * 3] - x0[j * 3], ey = x0[i * 3 + 1] - x0[j * 3 + 1], ez = x0[i * 3 + 2] - x0[j * 3 + 2];
        const float diff = (dx_ * dx_ + dy_ * dy_ + dz_ * dz_) - (ex * ex + ey * ey + ez * ez);
        const float dg = dx_ * gF[0] + dy_ * gF[1] + dz_ * gF[2];
        const float vx = 4.f * (diff * gF[0] + 2.f * dx_ * dg), vy = 4.f * (diff * gF[1] + 2.f * dy_ * dg),
                    vz = 4.f * (diff * gF[2] + 2.f * dz_ * dg);
        // (no packed fp32 math in this kernel: see las_step_kernel in attn.hip)
        const float sg = (i == gu ? 1.f : 0.f) - (j == gu ? 1.f : 0.f);
        ax = __builtin_fmaf(sg, vx, ax);
        ay = __builtin_fmaf(sg, vy, ay);
        az = __builtin_fmaf(sg, vz, az);
    }
    ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
    if (lane == 0) {
        dx[(size_t)gu * 3] = dout[(size_t)gu * 3] + ax; dx[(size_t)gu * 3 + 1] = dout[(size_t)gu * 3 + 1] + ay;
        dx[(size_t)gu * 3 + 2] = dout[(size_t)gu * 3 + 2] + az;
    }
}
extern "C" int fabind_las_step_bwd(const float* x, const float* x0, const float* xo, const int* las_i, const int* las_j,
                                   const int* las_off, const int* node_off, const int* c_cnt, int B, int max_n,
                                   float step, float clampv, const float* dout, float* dx, hipStream_t stream) {
    if (B <= 0) return 0;
    const int nb_copy = (max_n + 255) / 256;
    hipLaunchKernelGGL(las_step_bwd_kernel, dim3(nb_copy + (max_n + 3) / 4, B), dim3(256), 0, stream, x, x0, xo, las_i, las_j,
                       las_off, node_off, c_cnt, step, clampv, dout, dx, nb_copy);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// pair-bias backward (one attention block): out[(b,i,j), o] = sum_h a0[i,h] b0[j,h] w[o,h]
//   kernel A: da0[i,h] += sum_{j,o} D[i,j,o] w[o,h] b0[j,h]          (block = 8 protein rows, threads over h)
//   kernel B: T[o] = sum_i D[i,j,o] a0[i,h];  db0[j,h] += sum_o w[o,h] T[o];  dw[b][o,h] += b0[j,h] T[o]
// a0 = a0b0[:, 0:H] at protein-side nodes, b0 = a0b0[:, H:2H] at ligand-side nodes. NO == 8.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_bias_bwd_a_kernel(const float* __restrict__ dout, const float* __restrict__ ab,
                                                              int ld, int H, const float* __restrict__ w, const int* desc,
                                                              const int* p_index, const int* c_index, float* dab) {
    const int* ds = desc + blockIdx.y * 8;
    const int p_off = ds[0], P = ds[1], c_off = ds[2], C = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int i0 = blockIdx.x * 8;
    if (i0 >= P) return;
    for (int h = threadIdx.x; h < H; h += 256) {
        float wv[8], acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) { wv[o] = w[o * H + h]; acc[o] = 0.f; }
        for (int j = 0; j < C; ++j) {
            const float bj = ab[(size_t)c_index[c_off + j] * ld + H + h];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                if (i0 + ii >= P) break;
                const float* Dp = dout + ((size_t)pair_off + (size_t)(i0 + ii) * C + j) * 8;
                float t = 0.f;
#pragma unroll
                for (int o = 0; o < 8; ++o) t += Dp[o] * wv[o];
                acc[ii] += t * bj;
            }
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii)
            if (i0 + ii < P) dab[(size_t)p_index[p_off + i0 + ii] * ld + h] += acc[ii];
    }
}
#define PB_JT 4
__global__ __launch_bounds__(256) void pair_bias_bwd_b_kernel(const float* __restrict__ dout, const float* __restrict__ ab,
                                                              int ld, int H, const float* __restrict__ w, const int* desc,
                                                              const int* p_index, const int* c_index, float* dab,
                                                              float* dwk) {
    // block = PB_JT ligand-side nodes of one complex; threads over h; a0[i,h] is read once per PB_JT columns
    const int b = blockIdx.y;
    const int* ds = desc + b * 8;
    const int p_off = ds[0], P = ds[1], c_off = ds[2], C = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int j0 = blockIdx.x * PB_JT;
    if (j0 >= C) return;
    for (int h = threadIdx.x; h < H; h += 256) {
        float T[PB_JT][8];
#pragma unroll
        for (int jj = 0; jj < PB_JT; ++jj)
#pragma unroll
            for (int o = 0; o < 8; ++o) T[jj][o] = 0.f;
        // protein-side rows of a complex are contiguous nodes: p_index[p_off + i] = p_index[p_off] + i
        const float* a0 = ab + (size_t)p_index[p_off] * ld + h;
        const int jn = min(PB_JT, C - j0);
        int i = 0;
        for (; i + 4 <= P; i += 4) {              // 4 independent a0 loads in flight per lane
            float ai[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ai[u] = a0[(size_t)(i + u) * ld];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* Dp = dout + ((size_t)pair_off + (size_t)(i + u) * C + j0) * 8;
#pragma unroll
                for (int jj = 0; jj < PB_JT; ++jj) {
                    if (jj < jn) {
                        const float4 d0 = *(const float4*)(Dp + jj * 8), d1 = *(const float4*)(Dp + jj * 8 + 4);
                        T[jj][0] += d0.x * ai[u]; T[jj][1] += d0.y * ai[u]; T[jj][2] += d0.z * ai[u]; T[jj][3] += d0.w * ai[u];
                        T[jj][4] += d1.x * ai[u]; T[jj][5] += d1.y * ai[u]; T[jj][6] += d1.z * ai[u]; T[jj][7] += d1.w * ai[u];
                    }
                }
            }
        }
        for (; i < P; ++i) {
            const float av = a0[(size_t)i * ld];
            const float* Dp = dout + ((size_t)pair_off + (size_t)i * C + j0) * 8;
#pragma unroll
            for (int jj = 0; jj < PB_JT; ++jj) {
                if (jj < jn) {
                    const float4 d0 = *(const float4*)(Dp + jj * 8), d1 = *(const float4*)(Dp + jj * 8 + 4);
                    T[jj][0] += d0.x * av; T[jj][1] += d0.y * av; T[jj][2] += d0.z * av; T[jj][3] += d0.w * av;
                    T[jj][4] += d1.x * av; T[jj][5] += d1.y * av; T[jj][6] += d1.z * av; T[jj][7] += d1.w * av;
                }
            }
        }
        float wv[8], dwa[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) { wv[o] = w[o * H + h]; dwa[o] = 0.f; }
#pragma unroll
        for (int jj = 0; jj < PB_JT; ++jj) {
            if (j0 + jj >= C) break;
            const int cn = c_index[c_off + j0 + jj];
            const float bj = ab[(size_t)cn * ld + H + h];
            float db = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) { db += wv[o] * T[jj][o]; dwa[o] += bj * T[jj][o]; }
            dab[(size_t)cn * ld + H + h] += db;
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) atomicAdd(&dwk[((size_t)b * 8 + o) * H + h], dwa[o]);   // C/PB_JT adds per element
    }
}
extern "C" int fabind_pair_bias_bwd(const float* dout, int NO, const float* ab, int ld, int H, const float* w,
                                    const int* desc_p, int B, int max_P, int max_C, const int* p_index,
                                    const int* c_index, float* dab, float* dwk, hipStream_t stream) {
    FB_REQUIRE(NO == 8, "fabind_pair_bias_bwd: NO must be 8");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pair_bias_bwd_a_kernel, dim3((max_P + 7) / 8, B), dim3(256), 0, stream, dout, ab, ld, H, w, desc_p,
                       p_index, c_index, dab);
    hipLaunchKernelGGL(pair_bias_bwd_b_kernel, dim3((max_C + PB_JT - 1) / PB_JT, B), dim3(256), 0, stream, dout, ab, ld, H, w, desc_p, p_index,
                       c_index, dab, dwk);
    FB_CHECK_LAUNCH();
    return 0;
}

// finishing pass of the GEMM formulation of the pair-bias adjoint:  T[(jc,o), h] = sum_i D[i,(j,o)] a0[i,h]
//   db0[node(jc), h] += sum_o w[o,h] T[(jc,o),h];   dw_part[wg][o,h] = sum_{jc in wg} b0[node(jc),h] T[(jc,o),h]
// One work-group = 16 ligand-side nodes x 256 features: the weight gradient is summed over the work-group's nodes in registers and WRITTEN
// as this work-group's partial row (row stride ld_part: the caller lays the partials of all blocks side by side and sums the rows with
// ONE fabind_colsum) -- a fixed summation order.  (Round 3 added the partial with one float atomic per (o, h) and work-group: two runs
// of the same step differed in the last bit of these gradients, tests/test_gpu_headline.py test_bench_batch_backward_properties.  A
// "last work-group reduces" form with a device-scope fence cost 98 us instead of 14: the fence writes the XCD's L2 back.)
// (First form: one work-group per node, 8 atomics per node and feature -- 10.7 M float atomics per launch onto 4,096 addresses at the
// bench shape, 152 us for a 43 MB read; 8 launches per step.)
constexpr int PBF_NODES = 16;
__global__ __launch_bounds__(256) void pair_bias_finish_kernel(const float* __restrict__ T, const float* __restrict__ ab, int ld,
                                                               int H, const float* __restrict__ w, const int* c_index, int n_c,
                                                               float* dab, float* dw_part, long ld_part) {
    const int h = blockIdx.y * 256 + threadIdx.x;
    if (h >= H) return;
    const int j0 = blockIdx.x * PBF_NODES, j1 = min(n_c, j0 + PBF_NODES);
    float wv[8], dwv[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) { wv[o] = w[o * H + h]; dwv[o] = 0.f; }
    for (int jc = j0; jc < j1; ++jc) {
        const int cn = c_index[jc];
        const float bj = ab[(size_t)cn * ld + H + h];
        float db = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float t = T[((size_t)jc * 8 + o) * H + h];
            db += wv[o] * t;
            dwv[o] += bj * t;
        }
        dab[(size_t)cn * ld + H + h] += db;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) dw_part[(size_t)blockIdx.x * ld_part + o * H + h] = dwv[o];
}
extern "C" int fabind_pair_bias_finish_parts(int n_c) { return n_c <= 0 ? 0 : (n_c + PBF_NODES - 1) / PBF_NODES; }
extern "C" int fabind_pair_bias_finish(const float* T, const float* ab, int ld, int H, const float* w, const int* c_index,
                                       int n_c, float* dab, float* dw_part, long ld_part, hipStream_t stream) {
    if (n_c <= 0) return 0;
    FB_REQUIRE(dw_part && ld_part >= 8L * H, "fabind_pair_bias_finish: dw_part [fabind_pair_bias_finish_parts(n_c)] rows of >= 8 * H floats");
    hipLaunchKernelGGL(pair_bias_finish_kernel, dim3((n_c + PBF_NODES - 1) / PBF_NODES, (H + 255) / 256), dim3(256), 0, stream, T, ab, ld, H, w,
                       c_index, n_c, dab, dw_part, ld_part);
    FB_CHECK_LAUNCH();
    return 0;
}
