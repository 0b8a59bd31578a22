// Intra-graph message passing pieces (MC_E_GCL, reference models/egnn.py:68-144) for gfx950.
// HBM-bound gather / segmented-reduction kernels: 16-byte per-lane accesses, one wave walks one
// 2 KiB feature row (H = 512 fp32), CSR rows are reduced without atomics (deterministic).
#include "common.h"
#include "fabind_hip.h"

// pre[e,:] = act(AB[row[e], 0:H] + AB[col[e], H:2H] + rhohat[e] * w_r[:]),  dact_out[e,:] = act'(...)  (optional)
// Generic shape: one wave per edge, 4 columns per lane.
__global__ __launch_bounds__(256) void gcl_pre_kernel(const void* __restrict__ AB, int ab_dt, int ldab, int H,
                                                      const int* row, const int* col, const float* rhohat,
                                                      const float* __restrict__ w_r, void* pre, int pre_dt, int E, int act,
                                                      void* dact_out) {
    const int lane = threadIdx.x & 63;
    int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const size_t a = (size_t)row[e] * ldab, b = (size_t)col[e] * ldab + H;
    const float rh = rhohat[e];
    for (int c = lane * 4; c < H; c += 256) {
        float4 va = ld4_any(AB, ab_dt, a + c), vb = ld4_any(AB, ab_dt, b + c), w = *(const float4*)(w_r + c);
        const float4 z = make_float4(va.x + vb.x + rh * w.x, va.y + vb.y + rh * w.y, va.z + vb.z + rh * w.z,
                                     va.w + vb.w + rh * w.w);
        st4_any(pre, pre_dt, (size_t)e * H + c,
                make_float4(apply_act(z.x, act), apply_act(z.y, act), apply_act(z.z, act), apply_act(z.w, act)));
        if (dact_out)
            st4_any(dact_out, pre_dt, (size_t)e * H + c,
                    make_float4(apply_dact(z.x, act), apply_dact(z.y, act), apply_dact(z.z, act), apply_dact(z.w, act)));
    }
}

// H = 8 * LPE <= 512: LPE lanes cover one edge with 8 columns each (16-byte bf16 loads and stores), a wave covers
// 64/LPE edges per pass and GP_PASSES passes with every gather of all passes issued before the first use.
#define GP_PASSES 4
template <bool BF16> struct GpRaw;
template <> struct GpRaw<true> {
    uint4 u;
    __device__ __forceinline__ void load(const void* p, size_t i) { u = *(const uint4*)((const bf16_t*)p + i); }
    __device__ __forceinline__ void get(float* f) const {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
};
template <> struct GpRaw<false> {
    float4 a, b;
    __device__ __forceinline__ void load(const void* p, size_t i) {
        a = *(const float4*)((const float*)p + i);
        b = *(const float4*)((const float*)p + i + 4);
    }
    __device__ __forceinline__ void get(float* f) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};
template <int LPE, bool AB16, bool OUT16, bool DACT>
__global__ __launch_bounds__(256) void gcl_pre8_kernel(const void* __restrict__ AB, int ldab,
                                                       const int* __restrict__ row, const int* __restrict__ col,
                                                       const float* __restrict__ rhohat, const float* __restrict__ w_r,
                                                       void* pre, int E, int act, void* dact_out) {
    constexpr int H = LPE * 8, EPP = 64 / LPE;
    constexpr int odt = OUT16 ? FB_DT_BF16 : FB_DT_F32;
    const int lane = threadIdx.x & 63, sub = lane / LPE, c = (lane % LPE) * 8;
    const int ebase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (EPP * GP_PASSES) + sub;
    if (ebase - sub >= E) return;
    GpRaw<AB16> va[GP_PASSES], vb[GP_PASSES];
    float rh[GP_PASSES];
#pragma unroll
    for (int i = 0; i < GP_PASSES; ++i) {
        const int e = min(ebase + i * EPP, E - 1);
        const int r = row[e], q = col[e];
        rh[i] = rhohat[e];
        va[i].load(AB, (size_t)r * ldab + c);
        vb[i].load(AB, (size_t)q * ldab + H + c);
    }
    const float4 w0 = *(const float4*)(w_r + c), w1 = *(const float4*)(w_r + c + 4);
    const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int i = 0; i < GP_PASSES; ++i) {
        const int e = ebase + i * EPP;
        if (e < E) {
            float fa[8], fb[8];
            va[i].get(fa);
            vb[i].get(fb);
            F8 y, dy;
            if (act == FB_ACT_SILU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float z = fa[k] + fb[k] + rh[i] * w[k];
                    const float sg = sigmoid_f(z);
                    y.v[k] = z * sg;
                    if (DACT) dy.v[k] = sg * (1.0f + z * (1.0f - sg));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float z = fa[k] + fb[k] + rh[i] * w[k];
                    y.v[k] = apply_act(z, act);
                    if (DACT) dy.v[k] = apply_dact(z, act);
                }
            }
            st8_any(pre, odt, (size_t)e * H + c, y);
            if (DACT) st8_any(dact_out, odt, (size_t)e * H + c, dy);
        }
    }
}

extern "C" int fabind_gcl_pre(const void* AB, int ab_dt, int ldab, int H, const int* row, const int* col,
                              const float* rhohat, const float* w_r, void* pre, int pre_dt, int E, int act,
                              void* dact_out, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && ldab % 4 == 0, "fabind_gcl_pre: H and ldab must be multiples of 4");
    if (E <= 0) return 0;
    const bool vec8 = ldab % 8 == 0 && ((uintptr_t)AB % 16 == 0) && ((uintptr_t)pre % 16 == 0) &&
                      (!dact_out || (uintptr_t)dact_out % 16 == 0) && ((uintptr_t)w_r % 16 == 0);
#define GP_LAUNCH4(LPE_, A16_, O16_, D_)                                                                             \
    hipLaunchKernelGGL((gcl_pre8_kernel<LPE_, A16_, O16_, D_>),                                                      \
                       dim3((E + 4 * (64 / LPE_) * GP_PASSES - 1) / (4 * (64 / LPE_) * GP_PASSES)), dim3(256), 0, stream, \
                       AB, ldab, row, col, rhohat, w_r, pre, E, act, dact_out)
#define GP_LAUNCH(LPE_)                                                                                              \
    do {                                                                                                             \
        const bool a16 = ab_dt == FB_DT_BF16, o16 = pre_dt == FB_DT_BF16;                                            \
        if (a16 && o16 && dact_out) GP_LAUNCH4(LPE_, true, true, true);                                              \
        else if (a16 && o16) GP_LAUNCH4(LPE_, true, true, false);                                                    \
        else if (!a16 && o16 && dact_out) GP_LAUNCH4(LPE_, false, true, true);                                       \
        else if (!a16 && o16) GP_LAUNCH4(LPE_, false, true, false);                                                  \
        else if (!a16 && !o16 && dact_out) GP_LAUNCH4(LPE_, false, false, true);                                     \
        else if (!a16 && !o16) GP_LAUNCH4(LPE_, false, false, false);                                                \
        else if (dact_out) GP_LAUNCH4(LPE_, true, false, true);                                                      \
        else GP_LAUNCH4(LPE_, true, false, false);                                                                   \
    } while (0)
    if (vec8 && H == 512) GP_LAUNCH(64);
    else if (vec8 && H == 256) GP_LAUNCH(32);
    else if (vec8 && H == 128) GP_LAUNCH(16);
    else if (vec8 && H == 64) GP_LAUNCH(8);
    else
        hipLaunchKernelGGL(gcl_pre_kernel, dim3((E + 3) / 4), dim3(256), 0, stream, AB, ab_dt, ldab, H, row, col, rhohat,
                           w_r, pre, pre_dt, E, act, dact_out);
#undef GP_LAUNCH
#undef GP_LAUNCH4
    FB_CHECK_LAUNCH();
    return 0;
}

// out[r,:] = sum_{e in [rowptr[r], rowptr[r+1])} act(Z[eidx ? eidx[e] : e,:])
// One 256-thread work-group per row: its 4 waves take edges round-robin, each lane owns 8 consecutive columns
// per 512-column slab (16-byte loads for bf16); partials are combined through LDS in fixed wave order.

// One wave per CSR row (4 rows per work-group): no LDS, no barrier, few registers (8 waves/SIMD).  Each lane owns 8
// consecutive columns per 512-column slab (16-byte bf16 loads), 4 independent row loads in flight per lane.
template <int NSLAB, int U>
__device__ __forceinline__ void ss_accum(const void* __restrict__ Z, int z_dt, int ldz, int H, const int* eidx, int act,
                                         int lane, int e0, int e1, int estep, F8 (&acc)[NSLAB]) {
    int ee = e0;
    for (; ee + (U - 1) * estep < e1; ee += U * estep) {
        int e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = eidx ? eidx[ee + u * estep] : ee + u * estep;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int c = s * 512 + lane * 8;
            if (c < H) {
                F8 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld8_any(Z, z_dt, (size_t)e[u] * ldz + c);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[s].v[k] += apply_act(v[u].v[k], act);
            }
        }
    }
    if (ee < e1) {                                  // ragged last group, predicated (wave-uniform) -- one round trip
        int e[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = ee + u * estep < e1;
            e[u] = ok[u] ? (eidx ? eidx[ee + u * estep] : ee + u * estep) : 0;
        }
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int c = s * 512 + lane * 8;
            if (c < H) {
                F8 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ok[u]) v[u] = ld8_any(Z, z_dt, (size_t)e[u] * ldz + c);
                    else
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[u].v[k] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[s].v[k] += apply_act(v[u].v[k], act);
            }
        }
    }
}

template <int NSLAB>
__global__ __launch_bounds__(256) void segment_sum_kernel(const void* __restrict__ Z, int z_dt, int ldz, int H,
                                                          const int* rowptr, const int* eidx, int act, float* out,
                                                          int ldo, int n_rows, bf16_t* out16, int ldo16, int heavy_min) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    if (e1 - e0 > heavy_min) return;               // left to segment_sum_heavy_kernel
    F8 acc[NSLAB];
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[s].v[k] = 0.f;
    // (8 row loads in flight per lane at H <= 512: a 40-atom ligand's rows hold 40-128 edges, and the walk is bound by the latency of
    //  its dependent index -> row loads, not by bytes: 103 us per call at the pocket-sized bench shape with 4 in flight)
    ss_accum<NSLAB, 8 / NSLAB>(Z, z_dt, ldz, H, eidx, act, lane, e0, e1, 1, acc);
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        const int c = s * 512 + lane * 8;
        if (c < H) {
            if (out) {
                float* o = out + (size_t)r * ldo + c;
                *(float4*)o = make_float4(acc[s].v[0], acc[s].v[1], acc[s].v[2], acc[s].v[3]);
                *(float4*)(o + 4) = make_float4(acc[s].v[4], acc[s].v[5], acc[s].v[6], acc[s].v[7]);
            }
            if (out16) st8_any(out16, FB_DT_BF16, (size_t)r * ldo16 + c, acc[s]);
        }
    }
}

// Rows longer than heavy_min edges (the global nodes of a complex: ~1500 star edges): each 1024-thread work-group scans
// 1024 row lengths, then its 16 waves reduce every heavy row among them cooperatively (8 row loads in flight per
// wave), partials combined through LDS in fixed wave order.
template <int NSLAB>
__global__ __launch_bounds__(1024) void segment_sum_heavy_kernel(const void* __restrict__ Z, int z_dt, int ldz, int H,
                                                                 const int* rowptr, const int* eidx, int act, float* out,
                                                                 int ldo, int n_rows, bf16_t* out16, int ldo16, int rpw, int heavy_min) {
    // rpw (<= 1024): rows scanned per work-group.  1,024 suits 1,500-node complexes (one or two heavy rows per work-group); a batch of
    // 142-node complexes has 14 heavy rows per 1,024 and only 9 such work-groups on the whole chip (103 us per call at the
    // pocket-sized bench shape): the host sizes rpw so that the launch has >= ~256 work-groups.
    __shared__ F8 part[15][NSLAB * 64];
    __shared__ int heavy[1024];
    __shared__ int n_heavy;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) n_heavy = 0;
    __syncthreads();
    const int rr = blockIdx.x * rpw + threadIdx.x;
    if ((int)threadIdx.x < rpw && rr < n_rows && rowptr[rr + 1] - rowptr[rr] > heavy_min) heavy[atomicAdd(&n_heavy, 1)] = rr;
    __syncthreads();
    const int nh = n_heavy;
    for (int q = 0; q < nh; ++q) {
        const int r = heavy[q];
        F8 acc[NSLAB];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[s].v[k] = 0.f;
        ss_accum<NSLAB, 8 / NSLAB>(Z, z_dt, ldz, H, eidx, act, lane, rowptr[r] + w, rowptr[r + 1], 16, acc);
        if (w > 0) {
#pragma unroll
            for (int s = 0; s < NSLAB; ++s) part[w - 1][s * 64 + lane] = acc[s];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int s = 0; s < NSLAB; ++s) {
                for (int k = 0; k < 15; ++k) {
                    const F8 p = part[k][s * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[s].v[t] += p.v[t];
                }
                const int c = s * 512 + lane * 8;
                if (c < H) {
                    if (out) {
                        float* o = out + (size_t)r * ldo + c;
                        *(float4*)o = make_float4(acc[s].v[0], acc[s].v[1], acc[s].v[2], acc[s].v[3]);
                        *(float4*)(o + 4) = make_float4(acc[s].v[4], acc[s].v[5], acc[s].v[6], acc[s].v[7]);
                    }
                    if (out16) st8_any(out16, FB_DT_BF16, (size_t)r * ldo16 + c, acc[s]);
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int fabind_segment_sum(const void* Z, int z_dt, int ldz, int H, const int* rowptr, const int* eidx,
                                  int n_rows, int n_edges, int act, float* out, int ldo, void* out16v, int ldo16,
                                  hipStream_t stream) {
    bf16_t* out16 = (bf16_t*)out16v;
    FB_REQUIRE(H % 8 == 0 && ldz % 8 == 0 && ldo % 4 == 0, "fabind_segment_sum: H/ldz must be multiples of 8, ldo of 4");
    FB_REQUIRE(H <= 1024, "fabind_segment_sum: H <= 1024");
    FB_REQUIRE(out != nullptr || out16 != nullptr, "fabind_segment_sum: no output");
    FB_REQUIRE(((uintptr_t)Z % 16 == 0) && ((uintptr_t)out % 16 == 0), "fabind_segment_sum: 16-byte alignment");
    FB_REQUIRE(out16 == nullptr || (ldo16 % 8 == 0 && (uintptr_t)out16 % 16 == 0), "fabind_segment_sum: out16 needs 16-byte alignment and ldo16 % 8 == 0");
    if (n_rows <= 0) return 0;
    (void)n_edges;
    int rpw = 1024;
    while (rpw > 64 && (n_rows + rpw - 1) / rpw < 256) rpw >>= 1;
    // rows one wave sums alone vs rows a 16-wave work-group sums together: 128 edges on large batches (the one-wave rows stream at the HBM
    // rate, 1,500-edge star rows are the exception); on small ones the launch is as long as its longest one-wave row -- the 128 global
    // nodes of a pocket-sized batch (101 edges each) made it 90 us for 87 MB, plain or permuted rows alike: 32 there (53 us).  (Not for
    // the 18,000-row batches of the full model's pocket crop / FABind+ training: there thousands of ligand rows hold 33-128 edges, and
    // sending them all through the cooperative kernel cost FABind+ training +2.4 ms per step.)
    const int heavy_min = n_rows < 16384 ? 32 : 128;
    const dim3 g((n_rows + 3) / 4), gh((n_rows + rpw - 1) / rpw);
    if (H <= 512) {
        hipLaunchKernelGGL((segment_sum_kernel<1>), g, dim3(256), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo, n_rows, out16, ldo16, heavy_min);
        hipLaunchKernelGGL((segment_sum_heavy_kernel<1>), gh, dim3(1024), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo, n_rows, out16, ldo16, rpw, heavy_min);
    } else {
        hipLaunchKernelGGL((segment_sum_kernel<2>), g, dim3(256), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo, n_rows, out16, ldo16, heavy_min);
        hipLaunchKernelGGL((segment_sum_heavy_kernel<2>), gh, dim3(1024), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo, n_rows, out16, ldo16, rpw, heavy_min);
    }
    FB_CHECK_LAUNCH();
    return 0;
}

// Coordinate update of a CSR row (one wave per row):
//   s[e]     = (sum_k s_part[e,k]) * (weight ? weight[e] : 1)
//   x_out[r] = x[r] + clamp( (mean ? 1/max(deg,1) : 1) * sum_e d[e]*s[e], +-clampv )
// s_out (optional) receives s[e] (saved for backward).
__global__ __launch_bounds__(256) void coord_update_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                           const float* __restrict__ s_part, int n_part,
                                                           const float* weight, const int* rowptr, int n_rows, int mean,
                                                           float clampv, float* x_out, float* s_out) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) {
        float s = 0.f;
        for (int k = 0; k < n_part; ++k) s += s_part[(size_t)e * n_part + k];
        if (weight) s *= weight[e];
        if (s_out) s_out[e] = s;
        ax += d[(size_t)e * 3] * s; ay += d[(size_t)e * 3 + 1] * s; az += d[(size_t)e * 3 + 2] * s;
    }
    ax = wave_sum(ax); ay = wave_sum(ay); az = wave_sum(az);
    if (lane == 0) {
        const int deg = e1 - e0;
        const float den = mean ? (float)(deg < 1 ? 1 : deg) : 1.f;
        float tx = ax / den, ty = ay / den, tz = az / den;
        tx = fminf(fmaxf(tx, -clampv), clampv); ty = fminf(fmaxf(ty, -clampv), clampv); tz = fminf(fmaxf(tz, -clampv), clampv);
        x_out[(size_t)r * 3] = x[(size_t)r * 3] + tx;
        x_out[(size_t)r * 3 + 1] = x[(size_t)r * 3 + 1] + ty;
        x_out[(size_t)r * 3 + 2] = x[(size_t)r * 3 + 2] + tz;
    }
}

extern "C" int fabind_coord_update(const float* x, const float* d, const float* s_part, int n_part,
                                   const float* weight, const int* rowptr, int n_rows, int mean, float clampv,
                                   float* x_out, float* s_out, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(coord_update_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, x, d, s_part, n_part, weight,
                       rowptr, n_rows, mean, clampv, x_out, s_out);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Rows of a CSR-reduced output that receive nothing: out[r, 0:C) = 0 for every r with rowptr[r] == rowptr[r+1].  The fused edge
// kernels write the sums of nodes WITH edges only; this replaces a memset of the whole [N, C] output (200 MB per layer at the bench
// shape, where no node is without edges) by a pass over N + 1 integers.  One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_empty_rows_kernel(const int* __restrict__ rowptr, int n_rows, void* out, int out_dt, int ld,
                                                              int C, void* out2, int out2_dt, int ld2) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows || rowptr[r] != rowptr[r + 1]) return;
    for (int c = lane; c < C; c += 64) {
        st_any(out, out_dt, (size_t)r * ld + c, 0.f);
        if (out2) st_any(out2, out2_dt, (size_t)r * ld2 + c, 0.f);
    }
}

extern "C" int fabind_zero_empty_rows(const int* rowptr, int n_rows, void* out, int out_dt, int ld, int C, void* out2, int out2_dt,
                                      int ld2, hipStream_t stream) {
    if (n_rows <= 0 || C <= 0) return 0;
    FB_REQUIRE(rowptr != nullptr && out != nullptr, "fabind_zero_empty_rows: rowptr / out");
    hipLaunchKernelGGL(zero_empty_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, rowptr, n_rows, out, out_dt, ld, C, out2,
                       out2_dt, ld2);
    FB_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Many weights -> bf16 MFMA-fragment order [K/32][N/16][4][16][8] in ONE launch (the operand packs of the fused edge kernels: W2, Wc and
// their transposes of every GCL layer -- 30 torch permute-copies per training step before).  A segment reads W'[n, k] =
// src[n * src_sr + k * src_sc] (a transposed source is a stride swap), fp32 or bf16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_frag_multi_kernel(const FabindPackSeg* __restrict__ segs) {
    const FabindPackSeg sg = segs[blockIdx.y];
    const int N = sg.N, Kd = sg.K;
    const long n_vec = (long)N * Kd / 8;
    bf16_t* dst = (bf16_t*)sg.dst;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n_vec; t += (long)gridDim.x * 256) {
        // t = ((kb * (N / 16) + nb) * 4 + q) * 16 + r
        const int r = (int)(t & 15), q = (int)((t >> 4) & 3);
        const long tb = t >> 6;
        const int nb = (int)(tb % (N / 16)), kb = (int)(tb / (N / 16));
        const long n = (long)nb * 16 + r, k0 = (long)kb * 32 + q * 8;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a, b;
            const long i0 = n * sg.src_sr + (k0 + 2 * j) * sg.src_sc, i1 = i0 + sg.src_sc;
            if (sg.src_dt == FB_DT_BF16) {
                a = bf16_to_f32(((const bf16_t*)sg.src)[i0]);
                b = bf16_to_f32(((const bf16_t*)sg.src)[i1]);
            } else {
                a = ((const float*)sg.src)[i0];
                b = ((const float*)sg.src)[i1];
                if (sg.src_dt == 2) {          // round 6: the LO plane of a split-bf16 operand: bf16(w - bf16(w)) of an fp32 source
                    const uint32_t h = pack2_bf16(a, b);
                    a -= __uint_as_float(h << 16);
                    b -= __uint_as_float(h & 0xffff0000u);
                }
            }
            w[j] = pack2_bf16(a, b);
        }
        *(uint4*)(dst + t * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
extern "C" int fabind_pack_frag_multi(const FabindPackSeg* segs_dev, const FabindPackSeg* segs_host, int n_segs, hipStream_t stream) {
    if (n_segs <= 0) return 0;
    FB_REQUIRE(segs_dev && segs_host, "fabind_pack_frag_multi: null table");
    long big = 0;
    for (int i = 0; i < n_segs; ++i) {
        const FabindPackSeg& g = segs_host[i];
        FB_REQUIRE(g.src && g.dst && g.N > 0 && g.K > 0 && g.N % 16 == 0 && g.K % 32 == 0 && ((uintptr_t)g.dst % 16 == 0) &&
                   (g.src_dt == FB_DT_F32 || g.src_dt == FB_DT_BF16 || g.src_dt == 2), "fabind_pack_frag_multi: N % 16 == 0, K % 32 == 0, 16-byte aligned dst");
        const long nv = (long)g.N * g.K / 8;
        big = nv > big ? nv : big;
    }
    const int bx = (int)((big + 255) / 256 < 256 ? (big + 255) / 256 : 256);
    hipLaunchKernelGGL(pack_frag_multi_kernel, dim3(bx, n_segs), dim3(256), 0, stream, segs_dev);
    FB_CHECK_LAUNCH();
    return 0;
}
