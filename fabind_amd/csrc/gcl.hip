// Intra-graph message passing pieces (MC_E_GCL, reference models/egnn.py:68-144) for gfx950.
// HBM-bound gather / segmented-reduction kernels: 16-byte per-lane accesses, one wave walks one
// 2 KiB feature row (H = 512 fp32), CSR rows are reduced without atomics (deterministic).
#include "common.h"
#include "fabind_hip.h"

// pre[e,:] = AB[row[e], 0:H] + AB[col[e], H:2H] + rhohat[e] * w_r[:]       (one wave per edge)
__global__ __launch_bounds__(256) void gcl_pre_kernel(const float* __restrict__ AB, int ldab, int H, const int* row,
                                                      const int* col, const float* rhohat, const float* __restrict__ w_r,
                                                      void* pre, int pre_dt, int E, int act, void* dact_out) {
    const int lane = threadIdx.x & 63;
    int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float* a = AB + (size_t)row[e] * ldab;
    const float* b = AB + (size_t)col[e] * ldab + H;
    const float rh = rhohat[e];
    for (int c = lane * 4; c < H; c += 256) {
        float4 va = *(const float4*)(a + c), vb = *(const float4*)(b + c), w = *(const float4*)(w_r + c);
        const float4 z = make_float4(va.x + vb.x + rh * w.x, va.y + vb.y + rh * w.y, va.z + vb.z + rh * w.z,
                                     va.w + vb.w + rh * w.w);
        st4_any(pre, pre_dt, (size_t)e * H + c,
                make_float4(apply_act(z.x, act), apply_act(z.y, act), apply_act(z.z, act), apply_act(z.w, act)));
        if (dact_out)
            st4_any(dact_out, pre_dt, (size_t)e * H + c,
                    make_float4(apply_dact(z.x, act), apply_dact(z.y, act), apply_dact(z.z, act), apply_dact(z.w, act)));
    }
}

extern "C" int fabind_gcl_pre(const float* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                              const float* w_r, void* pre, int pre_dt, int E, int act, void* dact_out, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && ldab % 4 == 0, "fabind_gcl_pre: H and ldab must be multiples of 4");
    if (E <= 0) return 0;
    hipLaunchKernelGGL(gcl_pre_kernel, dim3((E + 3) / 4), dim3(256), 0, stream, AB, ldab, H, row, col, rhohat, w_r, pre,
                       pre_dt, E, act, dact_out);
    FB_CHECK_LAUNCH();
    return 0;
}

// out[r,:] = sum_{e in [rowptr[r], rowptr[r+1])} act(Z[eidx ? eidx[e] : e,:])
// One 256-thread work-group per row: its 4 waves take edges round-robin, each lane owns 8 consecutive columns
// per 512-column slab (16-byte loads for bf16); partials are combined through LDS in fixed wave order.
struct F8 { float v[8]; };
__device__ __forceinline__ F8 ld8_any(const void* p, int dt, size_t i) {
    F8 r;
    if (dt == FB_DT_F32) {
        const float4 a = *(const float4*)((const float*)p + i), b = *(const float4*)((const float*)p + i + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const bf16_t*)p + i);
        r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
        r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
        r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
        r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    return r;
}

template <int NSLAB>
__global__ __launch_bounds__(256) void segment_sum_kernel(const void* __restrict__ Z, int z_dt, int ldz, int H,
                                                          const int* rowptr, const int* eidx, int act, float* out,
                                                          int ldo) {
    __shared__ F8 part[3][NSLAB * 64];
    const int r = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    F8 acc[NSLAB];
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[s].v[k] = 0.f;
    for (int ee = e0 + w; ee < e1; ee += 4) {
        const int e = eidx ? eidx[ee] : ee;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int c = s * 512 + lane * 8;
            if (c < H) {
                const F8 v = ld8_any(Z, z_dt, (size_t)e * ldz + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[s].v[k] += apply_act(v.v[k], act);
            }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) part[w - 1][s * 64 + lane] = acc[s];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int c = s * 512 + lane * 8;
            if (c < H) {
                F8 a = acc[s];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const F8 p = part[k][s * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a.v[q] += p.v[q];
                }
                *(float4*)(out + (size_t)r * ldo + c) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
                *(float4*)(out + (size_t)r * ldo + c + 4) = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
            }
        }
    }
}

extern "C" int fabind_segment_sum(const void* Z, int z_dt, int ldz, int H, const int* rowptr, const int* eidx,
                                  int n_rows, int act, float* out, int ldo, hipStream_t stream) {
    FB_REQUIRE(H % 8 == 0 && ldz % 8 == 0 && ldo % 4 == 0, "fabind_segment_sum: H/ldz must be multiples of 8, ldo of 4");
    FB_REQUIRE(H <= 1024, "fabind_segment_sum: H <= 1024");
    FB_REQUIRE(((uintptr_t)Z % 16 == 0) && ((uintptr_t)out % 16 == 0), "fabind_segment_sum: 16-byte alignment");
    if (n_rows <= 0) return 0;
    if (H <= 512)
        hipLaunchKernelGGL((segment_sum_kernel<1>), dim3(n_rows), dim3(256), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo);
    else
        hipLaunchKernelGGL((segment_sum_kernel<2>), dim3(n_rows), dim3(256), 0, stream, Z, z_dt, ldz, H, rowptr, eidx, act, out, ldo);
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
