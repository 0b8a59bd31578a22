// Forward of the fused intra-graph edge pipeline in SPLIT-bf16 ("bf16x3") arithmetic: the mode that meets the 1e-4 A parity gate at
// speed (algebra and reference lines: fused_edge.hip; reference arithmetic is fp32, FABind/fabind/models/egnn.py:68-128).
//
// Same pipeline and outputs as fused_edge_fwd2.hip -- gather, S1 = silu(A[row] + B[col] + rhohat w_r), M = silu(S1 W2^T + b2),
// agg[row] = sum M, s = w3 . silu(M Wc^T + bc) -- with fp32 inputs and fp32-grade contractions:
//   * the per-node projections AB arrive as fp32 rows (two 16-byte loads per 8-feature chunk and operand);
//   * an activation tile is TWO swizzled [64][H] bf16 LDS planes, hi = bf16(x) and lo = bf16(x - hi) (16 significand bits), written by the
//     epilogue that produces the tile; the packed weights come as hi | lo fragment arrays (host: kernels.pack_frag_split);
//   * a product term is three v_mfma_f32_16x16x32_bf16 (fe_gemm_x3, fused_common.h), fp32 accumulate;
//   * the segment sum adds hi + lo (the fp32 message to 2^-17).
// One work-group of H/64 waves per CU (2 x 64 KiB of LDS at H = 512); the contractions carry three times the MFMA work of the bf16
// kernel, which is what the other phases of a tile now hide behind.
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

typedef float f3_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f3_f2 f3_silu_pair(const f3_f2 z) {
    const f3_f2 t = z * -1.44269504f;
    const f3_f2 o = f3_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return z * f3_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
}

// value and derivative of silu from one sigmoid (the SAVE form's P1 epilogue)
__device__ __forceinline__ void f3_silu_pair_d(const f3_f2 z, f3_f2& m, f3_f2& d) {
    const f3_f2 t = z * -1.44269504f;
    const f3_f2 o = f3_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    const f3_f2 s = f3_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
    m = z * s;
    d = s + (m - m * s);
}
__device__ __forceinline__ uint32_t f3_pack(const f3_f2 v) { return pack2_bf16(v.x, v.y); }
typedef unsigned f3_u32x2_t __attribute__((ext_vector_type(2)));
// (non-temporal like fused_edge_fwd2.hip's: the saved tiles are read back once, a backward pass later)
__device__ __forceinline__ void f3_stq(bf16_t* p_, uint32_t a, uint32_t b) { __builtin_nontemporal_store(f3_u32x2_t{a, b}, (f3_u32x2_t*)p_); }

// SAVE (round 5, the training forward of the 'bf16x3' step): like the SAVE form of fused_edge_fwd2.hip, the kernel leaves what the
// two-contraction backward (fused_edge_bwd4.hip) reads -- the messages M as bf16 row-major (= the hi plane: bf16(M), the operand the
// bf16-grade weight gradient of this mode takes anyway), silu'(pre2) x keep and pre3 as bf16 quads in accumulator-fragment order (same
// lane <-> (edge, feature) map as fused_edge_fwd2.hip: one wave per 64 features, quad (i, j) of lane l at ((tile * NW + wave) * 16 + i * 4 + j) * 64 + l) --
// evaluated on the split-bf16 accumulators, i.e. closer to the fp32 values than the bf16 kernel's.
template <int H, bool DROP, bool SAVE>
__global__ __launch_bounds__(H, 2) void gcl_edge_fused_x3_kernel(const float* __restrict__ AB, int ldab, const int* __restrict__ row,
                                                              const int* __restrict__ col, const float* __restrict__ rhohat,
                                                              const float* __restrict__ w_r, const bf16_t* __restrict__ W2ph,
                                                              const bf16_t* __restrict__ W2pl, const float* __restrict__ b2,
                                                              const bf16_t* __restrict__ Wcph, const bf16_t* __restrict__ Wcpl,
                                                              const float* __restrict__ bc, const float* __restrict__ w3, int E,
                                                              float* agg, float* s_out, uint32_t thr16, float dscale, uint32_t seed,
                                                              float* bnd, int xcd_aware, bf16_t* __restrict__ Msave,
                                                              bf16_t* __restrict__ d2f, bf16_t* __restrict__ z3f) {
    constexpr int BM = FE_BM, MI = BM / 16;
    constexpr int NW = H / 64;
    constexpr int CH = H / 8;                                     // 8-feature chunks per edge row
    constexpr int EPW = 64 / CH;                                  // edges per wave and iteration of the row-wise phase (1 at H = 512)
    constexpr int EPI = NW * EPW;                                 // = 8
    constexpr int NIT = BM / EPI;                                 // = 8
    constexpr int PB = 4;                                         // iterations per load batch (16 x 16-byte loads in flight per lane)
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sXh = (bf16_t*)smem;                                  // [64][H] swizzled, hi plane: S1 -> M
    bf16_t* sXl = sXh + BM * H;                                   // lo plane
    int* sRow = (int*)(sXl + BM * H);
    int* sCol = sRow + BM;
    float* sRh = (float*)(sCol + BM);
    float* sDot = sRh + BM;                                       // [NW][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int sub = lane / CH, ch = lane % CH;
    const int n_tiles = (E + BM - 1) / BM, tpx = (n_tiles + 7) / 8;
    const int tile = xcd_aware ? (int)((blockIdx.x & 7) * tpx + (blockIdx.x >> 3)) : (int)blockIdx.x;   // (see gcl_edge_fused_kernel)
    if ((xcd_aware && (int)(blockIdx.x >> 3) >= tpx) || tile >= n_tiles) return;
    const int e0 = tile * BM;
    const int ne = min(BM, E - e0);

    if (tid < BM) {
        const bool ok = tid < ne;
        const unsigned ge = (unsigned)(e0 + (ok ? tid : 0));
        const int r = row[ge], c = col[ge];
        const float h = rhohat[ge];
        sRow[tid] = ok ? r : -1;
        sCol[tid] = ok ? c : 0;
        sRh[tid] = ok ? h : 0.f;
    }
    __syncthreads();

    // ---- P0 (row-wise): S1 = silu(A[row] + B[col] + rhohat w_r) -> hi | lo planes
    {
        const __amdgpu_buffer_rsrc_t rs_ab = fe_rsrc(const_cast<float*>(AB), 0xffffffffu);
        float wv[8];
        {
            const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
            wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
        }
#pragma unroll
        for (int b0 = 0; b0 < NIT; b0 += PB) {
            fe_u32x4_t ga[PB][2], gb[PB][2];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = (b0 + u) * EPI + wave * EPW + sub;
                const unsigned r = (unsigned)max(sRow[e], 0), c = (unsigned)sCol[e];
                const int oa = (int)((r * (unsigned)ldab + ch * 8) * 4), ob = (int)((c * (unsigned)ldab + H + ch * 8) * 4);
                ga[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, oa, 0, 0);
                ga[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, oa + 16, 0, 0);
                gb[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, ob, 0, 0);
                gb[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, ob + 16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = (b0 + u) * EPI + wave * EPW + sub;
                const float rh = sRh[e];
                uint32_t ph[4], pl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = k >> 1, w = (k & 1) * 2;
                    const f3_f2 a = f3_f2{__uint_as_float(ga[u][q][w]), __uint_as_float(ga[u][q][w + 1])};
                    const f3_f2 b = f3_f2{__uint_as_float(gb[u][q][w]), __uint_as_float(gb[u][q][w + 1])};
                    const f3_f2 m = f3_silu_pair(a + b + f3_f2{wv[2 * k], wv[2 * k + 1]} * rh);
                    fe_split2(m.x, m.y, ph[k], pl[k]);
                }
                const bool ok = e < ne;
                const int o = e * H + ((ch ^ (e & SWZ)) * 8);
                *(uint4*)&sXh[o] = ok ? make_uint4(ph[0], ph[1], ph[2], ph[3]) : make_uint4(0u, 0u, 0u, 0u);
                *(uint4*)&sXl[o] = ok ? make_uint4(pl[0], pl[1], pl[2], pl[3]) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    __syncthreads();

    // LDS element offset of this lane's accumulator quad (i, j): edge i*16 + fr, features wave*64 + j*16 + cq*4 .. +3
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
#define F3_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    const int fcol = wave * 64 + cq * 4;
    const size_t fbase = (((size_t)tile * NW + wave) * (MI * 4) * 64 + lane) * 4;
#define F3_FOFF(i, j) (fbase + (size_t)((i) * 4 + (j)) * 256)

    // ---- P1: M = silu(S1 W2^T + b2) -> planes in place
    f32x4_t acc[MI][4];
    fe_zero(acc);
    fe_gemm_x3<H, MI, SWZ>(sXh, sXl, W2ph, W2pl, wave, lane, acc);
    float4 bq4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bq4[j] = *(const float4*)(b2 + fcol + j * 16);
    __syncthreads();                                              // every wave has finished reading S1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f3_f2 b01 = f3_f2{bq4[j].x, bq4[j].y}, b23 = f3_f2{bq4[j].z, bq4[j].w};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f3_f2 m01, m23, d01, d23;
            if constexpr (SAVE) {
                f3_silu_pair_d(f3_f2{acc[i][j][0], acc[i][j][1]} + b01, m01, d01);
                f3_silu_pair_d(f3_f2{acc[i][j][2], acc[i][j][3]} + b23, m23, d23);
            } else {
                m01 = f3_silu_pair(f3_f2{acc[i][j][0], acc[i][j][1]} + b01);
                m23 = f3_silu_pair(f3_f2{acc[i][j][2], acc[i][j][3]} + b23);
            }
            if constexpr (DROP) {
                const uint32_t ee = (uint32_t)(e0 + i * 16 + fr), cc = (uint32_t)(fcol + j * 16);
                const f3_f2 k01 = f3_f2{fe_keep(seed, ee, cc, H, thr16, dscale), fe_keep(seed, ee, cc + 1, H, thr16, dscale)};
                const f3_f2 k23 = f3_f2{fe_keep(seed, ee, cc + 2, H, thr16, dscale), fe_keep(seed, ee, cc + 3, H, thr16, dscale)};
                m01 *= k01; m23 *= k23;
                if constexpr (SAVE) { d01 *= k01; d23 *= k23; }
            }
            if constexpr (SAVE) f3_stq(&d2f[F3_FOFF(i, j)], f3_pack(d01), f3_pack(d23));
            uint32_t h01, l01, h23, l23;
            fe_split2(m01.x, m01.y, h01, l01);
            fe_split2(m23.x, m23.y, h23, l23);
            *(uint2*)&sXh[F3_QOFF(i, j)] = make_uint2(h01, h23);
            *(uint2*)&sXl[F3_QOFF(i, j)] = make_uint2(l01, l23);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if constexpr (SAVE) {                                         // the hi plane = bf16(M), row-major: the weight gradient's operand
        const __amdgpu_buffer_rsrc_t rs_m = fe_rsrc(Msave + (size_t)e0 * H, (unsigned)ne * H * 2);
#pragma unroll
        for (int q = tid; q < BM * CH; q += H) {
            const int rw = q / CH, c8 = q % CH;
            const uint4 v = *(const uint4*)&sXh[rw * H + ((c8 ^ (rw & SWZ)) * 8)];
            __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rs_m, (int)((rw * H + c8 * 8) * 2), 0, /*aux: nt*/ 2);
        }
    }

    // ---- P2: agg[row] = sum of M (= hi + lo) over the node's edges (deterministic: fused_common.h fe_scan_runs64)
    {
        const bool head_cont = e0 > 0 && row[e0 - 1] == sRow[0];
        const bool tail_cont = e0 + ne < E && row[e0 + ne] == sRow[ne - 1];
        (void)fe_scan_runs64<H, false, SWZ>(sXh, sRow, nullptr, ne, head_cont, tail_cont, agg, (unsigned)H, bnd + (size_t)tile * 2 * H, tid,
                                            nullptr, 0u, sXl);
    }

    // ---- P3: s = w3 . silu(M Wc^T + bc)
    fe_zero(acc);
    fe_gemm_x3<H, MI, SWZ>(sXh, sXl, Wcph, Wcpl, wave, lane, acc);
    {
        f3_f2 da[MI], db[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) { da[i] = f3_f2{0.f, 0.f}; db[i] = da[i]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bq = *(const float4*)(bc + fcol + j * 16), wq = *(const float4*)(w3 + fcol + j * 16);
            const f3_f2 b01 = f3_f2{bq.x, bq.y}, b23 = f3_f2{bq.z, bq.w}, w01 = f3_f2{wq.x, wq.y}, w23 = f3_f2{wq.z, wq.w};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const f3_f2 z01 = f3_f2{acc[i][j][0], acc[i][j][1]} + b01, z23 = f3_f2{acc[i][j][2], acc[i][j][3]} + b23;
                da[i] += f3_silu_pair(z01) * w01;
                db[i] += f3_silu_pair(z23) * w23;
                if constexpr (SAVE) f3_stq(&z3f[F3_FOFF(i, j)], f3_pack(z01), f3_pack(z23));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float t = (da[i].x + da[i].y) + (db[i].x + db[i].y);  // this lane's 16 features of edge i*16 + fr
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);                           // the wave's 64 features
            if (cq == 0) sDot[wave * BM + i * 16 + fr] = t;
        }
    }
    __syncthreads();
    if (tid < ne) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sDot[w * BM + tid];
        s_out[e0 + tid] = s;
    }
#undef F3_QOFF
#undef F3_FOFF
}

int fe_fwd3_launch(const float* AB, int ldab, int H, const int* row, const int* col, const float* rhohat, const float* w_r,
                   const void* W2ph, const void* W2pl, const float* b2, const void* Wcph, const void* Wcpl, const float* bc,
                   const float* w3, int E, float* agg, float* s_out, uint32_t thr16, float dscale, unsigned seed, float* bnd,
                   int xcd_aware, hipStream_t stream, void* Msave, void* d2f, void* z3f) {
    const dim3 grid((((E + FE_BM - 1) / FE_BM + 7) / 8) * 8);
    const size_t lds = (size_t)2 * FE_BM * H * 2 + 3 * FE_BM * sizeof(int) + (size_t)(H / 64) * FE_BM * sizeof(float);
#define F3_LAUNCH_(HH, DD, SS)                                                                                     \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)gcl_edge_fused_x3_kernel<HH, DD, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((gcl_edge_fused_x3_kernel<HH, DD, SS>), grid, dim3(HH), lds, stream, AB, ldab, row, col, rhohat, w_r,  \
                           (const bf16_t*)W2ph, (const bf16_t*)W2pl, b2, (const bf16_t*)Wcph, (const bf16_t*)Wcpl, bc, w3, E, agg, \
                           s_out, thr16, dscale, (uint32_t)seed, bnd, xcd_aware, (bf16_t*)Msave, (bf16_t*)d2f, (bf16_t*)z3f); \
    } while (0)
#define F3_LAUNCH(HH)                                                                                              \
    do {                                                                                                           \
        if (Msave) { if (thr16) F3_LAUNCH_(HH, true, true); else F3_LAUNCH_(HH, false, true); }                    \
        else { if (thr16) F3_LAUNCH_(HH, true, false); else F3_LAUNCH_(HH, false, false); }                        \
    } while (0)
    if (H == 512) F3_LAUNCH(512); else if (H == 256) F3_LAUNCH(256); else if (H == 128) F3_LAUNCH(128); else F3_LAUNCH(64);
#undef F3_LAUNCH
#undef F3_LAUNCH_
    return 0;
}
