// Forward of the fused intra-graph edge pipeline (algebra and reference lines: fused_edge.hip), second form.  Same tile (64 edges,
// one [64][H] LDS tile rewritten in place, two work-groups per CU), same outputs; what changed follows the backward's
// fused_edge_bwd3.hip:
//   * P0 is ROW-WISE: a wave handles one edge per iteration, its lanes the edge's 16-byte chunks, all eight iterations' gathered
//     rows in flight before the first is used (full 1 KiB coalesced reads instead of 64 cache lines per load instruction);
//   * both contractions run in the OPERAND-SWAPPED form C^T = W x X^T: a lane's accumulator quad is four consecutive features of
//     one edge, so the message tile is written with one 8-byte LDS store per quad (not four 2-byte stores), biases / w3 are 16-byte
//     loads, and the per-edge scalar s = w3 . silu(...) needs two cross-lane steps per 16-edge block instead of four per feature block;
//   * the element-wise arithmetic is written on fp32 PAIRS (v_pk_add / mul / fma_f32);
//   * the LDS tile swizzle uses four row bits (conflict-free ds_read_b128 of the A fragments, fused_common.h);
//   * dropout is a compile-time variant.
// SAVE (round 4, the training forward of the bf16 step): the kernel also leaves what the adjoint of fused_edge_bwd4.hip would otherwise
// recompute with two more H x H contractions per edge -- the messages M as a row-major [E, H] bf16 operand (copied out of the LDS tile;
// the weight gradient d Wc = dT^T M reads it as it is), silu'(pre2) (times the dropout keep factor) and pre3 = M Wc^T + bc as bf16 in
// the ACCUMULATOR-FRAGMENT order of this kernel: tile t, wave w, quad (i, j), lane l -> 4 consecutive features at
// ((t * NW + w) * 16 + i * 4 + j) * 64 + l) * 4 -- every store instruction is 512 contiguous bytes, and the backward's waves have the
// same lane <-> (edge, feature) map, so they load their quads back the same way.  +3 x E x H x 2 bytes of stores per launch.
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

typedef float fw_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fw_f2 fw_silu_pair(const fw_f2 z) {
    const fw_f2 t = z * -1.44269504f;
    const fw_f2 o = fw_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return z * fw_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
}
__device__ __forceinline__ uint32_t fw_pack(const fw_f2 v) { return pack2_bf16(v.x, v.y); }
__device__ __forceinline__ fw_f2 fw_unpack(const uint32_t u) { return fw_f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

// (m, d) = (silu(z), silu'(z)) on a pair
__device__ __forceinline__ void fw_silu_pair_d(const fw_f2 z, fw_f2& m, fw_f2& d) {
    const fw_f2 t = z * -1.44269504f;
    const fw_f2 o = fw_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    const fw_f2 s = fw_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
    m = z * s;
    d = s + (m - m * s);
}

#ifndef FW2_NT_SAVE
#define FW2_NT_SAVE 1      // 1 = the fragment-ordered saved tiles (silu'(pre2), pre3) are stored non-temporally (they are read back once, a backward pass later)
#endif
typedef unsigned fw_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fw_stq(bf16_t* p_, uint32_t a, uint32_t b) {
#if FW2_NT_SAVE
    __builtin_nontemporal_store(fw_u32x2_t{a, b}, (fw_u32x2_t*)p_);
#else
    *(uint2*)p_ = make_uint2(a, b);
#endif
}

template <int H, bool DROP, bool SAVE>
__global__ __launch_bounds__(H, 4) void gcl_edge_fused2_kernel(const bf16_t* __restrict__ AB, int ldab, const int* __restrict__ row,
                                                            const int* __restrict__ col, const float* __restrict__ rhohat,
                                                            const float* __restrict__ w_r, const bf16_t* __restrict__ W2p,
                                                            const float* __restrict__ b2, const bf16_t* __restrict__ Wcp,
                                                            const float* __restrict__ bc, const float* __restrict__ w3, int E,
                                                            float* agg, float* s_out, uint32_t thr16, float dscale,
                                                            uint32_t seed, float* bnd, bf16_t* agg16, int xcd_aware,
                                                            bf16_t* __restrict__ Msave, bf16_t* __restrict__ d2f, bf16_t* __restrict__ z3f) {
    constexpr int BM = FE_BM, MI = BM / 16;
    constexpr int NW = H / 64;
    constexpr int CH = H / 8;                                     // 16-byte chunks per edge row
    constexpr int EPW = 64 / CH;                                  // edges per wave and iteration of the row-wise phase (1 at H = 512)
    constexpr int EPI = NW * EPW;                                 // = 8
    constexpr int NIT = BM / EPI;                                 // = 8
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [64][H] swizzled: S1 -> M
    int* sRow = (int*)(sX + BM * H);
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

    // ---- P0 (row-wise): S1 = silu(A[row] + B[col] + rhohat w_r) -> tile
    {
        const __amdgpu_buffer_rsrc_t rs_ab = fe_rsrc(const_cast<bf16_t*>(AB), 0xffffffffu);
        fw_f2 wv2[4];
        {
            const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
            wv2[0] = fw_f2{w0.x, w0.y}; wv2[1] = fw_f2{w0.z, w0.w}; wv2[2] = fw_f2{w1.x, w1.y}; wv2[3] = fw_f2{w1.z, w1.w};
        }
        fe_u32x4_t ga[NIT], gb[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = u * EPI + wave * EPW + sub;
            const unsigned r = (unsigned)max(sRow[e], 0), c = (unsigned)sCol[e];
            ga[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, (int)((r * (unsigned)ldab + ch * 8) * 2), 0, 0);
            gb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_ab, (int)((c * (unsigned)ldab + H + ch * 8) * 2), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = u * EPI + wave * EPW + sub;
            const float rh = sRh[e];
            uint32_t po[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) po[k] = fw_pack(fw_silu_pair(fw_unpack(ga[u][k]) + fw_unpack(gb[u][k]) + wv2[k] * rh));
            *(uint4*)&sX[e * H + ((ch ^ (e & SWZ)) * 8)] = e < ne ? make_uint4(po[0], po[1], po[2], po[3]) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();

    // LDS element offset of this lane's accumulator quad (i, j): edge i*16 + fr, features wave*64 + j*16 + cq*4 .. +3
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
#define FW_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    // element offset of the same quad in the fragment-ordered save arrays (SAVE)
    const size_t fbase = (((size_t)tile * NW + wave) * (MI * 4) * 64 + lane) * 4;
#define FW_FOFF(i, j) (fbase + (size_t)((i) * 4 + (j)) * 256)
    const int fcol = wave * 64 + cq * 4;

    // ---- P1: M = silu(S1 W2^T + b2) -> tile in place
    f32x4_t acc[MI][4];
    fe_zero(acc);
    fe_gemm_rolled<H, MI, true, SWZ>(sX, W2p, wave, lane, acc);
    float4 bq4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bq4[j] = *(const float4*)(b2 + fcol + j * 16);
    __syncthreads();                                              // every wave has finished reading S1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const fw_f2 b01 = fw_f2{bq4[j].x, bq4[j].y}, b23 = fw_f2{bq4[j].z, bq4[j].w};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            fw_f2 m01, m23, d01, d23;
            if constexpr (SAVE) {
                fw_silu_pair_d(fw_f2{acc[i][j][0], acc[i][j][1]} + b01, m01, d01);
                fw_silu_pair_d(fw_f2{acc[i][j][2], acc[i][j][3]} + b23, m23, d23);
            } else {
                m01 = fw_silu_pair(fw_f2{acc[i][j][0], acc[i][j][1]} + b01);
                m23 = fw_silu_pair(fw_f2{acc[i][j][2], acc[i][j][3]} + b23);
            }
            if constexpr (DROP) {
                const uint32_t ee = (uint32_t)(e0 + i * 16 + fr), cc = (uint32_t)(fcol + j * 16);
                const fw_f2 k01 = fw_f2{fe_keep(seed, ee, cc, H, thr16, dscale), fe_keep(seed, ee, cc + 1, H, thr16, dscale)};
                const fw_f2 k23 = fw_f2{fe_keep(seed, ee, cc + 2, H, thr16, dscale), fe_keep(seed, ee, cc + 3, H, thr16, dscale)};
                m01 *= k01; m23 *= k23;
                if constexpr (SAVE) { d01 *= k01; d23 *= k23; }
            }
            *(uint2*)&sX[FW_QOFF(i, j)] = make_uint2(fw_pack(m01), fw_pack(m23));
            if constexpr (SAVE) fw_stq(&d2f[FW_FOFF(i, j)], fw_pack(d01), fw_pack(d23));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if constexpr (SAVE) {                                         // the message tile as the weight gradient's row-major operand
        const __amdgpu_buffer_rsrc_t rs_m = fe_rsrc(Msave + (size_t)e0 * H, (unsigned)ne * H * 2);
#pragma unroll
        for (int q = tid; q < BM * CH; q += H) {
            const int rw = q / CH, c8 = q % CH;
            const uint4 v = *(const uint4*)&sX[rw * H + ((c8 ^ (rw & SWZ)) * 8)];
            __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rs_m, (int)((rw * H + c8 * 8) * 2), 0, /*aux: nt*/ 2);
        }
    }

    // ---- P2: agg[row] = sum of M over the node's edges (deterministic: fused_common.h fe_scan_rows)
    {
        const bool head_cont = e0 > 0 && row[e0 - 1] == sRow[0];
        const bool tail_cont = e0 + ne < E && row[e0 + ne] == sRow[ne - 1];
        (void)fe_scan_runs64<H, false, SWZ>(sX, sRow, nullptr, ne, head_cont, tail_cont, agg, (unsigned)H, bnd + (size_t)tile * 2 * H, tid, agg16, (unsigned)H);
    }

    // ---- P3: s = w3 . silu(M Wc^T + bc)
    fe_zero(acc);
    fe_gemm_rolled<H, MI, true, SWZ>(sX, Wcp, wave, lane, acc);
    {
        fw_f2 da[MI], db[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) { da[i] = fw_f2{0.f, 0.f}; db[i] = da[i]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bq = *(const float4*)(bc + fcol + j * 16), wq = *(const float4*)(w3 + fcol + j * 16);
            const fw_f2 b01 = fw_f2{bq.x, bq.y}, b23 = fw_f2{bq.z, bq.w}, w01 = fw_f2{wq.x, wq.y}, w23 = fw_f2{wq.z, wq.w};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const fw_f2 z01 = fw_f2{acc[i][j][0], acc[i][j][1]} + b01, z23 = fw_f2{acc[i][j][2], acc[i][j][3]} + b23;
                da[i] += fw_silu_pair(z01) * w01;
                db[i] += fw_silu_pair(z23) * w23;
                if constexpr (SAVE) fw_stq(&z3f[FW_FOFF(i, j)], fw_pack(z01), fw_pack(z23));
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
#undef FW_QOFF
#undef FW_FOFF
}

// launched from fabind_gcl_edge_fused (fused_edge.hip) when the forward variant is 1
int fe_fwd2_launch(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat, const float* w_r,
                   const void* W2p, const float* b2, const void* Wcp, const float* bc, const float* w3, int E, float* agg,
                   float* s_out, uint32_t thr16, float dscale, unsigned seed, float* bnd, void* agg16, int xcd_aware, hipStream_t stream,
                   void* Msave, void* d2f, void* z3f) {
    const dim3 grid((((E + FE_BM - 1) / FE_BM + 7) / 8) * 8);
    const size_t lds = (size_t)FE_BM * H * 2 + 3 * FE_BM * sizeof(int) + (size_t)(H / 64) * FE_BM * sizeof(float);
#define FW_LAUNCH_(HH, DD, SS)                                                                                       \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)gcl_edge_fused2_kernel<HH, DD, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((gcl_edge_fused2_kernel<HH, DD, SS>), grid, dim3(HH), lds, stream, (const bf16_t*)AB, ldab, row, col, rhohat, \
                           w_r, (const bf16_t*)W2p, b2, (const bf16_t*)Wcp, bc, w3, E, agg, s_out, thr16, dscale, \
                           (uint32_t)seed, bnd, (bf16_t*)agg16, xcd_aware, (bf16_t*)Msave, (bf16_t*)d2f, (bf16_t*)z3f); \
    } while (0)
#define FW_LAUNCH(HH)                                                                  \
    do {                                                                               \
        if (Msave) { if (thr16) FW_LAUNCH_(HH, true, true); else FW_LAUNCH_(HH, false, true); } \
        else { if (thr16) FW_LAUNCH_(HH, true, false); else FW_LAUNCH_(HH, false, false); }    \
    } while (0)
    if (H == 512) FW_LAUNCH(512); else if (H == 256) FW_LAUNCH(256); else if (H == 128) FW_LAUNCH(128); else FW_LAUNCH(64);
#undef FW_LAUNCH
#undef FW_LAUNCH_
    return 0;
}
