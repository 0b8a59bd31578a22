// NT GEMM on MFMA for gfx950:  C[M,N] = epilogue( prologue(A)[M,K] * W[N,K]^T )
//
// One kernel family serves every dense contraction of the FABind hot path (node / edge / pair linears
// and their backward passes, reference call sites: egnn.py:40-60,160-170, model_utils.py:83-92,168-169).
//   * A is fp32 or bf16 in HBM, optionally the K-concatenation of two tensors (replaces torch.cat of
//     [h, agg] in MC_E_GCL.node_model, egnn.py:102) with an activation applied while it is staged
//     (the consumer applies SiLU/ReLU so that the producer can keep the pre-activation for backward).
//   * W is pre-packed to the MMA type: bf16 -> v_mfma_f32_16x16x32_bf16, fp32 -> v_mfma_f32_16x16x4_f32
//     (exact fp32 accumulate chain; used for the 1e-4 A parity mode).
//   * epilogue: alpha, bias, activation, derivative-of-activation multiply (backward), residual
//     (optionally row-gathered), accumulate, row-dot with a vector (coord_mlp last layer, H->1).
//   * grouped (ragged batched) mode: blockIdx.z selects a descriptor {a_row0, M, w_row0, N, c_off, ldc}.
//
// Tile: 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 tiles.
#include "common.h"
#include "fabind_hip.h"

#define BM 128
#define BN 128
#define BK 32

template <typename T> struct LdsStride;
template <> struct LdsStride<float> { static constexpr int v = BK + 4; };   // 36 dwords: conflict-free b128
template <> struct LdsStride<bf16_t> { static constexpr int v = BK + 8; };  // 40 halfs = 20 dwords

struct Vec8 { float v[8]; };

template <typename T> __device__ __forceinline__ Vec8 load8(const T* p);
template <> __device__ __forceinline__ Vec8 load8<float>(const float* p) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    Vec8 r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <> __device__ __forceinline__ Vec8 load8<bf16_t>(const bf16_t* p) {
    uint4 u = *(const uint4*)p;
    Vec8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
template <typename T> __device__ __forceinline__ void store8_lds(T* p, const Vec8& r);
template <> __device__ __forceinline__ void store8_lds<float>(float* p, const Vec8& r) {
    *(float4*)p = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    *(float4*)(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
template <> __device__ __forceinline__ void store8_lds<bf16_t>(bf16_t* p, const Vec8& r) {
    uint4 u;
    u.x = pack2_bf16(r.v[0], r.v[1]);
    u.y = pack2_bf16(r.v[2], r.v[3]);
    u.z = pack2_bf16(r.v[4], r.v[5]);
    u.w = pack2_bf16(r.v[6], r.v[7]);
    *(uint4*)p = u;
}

// one K=32 step of a 16x16 tile: a/b point at 8 consecutive K elements of this lane's row/col
template <typename TM> __device__ __forceinline__ void mma_k32(f32x4_t& acc, const TM* a, const TM* b);
template <> __device__ __forceinline__ void mma_k32<bf16_t>(f32x4_t& acc, const bf16_t* a, const bf16_t* b) {
    bf16x8_t av = *(const bf16x8_t*)a, bv = *(const bf16x8_t*)b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_k32<float>(f32x4_t& acc, const float* a, const float* b) {
    float4 a0 = *(const float4*)a, a1 = *(const float4*)(a + 4);
    float4 b0 = *(const float4*)b, b1 = *(const float4*)(b + 4);
    // the K index inside the 32-chunk is permuted identically for A and B (k = 8*(lane>>4)+j), which
    // leaves the dot product unchanged and lets each lane read 8 contiguous floats.
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
}

// ---- shared epilogue: C/D layout of mfma 16x16: col = lane&15, row = (lane>>4)*4 + reg.
// BM_ = rows of the block tile (64 per wave-row).  sDot: float[2][BM_].  sOut (optional): bf16 staging
// tile [BM_][BN+8] in LDS so that bf16 outputs leave as whole 16-B chunks of a row instead of 2-B scalars.
#define OUT_LD (BN + 8)
template <int BM_, bool RAW_BARRIER = false>
__device__ __forceinline__ void gemm_epilogue(const FabindGemmArgs& p, f32x4_t (&acc)[4][4], float* sDot, bf16_t* sOut,
                                              int M, int N, int ldc, long a_row0, long w_row0, long c_off, int m0,
                                              int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15;
    const int cq = lane >> 4;
    const bool want_dot = p.dotvec != nullptr;
    const bool staged = sOut != nullptr && p.C != nullptr && p.c_dtype == FB_DT_BF16 && !p.accumulate;
    const uint32_t drop_thr = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float drop_scale = 1.0f / (1.0f - (float)drop_thr / 65536.0f);
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i) {
        float ds0 = 0.f, ds1 = 0.f, ds2 = 0.f, ds3 = 0.f;
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j) {
            const f32x4_t a4 = acc[i][j];
            const float av[4] = {a4[0], a4[1], a4[2], a4[3]};
            const int coll = wn * 64 + j * 16 + fr;
            const int col = n0 + coll;
            const float dv = (want_dot && col < N) ? p.dotvec[col] : 0.f;
            const float bv = (p.bias && col < N) ? p.bias[w_row0 + col] : 0.f;
#pragma clang loop unroll(full)
            for (int r = 0; r < 4; ++r) {
                const int rowl = wm * 64 + i * 16 + cq * 4 + r;
                const int row = m0 + rowl;
                float v = 0.f;
                if (row < M && col < N) {
                    v = av[r] * p.alpha + bv;
                    const float vpre = v;
                    v = apply_act(v, p.act_epi);
                    if (p.aux) v *= apply_dact(ld_any(p.aux, p.aux_dtype, (size_t)(a_row0 + row) * p.ldaux + col), p.dact_epi);
                    if (drop_thr) v *= ((fb_hash32(p.drop_seed + (uint32_t)row * (uint32_t)N + (uint32_t)col) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                    if (p.R) {
                        long rr = p.r_index ? (long)p.r_index[a_row0 + row] : (a_row0 + row);
                        v += ld_any(p.R, p.r_dtype, (size_t)rr * p.ldr + col);
                    }
                    if (p.C2) st_any(p.C2, p.c2_bf16 ? FB_DT_BF16 : p.c_dtype, (size_t)c_off + (size_t)row * ldc + col, apply_dact(vpre, p.act_epi));
                    if (staged) {
                        sOut[rowl * OUT_LD + coll] = f32_to_bf16(p.store_preact ? vpre : v);
                    } else if (p.C) {
                        size_t ci = (size_t)c_off + (size_t)row * ldc + col;
                        if (p.accumulate) v += ld_any(p.C, p.c_dtype, ci);
                        st_any(p.C, p.c_dtype, ci, p.store_preact ? vpre : v);
                        if (p.C16) ((bf16_t*)p.C16)[(size_t)row * p.ldc16 + col] = f32_to_bf16(v);
                    }
                }
                const float t = v * dv;
                if (r == 0) ds0 += t; else if (r == 1) ds1 += t; else if (r == 2) ds2 += t; else ds3 += t;
            }
        }
        if (want_dot) {
            float dsr[4] = {ds0, ds1, ds2, ds3};
#pragma clang loop unroll(full)
            for (int r = 0; r < 4; ++r) {
                float dsum = dsr[r];
                dsum += __shfl_xor(dsum, 1, 64);
                dsum += __shfl_xor(dsum, 2, 64);
                dsum += __shfl_xor(dsum, 4, 64);
                dsum += __shfl_xor(dsum, 8, 64);
                if (fr == 0) sDot[wn * BM_ + wm * 64 + i * 16 + cq * 4 + r] = dsum;
            }
        }
    }
    if (want_dot || staged) {
        if (RAW_BARRIER) {   // do not drain in-flight LDS-DMA (a plain __syncthreads would wait vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
    }
    if (want_dot) {
        for (int t = tid; t < BM_; t += blockDim.x)
            if (m0 + t < M) p.dot_out[(size_t)(a_row0 + m0 + t) * p.dot_ld + n0 / BN] = sDot[t] + sDot[BM_ + t];
        if (RAW_BARRIER) {   // sDot is rewritten by the next tile's epilogue
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (staged) {
        bf16_t* C = (bf16_t*)p.C + c_off;
        const bool vec_ok = (ldc % 8 == 0) && (((uintptr_t)C & 15) == 0);
        for (int idx = tid; idx < BM_ * (BN / 8); idx += blockDim.x) {
            const int rowl = idx / (BN / 8), ch = idx % (BN / 8);
            const int row = m0 + rowl, col = n0 + ch * 8;
            if (row >= M || col >= N) continue;
            const bf16_t* sp = &sOut[rowl * OUT_LD + ch * 8];
            if (vec_ok && col + 8 <= N) {
                *(uint4*)(C + (size_t)row * ldc + col) = *(const uint4*)sp;
            } else {
                for (int e = 0; e < 8 && col + e < N; ++e) C[(size_t)row * ldc + col + e] = sp[e];
            }
        }
    }
}

// ---- specialised epilogue for the hot configurations (alpha = 1; no aux / residual / accumulate / gather):
// compile-time activation and outputs, bf16 tile staged through LDS and stored as 16-B chunks.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
template <int ACT> __device__ __forceinline__ float fast_act(float x) {
    if (ACT == FB_ACT_SILU) return x * fast_sigmoid(x);
    if (ACT == FB_ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}
template <int ACT> __device__ __forceinline__ float fast_dact(float x) {
    if (ACT == FB_ACT_SILU) { const float sg = fast_sigmoid(x); return sg * (1.0f + x * (1.0f - sg)); }
    if (ACT == FB_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    return 1.f;
}

template <int BM_, int ACT, bool HAS_C, bool HAS_C2, bool HAS_DOT, bool RAW_BARRIER, bool STORE_PRE = false, bool HAS_RG = false,
          bool FOLD = false,   // FOLD: LayerNorm of the A rows folded in (row_mu / row_rs / col_c); its own dispatch codes, so
                               // every other GEMM keeps the un-folded instruction stream
          bool AUXMUL = false, // AUXMUL: C = acc * act'(aux) with aux a bf16 [M, N] tile read row-contiguously while the staged
                               // tile is flushed (activation adjoint of an MLP fused into the input-gradient GEMM of its
                               // second Linear); dact_epi = RELU (aux = the activation's output) or STORED_DERIV (aux = act')
          bool RESADD = false> // RESADD: C = (acc + bias) + R with R a bf16 [M, N] residual read row-contiguously during the flush
__device__ __forceinline__ void gemm_epilogue_fast(const FabindGemmArgs& p, f32x4_t (&acc)[4][4], float* sDot, bf16_t* sOut,
                                                   int M, int N, int ldc, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, cq = lane >> 4;
    bf16_t* C = (bf16_t*)p.C;
    bf16_t* C2 = (bf16_t*)p.C2;
    auto barrier = [&]() {
        if (RAW_BARRIER) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else __syncthreads();
    };
    auto flush = [&](bf16_t* dst) {       // LDS tile -> global, 16-B chunks of a row
        const bool vec_ok = (ldc % 8 == 0) && (((uintptr_t)dst & 15) == 0);
        for (int idx = tid; idx < BM_ * (BN / 8); idx += blockDim.x) {
            const int rowl = idx / (BN / 8), ch = idx % (BN / 8);
            const int row = m0 + rowl, col = n0 + ch * 8;
            if (row >= M || col >= N) continue;
            const bf16_t* sp = &sOut[rowl * OUT_LD + ch * 8];
            if constexpr (AUXMUL) {
                const bf16_t* ap = (const bf16_t*)p.aux + (size_t)row * p.ldaux + col;
                const bool relu = p.dact_epi == FB_ACT_RELU;
                const float al = p.alpha;      // (the 1 / (1 - p) of an epilogue dropout whose kept positions are aux's non-zeros)
                if (vec_ok && col + 8 <= N && (p.ldaux % 8 == 0) && (((uintptr_t)p.aux & 15) == 0)) {
                    const uint4 v = *(const uint4*)sp, a = *(const uint4*)ap;
                    const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, aa[4] = {a.x, a.y, a.z, a.w};
                    uint32_t oo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = __uint_as_float(aa[e] << 16), a1 = __uint_as_float(aa[e] & 0xffff0000u);
                        const float f0 = relu ? (a0 > 0.f ? al : 0.f) : a0 * al, f1 = relu ? (a1 > 0.f ? al : 0.f) : a1 * al;
                        oo[e] = pack2_bf16(__uint_as_float(vv[e] << 16) * f0, __uint_as_float(vv[e] & 0xffff0000u) * f1);
                    }
                    *(uint4*)(dst + (size_t)row * ldc + col) = make_uint4(oo[0], oo[1], oo[2], oo[3]);
                } else {
                    for (int e = 0; e < 8 && col + e < N; ++e) {
                        const float a0 = bf16_to_f32(ap[e]);
                        dst[(size_t)row * ldc + col + e] = f32_to_bf16(bf16_to_f32(sp[e]) * (relu ? (a0 > 0.f ? al : 0.f) : a0 * al));
                    }
                }
                continue;
            }
            if constexpr (RESADD) {
                const bf16_t* rp = (const bf16_t*)p.R + (size_t)row * p.ldr + col;
                if (vec_ok && col + 8 <= N && (p.ldr % 8 == 0) && (((uintptr_t)p.R & 15) == 0)) {
                    const uint4 v = *(const uint4*)sp, a = *(const uint4*)rp;
                    const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, aa[4] = {a.x, a.y, a.z, a.w};
                    uint32_t oo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        oo[e] = pack2_bf16(__uint_as_float(vv[e] << 16) + __uint_as_float(aa[e] << 16),
                                           __uint_as_float(vv[e] & 0xffff0000u) + __uint_as_float(aa[e] & 0xffff0000u));
                    *(uint4*)(dst + (size_t)row * ldc + col) = make_uint4(oo[0], oo[1], oo[2], oo[3]);
                } else {
                    for (int e = 0; e < 8 && col + e < N; ++e)
                        dst[(size_t)row * ldc + col + e] = f32_to_bf16(bf16_to_f32(sp[e]) + bf16_to_f32(rp[e]));
                }
                continue;
            }
            if (vec_ok && col + 8 <= N) *(uint4*)(dst + (size_t)row * ldc + col) = *(const uint4*)sp;
            else for (int e = 0; e < 8 && col + e < N; ++e) dst[(size_t)row * ldc + col + e] = sp[e];
        }
    };
    float bv[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + fr;
        bv[j] = (p.bias && col < N) ? p.bias[col] : 0.f;
        dv[j] = (HAS_DOT && col < N) ? p.dotvec[col] : 0.f;
    }
    const uint32_t drop_thr = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float drop_scale = 1.0f / (1.0f - (float)drop_thr / 65536.0f);
    float cv[4] = {0.f, 0.f, 0.f, 0.f};
    if (FOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fr;
            cv[j] = col < N ? p.col_c[col] : 0.f;
        }
    }
    // (Round 4 tried value AND derivative of the SiLU from ONE sigmoid per element, the two half-tiles staged side by side and flushed
    //  together: 429 us per [98,688 x 1024] -> 512 launch against 268 us for this form -- the fully unrolled two-output body does not fit
    //  the 128-VGPR budget of the two-work-groups-per-CU build and spills; reverted, tools/probes/gemm_node_epi.py.)
    if (HAS_C2) {                          // derivative tile first (uses the staging buffer), then the value tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sOut[(wm * 64 + i * 16 + cq * 4 + r) * OUT_LD + wn * 64 + j * 16 + fr] = f32_to_bf16(fast_dact<ACT>(acc[i][j][r] + bv[j]));
        barrier();
        flush(C2);
        barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ds[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vp = acc[i][j][r];
                if (FOLD) {            // compile-time: LayerNorm of the A rows folded into the epilogue
                    const int rl = min(m0 + wm * 64 + i * 16 + cq * 4 + r, M - 1);
                    vp = p.row_rs[rl] * (vp - p.row_mu[rl] * cv[j]);
                }
                vp += bv[j];
                if (HAS_RG) {          // + residual row gathered through r_index (adjoint of a segment-sum consumer)
                    const int rowg = m0 + wm * 64 + i * 16 + cq * 4 + r, colg = n0 + wn * 64 + j * 16 + fr;
                    if (rowg < M && colg < N) vp += ((const float*)p.R)[(size_t)p.r_index[rowg] * p.ldr + colg];
                }
                float v = fast_act<ACT>(vp);
                if (drop_thr) {            // uniform branch: epilogue dropout (same counter-based mask as the generic epilogue)
                    const uint32_t rowg = m0 + wm * 64 + i * 16 + cq * 4 + r, colg = n0 + wn * 64 + j * 16 + fr;
                    v *= ((fb_hash32(p.drop_seed + rowg * (uint32_t)N + colg) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                }
                if (HAS_C) sOut[(wm * 64 + i * 16 + cq * 4 + r) * OUT_LD + wn * 64 + j * 16 + fr] = f32_to_bf16(STORE_PRE ? vp : v);
                if (HAS_DOT) ds[r] += v * dv[j];
            }
        if (HAS_DOT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = ds[r];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                if (fr == 0) sDot[wn * BM_ + wm * 64 + i * 16 + cq * 4 + r] = t;
            }
        }
    }
    barrier();
    if (HAS_DOT)
        for (int t = tid; t < BM_; t += blockDim.x)
            if (m0 + t < M) p.dot_out[(size_t)(m0 + t) * p.dot_ld + n0 / BN] = sDot[t] + sDot[BM_ + t];
    if (HAS_C) flush(C);
    if (RAW_BARRIER) barrier();            // the staging tile / sDot are reused by the next tile
}

// fp32 output (node-level residual stream): bias (+ residual).  The C fragments hold 16 consecutive columns per 16 lanes
// (64-B pieces of 4 different rows per store); each wave therefore transposes its 64 x 64 sub-tile through a PRIVATE
// LDS slab in two 32-row halves and writes 256-B row segments with 16-B lanes (residual rows are read the same way).
// The slabs live in the (finished) pipeline stages; no work-group barrier is needed, LDS ops of one wave are in order.
#define F32_SLAB_LD 68
template <int BM_, bool HAS_R>
__device__ __forceinline__ void gemm_epilogue_f32(const FabindGemmArgs& p, f32x4_t (&acc)[4][4], float* sStage, int stage_bytes,
                                                  int M, int N, int ldc, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, cq = lane >> 4;
    float* C = (float*)p.C;
    const float* R = (const float*)p.R;
    // epilogue dropout (round 5: train-mode dropout AHEAD of the residual, C = R + drop(A W^T + b)): the counter-based mask of the generic
    // epilogue, keyed by (seed, row, col) -- the adjoint regenerates it (fabind_mul_dropmask_colsum), nothing is stored
    const uint32_t drop_thr = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float drop_scale = 1.0f / (1.0f - (float)drop_thr / 65536.0f);
    const bool vec = (ldc % 4 == 0) && (N % 4 == 0) && (((uintptr_t)C & 15) == 0) &&
                     (!HAS_R || ((p.ldr % 4 == 0) && (((uintptr_t)R & 15) == 0))) &&
                     stage_bytes >= (int)(blockDim.x / 64) * 32 * F32_SLAB_LD * 4;
    if (!vec) {                                          // odd shapes: direct stores from the fragments
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fr;
            if (col >= N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 64 + i * 16 + cq * 4 + r;
                    if (row < M) {
                        float v = acc[i][j][r] + bv;
                        if (drop_thr) v *= ((fb_hash32(p.drop_seed + (uint32_t)row * (uint32_t)N + (uint32_t)col) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                        if (HAS_R) v += R[(size_t)row * p.ldr + col];
                        C[(size_t)row * ldc + col] = v;
                        if (p.C16) ((bf16_t*)p.C16)[(size_t)row * p.ldc16 + col] = f32_to_bf16(v);
                    }
                }
        }
        return;
    }
    float* my = sStage + (size_t)wave * (32 * F32_SLAB_LD);
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + fr;
        bv[j] = (p.bias && col < N) ? p.bias[col] : 0.f;
    }
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    my[(ii * 16 + cq * 4 + r) * F32_SLAB_LD + j * 16 + fr] = acc[half * 2 + ii][j][r] + bv[j];
        const int col = n0 + wn * 64 + c4;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rl = k * 4 + rr;
            const int row = m0 + wm * 64 + half * 32 + rl;
            float4 v = *(const float4*)&my[rl * F32_SLAB_LD + c4];
            if (row < M && col < N) {
                if (drop_thr) {                          // (uniform branch)
                    const uint32_t k0 = p.drop_seed + (uint32_t)row * (uint32_t)N + (uint32_t)col;
                    v.x *= ((fb_hash32(k0) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                    v.y *= ((fb_hash32(k0 + 1u) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                    v.z *= ((fb_hash32(k0 + 2u) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                    v.w *= ((fb_hash32(k0 + 3u) & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
                }
                if (HAS_R) {
                    const float4 q = *(const float4*)&R[(size_t)row * p.ldr + col];
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                *(float4*)&C[(size_t)row * ldc + col] = v;
                if (p.C16) *(uint2*)((bf16_t*)p.C16 + (size_t)row * p.ldc16 + col) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
            }
        }
    }
}

// fp32 outputs WITH an activation (value + optionally the stored derivative as a second fp32 tile) or with the activation adjoint
// of an MLP (C = acc * act'(aux), aux an fp32 [M, N] tile): the epilogues the fp32-storage modes ('fp32', 'bf16x3') run on every
// hidden layer -- same wave-private slab transpose and 256-byte row segments as gemm_epilogue_f32 (the generic epilogue stores
// 4-byte elements, 64-byte pieces of four rows per instruction).  AUXD: 0 none, 1 relu'(aux) = [aux > 0], 2 aux holds act' itself.
// Returns false when the shape / alignment does not allow 16-byte accesses (the caller falls back to the generic epilogue).
template <int BM_, int ACT, bool HAS_C2, int AUXD>
__device__ __forceinline__ bool gemm_epilogue_f32x(const FabindGemmArgs& p, f32x4_t (&acc)[4][4], float* sStage, int stage_bytes,
                                                   int M, int N, int ldc, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, cq = lane >> 4;
    float* C = (float*)p.C;
    float* C2 = (float*)p.C2;
    const float* AX = (const float*)p.aux;
    const bool vec = (ldc % 4 == 0) && (N % 4 == 0) && (((uintptr_t)C & 15) == 0) && (!HAS_C2 || (((uintptr_t)C2 & 15) == 0)) &&
                     (AUXD == 0 || ((p.ldaux % 4 == 0) && (((uintptr_t)AX & 15) == 0))) &&
                     stage_bytes >= (int)(blockDim.x / 64) * 32 * F32_SLAB_LD * 4;
    if (!vec) return false;
    float* my = sStage + (size_t)wave * (32 * F32_SLAB_LD);
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + fr;
        bv[j] = (p.bias && col < N) ? p.bias[col] : 0.f;
    }
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    my[(ii * 16 + cq * 4 + r) * F32_SLAB_LD + j * 16 + fr] = acc[half * 2 + ii][j][r] + bv[j];
        const int col = n0 + wn * 64 + c4;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rl = k * 4 + rr;
            const int row = m0 + wm * 64 + half * 32 + rl;
            const float4 v = *(const float4*)&my[rl * F32_SLAB_LD + c4];
            if (row < M && col < N) {
                float o[4] = {v.x, v.y, v.z, v.w};
                if (AUXD != 0) {
                    const float4 a = *(const float4*)&AX[(size_t)row * p.ldaux + col];
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= (AUXD == 1) ? (av[e] > 0.f ? 1.f : 0.f) : av[e];
                }
                if (HAS_C2) {
                    const float d0 = fast_dact<ACT>(o[0]), d1 = fast_dact<ACT>(o[1]), d2 = fast_dact<ACT>(o[2]), d3 = fast_dact<ACT>(o[3]);
                    if (p.c2_bf16) *(uint2*)((bf16_t*)p.C2 + (size_t)row * ldc + col) = make_uint2(pack2_bf16(d0, d1), pack2_bf16(d2, d3));
                    else *(float4*)&C2[(size_t)row * ldc + col] = make_float4(d0, d1, d2, d3);
                }
                const float a0 = fast_act<ACT>(o[0]), a1 = fast_act<ACT>(o[1]), a2 = fast_act<ACT>(o[2]), a3 = fast_act<ACT>(o[3]);
                *(float4*)&C[(size_t)row * ldc + col] = make_float4(a0, a1, a2, a3);
                // (round 6) the bf16 operand copy of the activated tile next to it: the hidden layer of a split-precision MLP stays fp32 for
                // the second contraction while the backward keeps its bf16 operands -- no cast pass
                if (p.C16) *(uint2*)((bf16_t*)p.C16 + (size_t)row * p.ldc16 + col) = make_uint2(pack2_bf16(a0, a1), pack2_bf16(a2, a3));
            }
        }
    }
    return true;
}

// returns true when the fast epilogue applies (decided per launch on the host -> p.epi_fast)
template <int BM_, bool RAW_BARRIER>
__device__ __forceinline__ bool gemm_epilogue_dispatch(const FabindGemmArgs& p, f32x4_t (&acc)[4][4], float* sDot, bf16_t* sOut,
                                                       int M, int N, int ldc, int m0, int n0, int stage_bytes = 0) {
#define EPI(ACT, HC, HC2, HD) gemm_epilogue_fast<BM_, ACT, HC, HC2, HD, RAW_BARRIER>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true
    switch (p.epi_fast) {
        case 1: EPI(FB_ACT_NONE, true, false, false);
        case 2: EPI(FB_ACT_SILU, true, false, false);
        case 3: EPI(FB_ACT_SILU, true, true, false);
        case 4: EPI(FB_ACT_SILU, false, false, true);
        case 5: EPI(FB_ACT_RELU, true, false, false);
        case 6: EPI(FB_ACT_RELU, false, false, true);
        case 7: gemm_epilogue_fast<BM_, FB_ACT_SILU, true, false, true, RAW_BARRIER, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 8: gemm_epilogue_fast<BM_, FB_ACT_RELU, true, false, true, RAW_BARRIER, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 11: gemm_epilogue_fast<BM_, FB_ACT_NONE, true, false, false, RAW_BARRIER, false, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 12: gemm_epilogue_fast<BM_, FB_ACT_RELU, true, false, false, RAW_BARRIER, false, false, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 13: gemm_epilogue_fast<BM_, FB_ACT_RELU, false, false, true, RAW_BARRIER, false, false, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 14: gemm_epilogue_fast<BM_, FB_ACT_NONE, true, false, false, RAW_BARRIER, false, false, false, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;
        case 16: gemm_epilogue_fast<BM_, FB_ACT_NONE, true, false, false, RAW_BARRIER, false, false, false, false, true>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;   // bf16 C = acc + bias + bf16 R
        case 15: gemm_epilogue_fast<BM_, FB_ACT_RELU, true, false, true, RAW_BARRIER, false>(p, acc, sDot, sOut, M, N, ldc, m0, n0); return true;   // C = relu output (+ dropout), row-dot of it
        case 20: return gemm_epilogue_f32x<BM_, FB_ACT_NONE, false, 1>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 21: return gemm_epilogue_f32x<BM_, FB_ACT_NONE, false, 2>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 22: return gemm_epilogue_f32x<BM_, FB_ACT_SILU, true, 0>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 23: return gemm_epilogue_f32x<BM_, FB_ACT_SILU, false, 0>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 24: return gemm_epilogue_f32x<BM_, FB_ACT_RELU, false, 0>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 25: return gemm_epilogue_f32x<BM_, FB_ACT_RELU, true, 0>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0);
        case 9: gemm_epilogue_f32<BM_, false>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0); return true;
        case 10: gemm_epilogue_f32<BM_, true>(p, acc, (float*)sOut, stage_bytes, M, N, ldc, m0, n0); return true;
        default: return false;
    }
#undef EPI
}

template <typename TA, typename TM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(FabindGemmArgs p) {
    constexpr int LS = LdsStride<TM>::v;
    __shared__ __attribute__((aligned(16))) TM sA[BM * LS];
    __shared__ __attribute__((aligned(16))) TM sB[BN * LS];
    __shared__ float sDot[2][BM];

    int M = p.M, N = p.N, ldc = p.ldc;
    long a_row0 = 0, w_row0 = 0, c_off = 0;
    int Kg = p.K, lda_g = p.lda, K1g = p.K1;
    long a_eoff = 0, w_eoff = 0;
    if (p.groups && p.groups_ext) {
        // extended descriptor int32[16]: {M, N, K, lda, a_off lo/hi, w_off lo/hi, c_off lo/hi, ldc}: fully ragged operands
        const int* g = p.groups + (size_t)blockIdx.z * 16;
        M = g[0]; N = g[1]; Kg = g[2]; lda_g = g[3];
        a_eoff = (long)(unsigned)g[4] | ((long)g[5] << 32);
        w_eoff = (long)(unsigned)g[6] | ((long)g[7] << 32);
        c_off = (long)(unsigned)g[8] | ((long)g[9] << 32);
        ldc = g[10];
        K1g = Kg;
    } else if (p.groups) {
        const int* g = p.groups + (size_t)blockIdx.z * 8;
        a_row0 = g[0]; M = g[1]; w_row0 = g[2]; N = g[3];
        c_off = (long)(unsigned)g[4] | ((long)g[5] << 32);
        if (g[6] > 0) ldc = g[6];
    }
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M || n0 >= N) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const TA* A = (const TA*)p.A + a_row0 * lda_g + a_eoff;
    const TA* A2 = p.A2 ? (const TA*)p.A2 + a_row0 * p.lda2 : nullptr;
    const TM* W = (const TM*)p.W + w_row0 * p.ldw + w_eoff;
    int K = Kg;
    const int K1 = K1g;
    int kbeg = 0;
    if (!p.groups && p.k_splits > 1) {      // split-K: this work-group owns K range [kbeg, K)
        const int per = ((p.K / p.k_splits + BK - 1) / BK) * BK;
        kbeg = blockIdx.z * per;
        K = min(p.K, kbeg + per);
        c_off = (long)blockIdx.z * M * ldc;
    }
    if (p.groups_ext) { a_row0 = 0; w_row0 = 0; }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // each thread stages 2 x vec8 of A and 2 x vec8 of W per K-tile
    const int lr = tid >> 2, lc = (tid & 3) * 8;
    Vec8 ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = lr + 64 * i;
            int gm = m0 + r, gk = k0 + lc;
            Vec8 z;
#pragma unroll
            for (int q = 0; q < 8; ++q) z.v[q] = 0.f;
            ra[i] = z;
            rb[i] = z;
            if (gm < M && gk < K) {
                const TA* src = (gk < K1) ? A + (size_t)gm * lda_g + gk : A2 + (size_t)gm * p.lda2 + (gk - K1);
                ra[i] = load8<TA>(src);
                if (p.act_pro != FB_ACT_NONE) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) ra[i].v[q] = apply_act(ra[i].v[q], p.act_pro);
                }
            }
            int gn = n0 + r;
            if (gn < N && gk < K) rb[i] = load8<TM>(W + (size_t)gn * p.ldw + gk);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = lr + 64 * i;
            store8_lds<TM>(&sA[r * LS + lc], ra[i]);
            store8_lds<TM>(&sB[r * LS + lc], rb[i]);
        }
    };

    const int nk = (K - kbeg + BK - 1) / BK;
    fetch(kbeg);
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        stash();
        __syncthreads();
        if (kt + 1 < nk) fetch(kbeg + (kt + 1) * BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const TM* ap = &sA[(wm * 64 + i * 16 + fr) * LS + fk];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const TM* bp = &sB[(wn * 64 + j * 16 + fr) * LS + fk];
                mma_k32<TM>(acc[i][j], ap, bp);
            }
        }
    }

    gemm_epilogue<BM>(p, acc, &sDot[0][0], nullptr, M, N, ldc, a_row0, w_row0, c_off, m0, n0);
}

// ------------------------------------------------------------------------------------------------
// Fast path: bf16 A and W, no prologue activation, K % 64 == 0.  Both operand tiles go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double-buffered, BK = 64 (32 MFMAs per wave between
// barriers).  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row & 7) on the SOURCE
// address (the LDS image of an LDS-DMA is lane-linear) and on the fragment read.
// ------------------------------------------------------------------------------------------------
#define FBK 64
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(FabindGemmArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t sT[2][2][BM * FBK];   // [buf][A|B][128 rows x 64]
    __shared__ float sDot[2][BM];
    int M = p.M, N = p.N, ldc = p.ldc;
    long a_row0 = 0, w_row0 = 0, c_off = 0;
    if (p.groups) {
        const int* g = p.groups + (size_t)blockIdx.z * 8;
        a_row0 = g[0]; M = g[1]; w_row0 = g[2]; N = g[3];
        c_off = (long)(unsigned)g[4] | ((long)g[5] << 32);
        if (g[6] > 0) ldc = g[6];
    }
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M || n0 >= N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A + a_row0 * p.lda;
    const bf16_t* W = (const bf16_t*)p.W + w_row0 * p.ldw;
    const int K = p.K;

    // per-lane source rows/chunks of this wave's 4 LDS-DMA pieces per operand (8 rows x 128 B each)
    const int lrow = lane >> 3, lchunk = lane & 7;
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + lrow;
        const int chunk = lchunk ^ (row & 7);
        const int gm = min(m0 + row, M - 1), gn = min(n0 + row, N - 1);   // clamp: out-of-range rows are never stored
        srcA[j] = A + (size_t)gm * p.lda + chunk * 8;
        srcB[j] = W + (size_t)gn * p.ldw + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 4 + j;
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + k0), (lptr_t)&sT[buf][0][q * 8 * FBK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + k0), (lptr_t)&sT[buf][1][q * 8 * FBK], 16, 0, 0);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / FBK;
    const int fr = lane & 15, fq = lane >> 4;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * FBK);
        const bf16_t* tA = sT[buf][0];
        const bf16_t* tB = sT[buf][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
                af[i] = *(const bf16x8_t*)&tA[ra * FBK + (((kk * 4 + fq) ^ (ra & 7)) * 8)];
                bfr[i] = *(const bf16x8_t*)&tB[rb * FBK + (((kk * 4 + fq) ^ (rb & 7)) * 8)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    gemm_epilogue<BM>(p, acc, &sDot[0][0], &sT[0][0][0], M, N, ldc, a_row0, w_row0, c_off, m0, n0);
}

// ------------------------------------------------------------------------------------------------
// Pipelined LDS-DMA variant: NSTAGE-deep ring of (A|W) k-tiles, counted s_waitcnt vmcnt(N) so that
// NSTAGE-2 tiles stay in flight across the (raw) barrier -- the HBM latency of this short-K, long-M
// edge GEMM (K = 512: 8-16 k-steps per block) is what bounds it, not the matrix pipe.
// ------------------------------------------------------------------------------------------------
// bytes of the tile ring; at least one bf16 output staging tile [BM_][BN+8] must fit in it
template <int WM, int BK_, int NSTAGE> __host__ __device__ constexpr size_t lds_tile_bytes() {
    size_t ring = (size_t)NSTAGE * (WM * 64 + BN) * BK_ * 2, out = (size_t)WM * 64 * (BN + 8) * 2;
    return ring > out ? ring : out;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WM, int BK_, int NSTAGE, int MINW = 1>
__global__ __launch_bounds__(WM * 128, MINW) void gemm_bf16_pipe_kernel(FabindGemmArgs p, const int row_tile0) {
    constexpr int BM_ = WM * 64, NW = WM * 2, NT = NW * 64;
    constexpr int ROWS = BM_ + BN;                    // A rows then W rows share one row space per stage
    constexpr int RPP = 1024 / (BK_ * 2);             // rows per 1-KiB LDS-DMA piece
    constexpr int PT = ROWS / RPP, PPW = PT / NW;     // pieces per stage / per wave
    constexpr int LPR = 64 / RPP;                     // lanes (16-B chunks) per row
    static_assert(PT % NW == 0, "pieces must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sT = (bf16_t*)smem;                       // [NSTAGE][ROWS][BK_]
    float* sDot = (float*)(smem + lds_tile_bytes<WM, BK_, NSTAGE>());
    int M = p.M, N = p.N, ldc = p.ldc;
    long a_row0 = 0, w_row0 = 0, c_off = 0;
    if (p.groups) {
        const int* g = p.groups + (size_t)blockIdx.z * 8;
        a_row0 = g[0]; M = g[1]; w_row0 = g[2]; N = g[3];
        c_off = (long)(unsigned)g[4] | ((long)g[5] << 32);
        if (g[6] > 0) ldc = g[6];
    }
    // XCD-aware tile order (non-grouped launches are 1-D): block L runs on XCD L % 8, so give every XCD a
    // contiguous run of tiles -- the N-tiles of one M-panel then share that XCD's L2 copy of the A panel.
    int bx = blockIdx.x, by = blockIdx.y;
    if (!p.groups) {
        const int nbx = (N + BN - 1) / BN, total = gridDim.x;
        const int L = blockIdx.x, q = total / 8, r = total % 8, xcd = L % 8;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + L / 8;
        bx = t % nbx; by = t / nbx + row_tile0;      // (row_tile0: first row tile of this launch)
    }
    const int m0 = by * BM_, n0 = bx * BN;
    if (m0 >= M || n0 >= N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A + a_row0 * p.lda;
    const bf16_t* W = (const bf16_t*)p.W + w_row0 * p.ldw;
    int K = p.K, kbeg = 0;
    if (!p.groups && p.k_splits > 1) {      // split-K over blockIdx.y (non-grouped launches are 1-D in x)
        const int per = ((p.K / p.k_splits + BK_ - 1) / BK_) * BK_;
        kbeg = blockIdx.y * per;
        K = min(p.K, kbeg + per);
        c_off = (long)blockIdx.y * M * ldc;
    }

    const bf16_t* src[PPW];
    const bf16_t* src2[PPW];   // second K-segment of A (K-concatenated operand); W rows keep one segment
    const int K1 = p.K1;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wave * PPW + j) * RPP + lane / LPR;      // row in the combined (A|W) space
        const int cl = lane % LPR;                                 // LDS chunk slot
        const int chunk = (BK_ == 64) ? (cl ^ (row & 7)) : (cl ^ ((row >> 2) & 3));
        if (row < BM_) {
            const size_t gm = (size_t)min(m0 + row, M - 1);
            src[j] = A + gm * p.lda + chunk * 8;
            src2[j] = p.A2 ? (const bf16_t*)p.A2 + (a_row0 + gm) * p.lda2 + chunk * 8 - K1 : src[j];
        } else {
            src[j] = W + (size_t)min(n0 + row - BM_, N - 1) * p.ldw + chunk * 8;
            src2[j] = src[j];
        }
    }
    auto stage = [&](int st, int k0) {
        bf16_t* base = sT + (size_t)st * ROWS * BK_;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)((k0 < K1 ? src[j] : src2[j]) + k0),
                                             (lptr_t)(base + (size_t)(wave * PPW + j) * RPP * BK_), 16, 0, 0);
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = max(0, (K - kbeg) / BK_);
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, kbeg + s * BK_);
    for (int kt = 0; kt < nk; ++kt) {
        // tiles kt+1 .. kt+NSTAGE-2 may stay in flight
        const int ahead = min(NSTAGE - 2, nk - 1 - kt);
        if (ahead >= 2) wait_vmcnt<2 * PPW>(); else if (ahead == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < nk) stage((kt + NSTAGE - 1) % NSTAGE, kbeg + (kt + NSTAGE - 1) * BK_);
        const bf16_t* tA = sT + (size_t)(kt % NSTAGE) * ROWS * BK_;
        const bf16_t* tB = tA + BM_ * BK_;
#pragma unroll
        for (int kk = 0; kk < BK_ / 32; ++kk) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
                const int ca = (BK_ == 64) ? ((kk * 4 + fq) ^ (ra & 7)) : (fq ^ ((ra >> 2) & 3));
                const int cb = (BK_ == 64) ? ((kk * 4 + fq) ^ (rb & 7)) : (fq ^ ((rb >> 2) & 3));
                af[i] = *(const bf16x8_t*)&tA[ra * BK_ + ca * 8];
                bfr[i] = *(const bf16x8_t*)&tB[rb * BK_ + cb * 8];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (p.epi_fast && gemm_epilogue_dispatch<BM_, false>(p, acc, sDot, sT, M, N, ldc, m0, n0, (int)lds_tile_bytes<WM, BK_, NSTAGE>())) return;
    gemm_epilogue<BM_>(p, acc, sDot, sT, M, N, ldc, a_row0, w_row0, c_off, m0, n0);
}

// (Round 4's 256 x 256 big-tile variant -- bitwise equal and slower, profiles/r04_gemm_big_tile.txt -- was a knob nobody reached:
// retired in round 5, source in tools/probes/retired/gemm_big_tile.hip.txt.)
// ------------------------------------------------------------------------------------------------
template <int WM, int BK_, int NSTAGE>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_persist_kernel(FabindGemmArgs p, int tiles_total, int nbx) {
    constexpr int BM_ = WM * 64, NW = WM * 2;
    constexpr int ROWS = BM_ + BN;
    constexpr int RPP = 1024 / (BK_ * 2);
    constexpr int PT = ROWS / RPP, PPW = PT / NW;
    constexpr int LPR = 64 / RPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sT = (bf16_t*)smem;
    float* sDot = (float*)(smem + (size_t)NSTAGE * ROWS * BK_ * 2);
    bf16_t* sOutP = (bf16_t*)(smem + (size_t)NSTAGE * ROWS * BK_ * 2 + 2 * BM_ * sizeof(float));   // own staging tile
    const int M = p.M, N = p.N, ldc = p.ldc, K = p.K, K1 = p.K1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const int t0 = (int)((long)blockIdx.x * tiles_total / gridDim.x), t1 = (int)((long)(blockIdx.x + 1) * tiles_total / gridDim.x);
    const int nk = K / BK_;
    const int G = (t1 - t0) * nk;               // k-steps this work-group executes
    if (G <= 0) return;

    const bf16_t* src[PPW];
    const bf16_t* src2[PPW];
    auto set_src = [&](int tile) {
        const int m0 = (tile / nbx) * BM_, n0 = (tile % nbx) * BN;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = (wave * PPW + j) * RPP + lane / LPR;
            const int cl = lane % LPR;
            const int chunk = (BK_ == 64) ? (cl ^ (row & 7)) : (cl ^ ((row >> 2) & 3));
            if (row < BM_) {
                const size_t gm = (size_t)min(m0 + row, M - 1);
                src[j] = A + gm * p.lda + chunk * 8;
                src2[j] = p.A2 ? (const bf16_t*)p.A2 + gm * p.lda2 + chunk * 8 - K1 : src[j];
            } else {
                src[j] = W + (size_t)min(n0 + row - BM_, N - 1) * p.ldw + chunk * 8;
                src2[j] = src[j];
            }
        }
    };
    int s_tile = t0, s_kt = 0, s_g = 0;          // staging cursor (runs NSTAGE-1 steps ahead of compute)
    set_src(s_tile);
    auto stage_next = [&]() {
        bf16_t* base = sT + (size_t)(s_g % NSTAGE) * ROWS * BK_;
        const int k0 = s_kt * BK_;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)((k0 < K1 ? src[j] : src2[j]) + k0),
                                             (lptr_t)(base + (size_t)(wave * PPW + j) * RPP * BK_), 16, 0, 0);
        ++s_g;
        if (++s_kt == nk) { s_kt = 0; ++s_tile; if (s_tile < t1) set_src(s_tile); }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE - 1; ++s_)
        if (s_g < G) stage_next();
    int c_tile = t0, c_kt = 0;
    for (int g = 0; g < G; ++g) {
        const int ahead = min(NSTAGE - 2, G - 1 - g);
        if (ahead >= 2) wait_vmcnt<2 * PPW>(); else if (ahead == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s_g < G) stage_next();
        const bf16_t* tA = sT + (size_t)(g % NSTAGE) * ROWS * BK_;
        const bf16_t* tB = tA + BM_ * BK_;
#pragma unroll
        for (int kk = 0; kk < BK_ / 32; ++kk) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
                const int ca = (BK_ == 64) ? ((kk * 4 + fq) ^ (ra & 7)) : (fq ^ ((ra >> 2) & 3));
                const int cb = (BK_ == 64) ? ((kk * 4 + fq) ^ (rb & 7)) : (fq ^ ((rb >> 2) & 3));
                af[i] = *(const bf16x8_t*)&tA[ra * BK_ + ca * 8];
                bfr[i] = *(const bf16x8_t*)&tB[rb * BK_ + cb * 8];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (++c_kt == nk) {
            if (!(p.epi_fast && gemm_epilogue_dispatch<BM_, true>(p, acc, sDot, sOutP, M, N, ldc, (c_tile / nbx) * BM_, (c_tile % nbx) * BN, BM_ * OUT_LD * 2)))
                gemm_epilogue<BM_, true>(p, acc, sDot, nullptr, M, N, ldc, 0, 0, 0, (c_tile / nbx) * BM_, (c_tile % nbx) * BN);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            c_kt = 0;
            ++c_tile;
        }
    }
}

template <int WM, int BK_, int NSTAGE>
static int launch_persist(const FabindGemmArgs& p, hipStream_t stream) {
    constexpr int BM_ = WM * 64;
    const size_t lds = (size_t)NSTAGE * (BM_ + BN) * BK_ * 2 + 2 * BM_ * sizeof(float) + (size_t)BM_ * OUT_LD * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_persist_kernel<WM, BK_, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM_ - 1) / BM_;
    const int tiles = nbx * nby;
    int grid = 256;                              // 144 KiB of LDS per work-group: one per CU
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL((gemm_bf16_persist_kernel<WM, BK_, NSTAGE>), dim3(grid), dim3(WM * 128), lds, stream, p, tiles, nbx);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TN contraction for weight gradients:  C[m, n] = sum_e Y[e, m] * X[e, n]   (Y, X row-major bf16, e = rows)
// Both operands have the contraction index as their SLOW dimension, so the MFMA fragments (8 consecutive e per
// lane) are formed with the LDS transpose read ds_read_b64_tr_b16: in every 16-lane group, lane l receives
// element (l % 4) of the 8-byte pieces addressed by lanes {l/4, 4 + l/4, 8 + l/4, 12 + l/4} (probe:
// tools/probes/tr_probe.hip).  Tiles [32 e][256 m] and [32 e][128 n] arrive by LDS-DMA; their 32-byte granules
// are XOR-swizzled per row on the source side so that the transpose reads are bank-conflict free.
// Split over e across blockIdx.y; fp32 partials C[split][M][N].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* p_lo, const bf16_t* p_hi) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(uintptr_t)p_lo) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"((unsigned)(uintptr_t)p_hi) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)::"memory");
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ int tn_swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int N> __device__ __forceinline__ void wait_vmcnt_upto(int n) {   // s_waitcnt vmcnt(n * N) for a uniform n in 0..3
    if (n >= 3) wait_vmcnt<3 * N>(); else if (n == 2) wait_vmcnt<2 * N>(); else if (n == 1) wait_vmcnt<N>(); else wait_vmcnt<0>();
}

template <int NSTAGE, int NW, int TN_, bool JOBS = false>
__global__ __launch_bounds__(NW * 64) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ Y, int ldy, const bf16_t* __restrict__ X,
                                                           int ldx, float* C, int M, int N, int E, int e_per,
                                                           const bf16_t* __restrict__ zero_page, const int* groups,
                                                           int n_tiles, int n_splits, int xf, int cs,
                                                           const FabindTnJob* __restrict__ jobs, const int* __restrict__ wg_job) {
    // JOBS: one launch carries MANY independent contractions (fabind_gemm_tn_multi: the weight gradients of a whole backward pass,
    // queued by the host and flushed together -- at 9,000 rows one contraction is 32 work-groups on a 256-CU chip).  A work-group
    // reads its job from the device table (uniform: scalar loads) and then runs exactly the single-launch code path below.
    unsigned bid = blockIdx.x;
    if constexpr (JOBS) {
        const FabindTnJob* jb = jobs + wg_job[blockIdx.x];
        Y = (const bf16_t*)jb->Y; X = (const bf16_t*)jb->X; C = jb->C_part;
        ldy = jb->ldy; ldx = jb->ldx; M = jb->M; N = jb->N; E = jb->E; e_per = jb->e_per;
        n_tiles = jb->n_tiles; n_splits = jb->splits; cs = jb->with_colsum;
        bid -= (unsigned)jb->wg0;
    }
    // Output tile 256 x TN_, NW waves per work-group: TN_/64 across N (64 columns each), the rest across M.
    //   <3, 4, 128>: 128x64 per wave, 124 VGPRs + 128 AGPRs, two work-groups per CU (3 x 24 KiB of LDS each);
    //   <3, 8, 128>: 64x64 per wave, one work-group per CU (kept as a knob);
    //   <NSTAGE, 8, 256>: 128x64 per wave, ONE work-group per CU with NSTAGE x 32 KiB of LDS: a third fewer operand bytes per flop from
    //   L2, and -- what matters on long operands -- NSTAGE - 1 stages of DISTINCT rows in flight per e-range instead of two: the tiles of
    //   one e-range request the same rows, so the bytes in flight at the HBM are (ranges running) x (stages in flight) x 64 KiB, and the
    //   contraction over 1.5 M edges is bound by exactly that (probe knobs 1 / 4: loads off 625 us, loads hitting in cache 800 us,
    //   loads from HBM 1,050 us at 3.0 TB/s with two stages in flight).
    constexpr int TM = 256, BKE = 32;
    constexpr int WN = TN_ / 64, WMW = NW / WN;                    // waves across N / across M
    constexpr int MI = TM / WMW / 16;                             // 16-row granules of the Y tile per wave
    constexpr int PY = (BKE * TM * 2) / 1024, PX = (BKE * TN_ * 2) / 1024;    // one-KiB pieces per k-step
    constexpr int YPW = PY / NW, XPW = PX / NW, PPW = YPW + XPW;               // per wave: 4 + 2 (4 waves), 2 + 1 / 2 + 2 (8 waves)
    constexpr int XRPP = 1024 / (TN_ * 2), XCPR = TN_ * 2 / 16;               // X piece: rows per piece, 16-B chunks per row
    static_assert(NSTAGE >= 3 && NSTAGE <= 5 && YPW >= 1 && XPW >= 1, "tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr unsigned STAGE_BYTES = BKE * (TM + TN_) * 2, X_BYTES0 = BKE * TM * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: LDS-DMA destinations and piece ids live in SGPRs
    const int wm = wave / WN, wn = wave % WN;
    long c_goff = 0;
    if (groups) {   // ragged batch: int32[8] = {y_off lo/hi (elements), ldy, M, x_row0, E, c_off lo/hi (elements)}
        const int* g = groups + (size_t)blockIdx.z * 8;
        Y += (long)(unsigned)g[0] | ((long)g[1] << 32);
        ldy = g[2]; M = g[3];
        X += (size_t)g[4] * ldx;
        E = g[5];
        c_goff = (long)(unsigned)g[6] | ((long)g[7] << 32);
    }
    const int nbx = (N + TN_ - 1) / TN_;
    // XCD-aware order: consecutive work-group ids go round-robin to the 8 XCDs, so id = (split_hi, tile, split_lo)
    // puts every output tile of one e-range on the SAME XCD -- its Y / X slabs are fetched into that L2 once
    // instead of once per tile (measured 3x the algorithmic HBM reads with tile-major order).
    const int tile = (bid >> 3) % n_tiles;
    const int split = ((bid >> 3) / n_tiles) * 8 + (bid & 7);
    if (split >= n_splits) return;
    const int m0 = (tile / nbx) * TM, n0 = (tile % nbx) * TN_;
    if (m0 >= M) return;
    const int ebeg = split * e_per, eend = min(E, ebeg + e_per);
    const int nk = (eend - ebeg + BKE - 1) / BKE;

    // Staging.  A wave moves YPW pieces of the Y tile (2 rows of 512 B each) and XPW pieces of the X tile (4 rows of 256 B): which is
    // which is known at compile time, a lane's source pointer is formed ONCE and advanced by 32 rows per k-step, a column outside the
    // matrix points at the zero page with a zero stride, and a row past the end of the range selects the zero page (two v_cndmask).
    // (Round 2: the previous form recomputed a 64-bit address per piece per k-step under divergent branches -- ~40 instructions per
    // piece, VALU busy 31 % of the kernel's time.)
    const bf16_t* gp[PPW];
    long ginc[PPW];
    int grow[PPW];
    unsigned gdst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        if (j < YPW) {
            const int q = wave * YPW + j;
            const int r = q * 2 + (lane >> 5), pp = lane & 31;           // 16-B chunk position in the LDS row
            const int col = m0 + ((((pp >> 1) ^ tn_swz(r)) << 1) | (pp & 1)) * 8;
            const bool ok = col < M;
            gp[j] = ok ? Y + (size_t)(ebeg + r) * ldy + col : zero_page;
            ginc[j] = ok ? (long)BKE * ldy : 0;
            grow[j] = ebeg + r;
            gdst[j] = (unsigned)q * 1024u;
        } else {
            const int q = wave * XPW + (j - YPW);
            const int r = q * XRPP + lane / XCPR, pp = lane % XCPR;
            const int col = n0 + ((((pp >> 1) ^ tn_swz(r)) << 1) | (pp & 1)) * 8;
            const bool ok = col < N;
            gp[j] = ok ? X + (size_t)(ebeg + r) * ldx + col : zero_page;
            ginc[j] = ok ? (long)BKE * ldx : 0;
            grow[j] = ebeg + r;
            gdst[j] = X_BYTES0 + (unsigned)q * 1024u;
        }
    }
    const uintptr_t zp = (uintptr_t)zero_page;
    auto stage = [&](unsigned st_bytes) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const uintptr_t a = grow[j] < eend ? (uintptr_t)gp[j] : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(smem + st_bytes + gdst[j]), 16, 0, 0);
            if (!(xf & 4)) gp[j] += ginc[j];                        // probe bit 4: every k-step re-reads the first rows (cache hits)
            grow[j] += BKE;
        }
    };
    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE; ++s_)
        if (s_ < nk) stage((unsigned)s_ * STAGE_BYTES);
    // transpose-read offsets of this lane inside a stage (bytes): Y granule i at row r1, X granule i at row r1; the second half of
    // a fragment sits 4 rows further down in the SAME swizzle class (tn_swz(r1 + 4) == tn_swz(r1)) -> an immediate offset
    const int g = lane >> 4, sl = lane & 15;
    const int r1 = g * 8 + (sl >> 2), co = (sl & 3) * 4;
    const int z1 = tn_swz(r1);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned offY[MI], offX[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) offY[i] = lds0 + (unsigned)(r1 * TM + (((wm * MI + i) ^ z1) * 16) + co) * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) offX[i] = lds0 + X_BYTES0 + (unsigned)(r1 * TN_ + (((wn * 4 + i) ^ z1) * 16) + co) * 2u;
    // Schedule of one k-step (round 2 -- the previous loop read all fragments, waited, then ran all MFMAs: its LDS latency,
    // its barrier and its operand wait added up instead of overlapping, 34 % of the MFMA peak):
    //   top:  the first half of the Y granules (+ X) of stage kt is IN REGISTERS; issue the reads of the SECOND half
    //   A:    MFMAs of the first half
    //   mid:  second half landed in registers -> this wave is done with stage kt in LDS; wait for this wave's pieces of stage kt+1;
    //         barrier (every wave done with stage kt, all of stage kt+1 landed); refill the slot of stage kt with stage kt+NSTAGE; issue the
    //         reads of the first half (+ X) of stage kt+1 into the OTHER register set
    //   B:    MFMAs of the second half; then the wait for the reads just issued
    // so every LDS read has half a k-step of MFMAs (16 x 16 cycles) to land under, and operand loads have NSTAGE - 1 k-steps.
    //
    // ROUND 6 -- the root cause of the weight-gradient mismatches under device sharing (profiles/r05_contention.txt).  Every LDS read
    // here is inline asm whose result lands ASYNCHRONOUSLY; the compiler believes an asm output is valid the moment the statement has
    // issued.  Rounds 2-5 kept ONE set of first-half registers: the reads of stage kt+1 were issued in iteration kt and waited for at
    // the top of iteration kt+1.  The old values were still live (phase B), so the new ones landed in other registers and the register
    // allocator resolved the loop-carried value with copies at the back-edge -- `v_mov_b64 v[162:163], v[204:205]` AHEAD of the
    // `s_waitcnt lgkmcnt` that guarded v[204:205].  Whenever the LDS answered later than the phase-B MFMAs took to issue (another
    // process's waves on the CU), the copy took the PREVIOUS stage's fragments: 2-12 % of the passes with 16 MFMAs of cover (the
    // 256-row-per-wave layouts), every pass with 8 (the 8-wave 128-column layout); never on an otherwise idle device.  A wait tied to
    // the registers ("+v") does not help by itself: the tie makes the allocator place the same copy in front of the wait.  The loop is
    // now unrolled by two over PING-PONG register sets: a set is read, waited for and consumed without ever meeting a live older value
    // of itself, so no copy is needed, and the wait sits in the same straight-line block as the read, ahead of the back-edge.  The
    // code object is checked for this at build time (tools/isa_lint.py, tests/test_isa_lint.py): no instruction may touch the
    // destination of an LDS read that the LGKM counter still covers.
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr int HALF = MI / 2;
    struct Frag { u32x2 xa[4], xb[4], ya[HALF], yb[HALF]; };     // first-half fragments of one stage: X (4 granules) + Y granules [0, HALF)
    Frag fA, fB;
    u32x2 y2a[MI - HALF], y2b[MI - HALF];                        // second-half Y granules: read, waited for and consumed inside one k-step
    auto read_first = [&](unsigned sb, Frag& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.xa[i]) : "v"(offX[i] + sb) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.xb[i]) : "v"(offX[i] + sb), "n"(4 * TN_ * 2) : "memory");
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.ya[i]) : "v"(offY[i] + sb) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.yb[i]) : "v"(offY[i] + sb), "n"(4 * TM * 2) : "memory");
        }
    };
    auto wait_first = [&](Frag& f) {
        if constexpr (MI == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.ya[0]), "+v"(f.ya[1]), "+v"(f.ya[2]), "+v"(f.ya[3]), "+v"(f.yb[0]), "+v"(f.yb[1]), "+v"(f.yb[2]), "+v"(f.yb[3]),
                           "+v"(f.xa[0]), "+v"(f.xa[1]), "+v"(f.xa[2]), "+v"(f.xa[3]), "+v"(f.xb[0]), "+v"(f.xb[1]), "+v"(f.xb[2]), "+v"(f.xb[3])
                         :: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.ya[0]), "+v"(f.ya[1]), "+v"(f.yb[0]), "+v"(f.yb[1]),
                           "+v"(f.xa[0]), "+v"(f.xa[1]), "+v"(f.xa[2]), "+v"(f.xa[3]), "+v"(f.xb[0]), "+v"(f.xb[1]), "+v"(f.xb[2]), "+v"(f.xb[3])
                         :: "memory");
        }
    };
    auto read_second = [&](unsigned sb) {
#pragma unroll
        for (int i = HALF; i < MI; ++i) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(y2a[i - HALF]) : "v"(offY[i] + sb) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(y2b[i - HALF]) : "v"(offY[i] + sb), "n"(4 * TM * 2) : "memory");
        }
    };
    {   // stage 0 landed for everyone -> first-half reads of stage 0 (nk == 0: the reads return whatever the slot holds, never used)
        int fl = min(nk, NSTAGE) - 1;                              // stages still in flight behind stage 0
        if constexpr (NSTAGE == 5) { if (fl >= 4) wait_vmcnt<4 * PPW>(); else wait_vmcnt_upto<PPW>(fl); }
        else wait_vmcnt_upto<PPW>(fl);
        __builtin_amdgcn_s_barrier();
        read_first(0u, fA);
        wait_first(fA);
    }
    unsigned s_cur = 0, s_nxt = STAGE_BYTES;                       // byte offsets of the slots of stage kt and stage kt + 1
    // cs: the column sums of Y (= the bias gradient dY^T 1 of the Linear whose weight gradient this contraction is) ride along: the WN
    // waves that share a row block hold the same Y fragments, wave wn sums the granules i = wn (mod WN) of the work-groups in the first
    // tile column -- 16 VALU instructions per granule and k-step next to 32 MFMAs.  Written behind the M x N partial of the split.
    constexpr int GPW = MI / WN;
    float csum[GPW];
#pragma unroll
    for (int g_ = 0; g_ < GPW; ++g_) csum[g_] = 0.f;
    const bool do_cs = cs && n0 == 0;
    auto sum8 = [](const u32x2 lo, const u32x2 hi) {
        return (__uint_as_float(lo[0] << 16) + __uint_as_float(lo[0] & 0xffff0000u)) + (__uint_as_float(lo[1] << 16) + __uint_as_float(lo[1] & 0xffff0000u)) +
               (__uint_as_float(hi[0] << 16) + __uint_as_float(hi[0] & 0xffff0000u)) + (__uint_as_float(hi[1] << 16) + __uint_as_float(hi[1] & 0xffff0000u));
    };
    // one k-step: consumes the landed set `c` (stage kt), leaves the set `n` landed (stage kt + 1)
    auto k_step = [&](Frag& c, Frag& n, const int kt) {
        read_second(s_cur);
        bf16x8_t bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 rb = {c.xa[i][0], c.xa[i][1], c.xb[i][0], c.xb[i][1]};
            bfr[i] = __builtin_bit_cast(bf16x8_t, rb);
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const u32x4 ra = {c.ya[i][0], c.ya[i][1], c.yb[i][0], c.yb[i][1]};
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, ra);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t af2[MI - HALF];
        if constexpr (MI == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(y2a[0]), "+v"(y2a[1]), "+v"(y2a[2]), "+v"(y2a[3]), "+v"(y2b[0]), "+v"(y2b[1]), "+v"(y2b[2]), "+v"(y2b[3])
                         :: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y2a[0]), "+v"(y2a[1]), "+v"(y2b[0]), "+v"(y2b[1]) :: "memory");
        }
#pragma unroll
        for (int i = HALF; i < MI; ++i) {
            const u32x4 ra = {y2a[i - HALF][0], y2a[i - HALF][1], y2b[i - HALF][0], y2b[i - HALF][1]};
            af2[i - HALF] = __builtin_bit_cast(bf16x8_t, ra);
        }
        if (do_cs) {                                              // all MI fragments of stage kt are in registers here
#pragma unroll
            for (int w_ = 0; w_ < WN; ++w_)
                if (wn == w_) {
#pragma unroll
                    for (int g_ = 0; g_ < GPW; ++g_) {
                        const int gi = w_ + g_ * WN;              // compile-time after unrolling
                        csum[g_] += gi < HALF ? sum8(c.ya[gi < HALF ? gi : 0], c.yb[gi < HALF ? gi : 0])
                                              : sum8(y2a[gi >= HALF ? gi - HALF : 0], y2b[gi >= HALF ? gi - HALF : 0]);
                    }
                }
        }
        if (kt + 1 < nk) {
            wait_vmcnt_upto<PPW>(min(NSTAGE - 2, nk - kt - 2));  // stages kt+2 .. kt+NSTAGE-1 may stay in flight
            __builtin_amdgcn_s_barrier();
            if (kt + NSTAGE < nk && !(xf & 1)) stage(s_cur);       // xf: probe knobs (fabind_gemm_tn_set_exp), 0 in production
        }
        // first-half reads of stage kt+1 -- UNCONDITIONAL (in the last k-step they return the slot's old content, unused), so that the
        // reads, the phase-B MFMAs they hide under and their wait are one straight-line block with no join in between
        read_first(s_nxt, n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = HALF; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2[i - HALF], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wait_first(n);                                            // nothing asynchronous crosses the end of a k-step
        s_cur = s_nxt;
        s_nxt = (s_nxt + STAGE_BYTES == NSTAGE * STAGE_BYTES) ? 0u : s_nxt + STAGE_BYTES;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        k_step(fA, fB, kt);
        if (kt + 1 >= nk) break;
        k_step(fB, fA, kt + 1);
    }
    float* Cs = C + c_goff + (size_t)split * ((size_t)M * N + (cs ? M : 0));
    const int fr = lane & 15, cq = lane >> 4;
    if (do_cs) {
#pragma unroll
        for (int g_ = 0; g_ < GPW; ++g_) {
            float v = csum[g_];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int row = m0 + wm * (MI * 16) + (wn + g_ * WN) * 16 + fr;
            if (cq == 0 && row < M) Cs[(size_t)M * N + row] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (MI * 16) + i * 16 + cq * 4 + r, col = n0 + wn * 64 + j * 16 + fr;
                if (row < M && col < N) Cs[(size_t)row * N + col] = acc[i][j][r];
            }
}

static int g_tn_waves = 16;  // work-group layout of the TN kernel (see the kernel): 16 = 256x256 tile, 8 waves, 4-stage ring (default);
                             // 4 = 256x128, 4 waves, two work-groups per CU (round 1's default); 8 = 256x128, 8 waves
static int g_tn_exp = 0;     // probe knobs: 1 = no operand loads after the pipeline fill (2: retired in round 6 -- fragment reads are unconditional),
                             // 4 = operand loads re-read the first rows of the range (cache hits instead of HBM)
extern "C" void fabind_gemm_tn_set_exp(int x) { g_tn_exp = x; }
extern "C" void fabind_gemm_tn_set_waves(int w) { g_tn_waves = (w == 8 || w == 4 || w == 20) ? w : 16; }
extern "C" int fabind_gemm_tn_tile_n(void) { return g_tn_waves >= 16 ? 256 : 128; }

template <int NSTAGE, int NW, int TN_>
static void tn_launch(const void* Y, int ldy, const void* X, int ldx, float* C_part, int M, int N, int E, int splits, const void* zero_page,
                      const int* groups, int n_groups, int with_colsum, hipStream_t stream) {
    const size_t lds = (size_t)NSTAGE * 32 * (256 + TN_) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<NSTAGE, NW, TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int e_per = ((E + splits - 1) / splits + 31) / 32 * 32;
    const int n_tiles = ((M + 255) / 256) * ((N + TN_ - 1) / TN_);
    dim3 grid(n_tiles * ((splits + 7) / 8 * 8), 1, groups ? n_groups : 1);
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<NSTAGE, NW, TN_>), grid, dim3(NW * 64), lds, stream, (const bf16_t*)Y, ldy, (const bf16_t*)X, ldx,
                       C_part, M, N, E, e_per, (const bf16_t*)zero_page, groups, n_tiles, splits, g_tn_exp, with_colsum,
                       (const FabindTnJob*)nullptr, (const int*)nullptr);
}

// Reduction of the split partials of every job of a fabind_gemm_tn_multi launch: job j's [splits][M * N (+ M)] partials -> its weight
// gradient (out_dt, row stride ldo: two jobs may fill the column blocks of one [M, K1 + K2] gradient) and, with column sums, its fp32
// bias gradient.  Fixed split order, like split_sum_kernel.
__global__ __launch_bounds__(256) void tn_multi_sum_kernel(const FabindTnJob* __restrict__ jobs, const int* __restrict__ blk_job) {
    const FabindTnJob* jb = jobs + blk_job[blockIdx.x];
    const int M = jb->M, N = jb->N, splits = jb->splits;
    const long n_head = (long)M * N, n = n_head + (jb->with_colsum ? M : 0);
    const float* part = jb->C_part;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                              // a block reduces 4,096 elements (the block table stays small)
        const long i = (((long)(blockIdx.x - (unsigned)jb->blk0) * 4 + it) * 256 + threadIdx.x) * 4;
        if (i >= n) return;                                       // M % 8 == 0 and N % 8 == 0: n, n_head are multiples of 4
        float4 s = *(const float4*)(part + i);
        for (int k = 1; k < splits; ++k) {
            const float4 v = *(const float4*)(part + (size_t)k * n + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (i < n_head) {
            const long r = i / N, c = i - r * N;
            st4_any(jb->out, jb->out_dt, (size_t)(r * jb->ldo + c), s);
        } else {
            *(float4*)(jb->out_tail + (i - n_head)) = s;
        }
    }
}

extern "C" int fabind_gemm_tn_multi(const FabindTnJob* jobs_dev, const FabindTnJob* jobs_host, int n_jobs, const int* wg_job_dev, int total_wg,
                                    const int* blk_job_dev, int total_blk, const void* zero_page, hipStream_t stream) {
    if (n_jobs <= 0) return 0;
    FB_REQUIRE(jobs_dev && jobs_host && wg_job_dev && blk_job_dev && zero_page && ((uintptr_t)zero_page % 16 == 0), "fabind_gemm_tn_multi: null table");
    int wg = 0, blk = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const FabindTnJob& b = jobs_host[j];
        FB_REQUIRE(b.M > 0 && b.N > 0 && b.E >= 0 && b.splits >= 1 && b.M % 8 == 0 && b.N % 8 == 0 && b.ldy % 8 == 0 && b.ldx % 8 == 0,
                   "fabind_gemm_tn_multi: M, N, ldy, ldx must be positive multiples of 8");
        FB_REQUIRE(((uintptr_t)b.Y % 16 == 0) && ((uintptr_t)b.X % 16 == 0) && ((uintptr_t)b.C_part % 16 == 0) && b.out, "fabind_gemm_tn_multi: alignment");
        FB_REQUIRE(b.ldo >= b.N && b.ldo % 4 == 0 && ((uintptr_t)b.out % 8 == 0) && (!b.with_colsum || (b.out_tail && (uintptr_t)b.out_tail % 16 == 0)),
                   "fabind_gemm_tn_multi: output layout");
        const int nt = ((b.M + 255) / 256) * ((b.N + 255) / 256);
        FB_REQUIRE(b.n_tiles == nt && b.e_per == ((b.E + b.splits - 1) / b.splits + 31) / 32 * 32 && b.wg0 == wg && b.wg0 % 8 == 0 &&
                   b.n_wg == nt * ((b.splits + 7) / 8 * 8) && b.blk0 == blk, "fabind_gemm_tn_multi: inconsistent job table");
        const long n = (long)b.M * b.N + (b.with_colsum ? b.M : 0);
        FB_REQUIRE(b.n_blk == (int)((n + 4095) / 4096), "fabind_gemm_tn_multi: inconsistent reduction blocks");
        wg += b.n_wg; blk += b.n_blk;
    }
    FB_REQUIRE(wg == total_wg && blk == total_blk, "fabind_gemm_tn_multi: work-group totals");
    constexpr int NSTAGE = 4, NW = 8, TN_ = 256;
    const size_t lds = (size_t)NSTAGE * 32 * (256 + TN_) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<NSTAGE, NW, TN_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<NSTAGE, NW, TN_, true>), dim3(total_wg), dim3(NW * 64), lds, stream, (const bf16_t*)nullptr, 0,
                       (const bf16_t*)nullptr, 0, (float*)nullptr, 0, 0, 0, 0, (const bf16_t*)zero_page, (const int*)nullptr, 0, 0, 0, 0, jobs_dev,
                       wg_job_dev);
    FB_CHECK_LAUNCH();
    hipLaunchKernelGGL(tn_multi_sum_kernel, dim3(total_blk), dim3(256), 0, stream, jobs_dev, blk_job_dev);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_gemm_tn(const void* Y, int ldy, const void* X, int ldx, float* C_part, int M, int N, int E, int splits,
                              const void* zero_page, const int* groups, int n_groups, int with_colsum, hipStream_t stream) {
    FB_REQUIRE(!(groups && with_colsum), "fabind_gemm_tn: grouped launches carry no column sums");
    FB_REQUIRE(M % 8 == 0 && N % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0, "fabind_gemm_tn: M, N, ldy, ldx must be multiples of 8");
    FB_REQUIRE(((uintptr_t)Y % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)zero_page % 16 == 0), "fabind_gemm_tn: alignment");
    FB_REQUIRE(!(groups && splits != 1), "fabind_gemm_tn: grouped launches are not split");
    if (M <= 0 || N <= 0 || splits <= 0) return 0;
    if (g_tn_waves == 4) tn_launch<3, 4, 128>(Y, ldy, X, ldx, C_part, M, N, E, splits, zero_page, groups, n_groups, with_colsum, stream);
    else if (g_tn_waves == 8) tn_launch<3, 8, 128>(Y, ldy, X, ldx, C_part, M, N, E, splits, zero_page, groups, n_groups, with_colsum, stream);
    else if (g_tn_waves == 20) tn_launch<5, 8, 256>(Y, ldy, X, ldx, C_part, M, N, E, splits, zero_page, groups, n_groups, with_colsum, stream);
    else tn_launch<4, 8, 256>(Y, ldy, X, ldx, C_part, M, N, E, splits, zero_page, groups, n_groups, with_colsum, stream);
    FB_CHECK_LAUNCH();
    return 0;
}

static int g_gemm_persist = 0;   // measured slower than 2 work-groups/CU of the tile-per-block kernel (see tools/gemm_bench.py)
extern "C" void fabind_gemm_set_persistent(int on) { g_gemm_persist = on; }
// 13 = 256x128 tile, BK = 32, 3-stage ring (cfg 3, best tile of the measured set: tools/gemm_bench.py) compiled under a 4-waves-per-SIMD
// launch bound: 128 VGPRs + 336 B scratch instead of 213 VGPRs, so TWO 8-wave work-groups are resident per CU (LDS 2 x 76 KB) and one tile's
// pipeline fill / epilogue overlaps the other's main loop: 16-26 % faster on every node-level shape, bitwise-equal results
// (tools/probes/gemm_node_occ.py)
static int g_gemm_cfg = 13;
static int g_gemm_small_m = 100;   // launches with fewer 256x128 tiles than this use 128x128 tiles (0 = never): 2,624 ligand rows 15.4 -> 12.4 us,
                                   // but 9,088 rows x 512 columns (144 tiles) 17.7 -> 23.5 us (tools/probes/gemm_small_m.py)
extern "C" void fabind_gemm_set_small_m(int n) { g_gemm_small_m = n; }
extern "C" void fabind_gemm_set_config(int cfg) { g_gemm_cfg = cfg; }

template <int WM, int BK_, int NSTAGE, int MINW = 1>
static int launch_pipe(const FabindGemmArgs& p, int maxM, int maxN, hipStream_t stream, int row_tile0 = 0) {
    constexpr int BM_ = WM * 64;
    const size_t lds = lds_tile_bytes<WM, BK_, NSTAGE>() + 2 * BM_ * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pipe_kernel<WM, BK_, NSTAGE, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM_ - 1) / BM_ - row_tile0, p.groups ? p.n_groups : 1);
    if (grid.y == 0) return 0;
    if (!p.groups) grid = dim3(grid.x * grid.y, p.k_splits > 1 ? p.k_splits : 1, 1);
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<WM, BK_, NSTAGE, MINW>), grid, dim3(WM * 128), lds, stream, p, row_tile0);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") NT GEMM: fp32 A (optionally [A | A2]) and fp32 W at (nearly) fp32 accuracy on the bf16 matrix cores.
// Every operand element x is split WHILE IT IS STAGED into hi = bf16_rn(x) and lo = bf16_rn(x - hi) -- 16 significand bits
// together -- and a product term is three v_mfma_f32_16x16x32_bf16: lo*hi + hi*lo + hi*hi, fp32 accumulate (the lo*lo term,
// 2^-18 of the product, is dropped).  Against the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate) that is
// 3/16 of the matrix-core time; the reference computes these contractions in fp32 (FABind/fabind/models/egnn.py:68-144,
// model_utils.py:83-131), and this is the mode that meets the 1e-4 A gate at speed.
//   * tile (64 WM) x 128 x 32, WM x 2 waves of 64 x 64; operands go global -> registers (fp32, 16-B loads) -> split -> two bf16
//     LDS planes (hi | lo) per operand; two LDS stages and a register prefetch of the tile after next: ONE barrier per k-step,
//     and the splitting VALU work of tile k+1 sits next to the MFMAs of tile k in every wave;
//   * rows are 32 bf16 + 8 pad (80 B): the 16 lanes of a fragment read hit 16 different 16-byte bank groups;
//   * epilogues, ragged groups, split-K and the XCD-aware tile order are those of the bf16 kernels.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void x3_split8(const Vec8& x, uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = pack2_bf16(x.v[2 * q], x.v[2 * q + 1]);
        l[q] = pack2_bf16(x.v[2 * q] - __uint_as_float(h[q] << 16), x.v[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int WM> __host__ __device__ constexpr size_t x3_lds_bytes() {
    return (size_t)2 * 2 * (WM * 64 + BN) * (BK + 8) * 2;          // [stage][hi | lo][rows][40] bf16
}

template <int WM, int MINW, bool PRO>
__global__ __launch_bounds__(WM * 128, MINW) void gemm_x3_kernel(FabindGemmArgs p) {
    constexpr int BM_ = WM * 64, NT = WM * 128, LS = BK + 8;
    constexpr int ROWS = BM_ + BN, PLANE = ROWS * LS;             // A rows then W rows; one plane = hi or lo of one stage
    static_assert((BM_ * 8) % NT == 0 && (BN * 8) % NT == 0, "staging divides evenly");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sT = (bf16_t*)smem;                                   // [2][2][ROWS][LS]
    // epilogue scratch lives in the (finished) stages: the bf16 / fp32 staging tile first, the row-dot partials behind it
    float* sDot = (float*)(smem + x3_lds_bytes<WM>() - 2 * BM_ * sizeof(float));

    int M = p.M, N = p.N, ldc = p.ldc;
    long a_row0 = 0, w_row0 = 0, c_off = 0;
    int Kg = p.K, lda_g = p.lda, K1g = p.K1;
    long a_eoff = 0, w_eoff = 0;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.groups && p.groups_ext) {
        const int* g = p.groups + (size_t)blockIdx.z * 16;
        M = g[0]; N = g[1]; Kg = g[2]; lda_g = g[3];
        a_eoff = (long)(unsigned)g[4] | ((long)g[5] << 32);
        w_eoff = (long)(unsigned)g[6] | ((long)g[7] << 32);
        c_off = (long)(unsigned)g[8] | ((long)g[9] << 32);
        ldc = g[10];
        K1g = Kg;
    } else if (p.groups) {
        const int* g = p.groups + (size_t)blockIdx.z * 8;
        a_row0 = g[0]; M = g[1]; w_row0 = g[2]; N = g[3];
        c_off = (long)(unsigned)g[4] | ((long)g[5] << 32);
        if (g[6] > 0) ldc = g[6];
    } else {
        // 1-D launch, XCD-aware tile order (see gemm_bf16_pipe_kernel): the N-tiles of one M-panel share an XCD's L2 copy of the A panel
        const int nbx = (N + BN - 1) / BN, total = gridDim.x;
        const int L = blockIdx.x, q = total / 8, r = total % 8, xcd = L % 8;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + L / 8;
        bx = t % nbx; by = t / nbx;
    }
    const int m0 = by * BM_, n0 = bx * BN;
    if (m0 >= M || n0 >= N) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* A = (const float*)p.A + a_row0 * lda_g + a_eoff;
    const float* A2 = p.A2 ? (const float*)p.A2 + a_row0 * p.lda2 : nullptr;
    const float* W = (const float*)p.W + w_row0 * p.ldw + w_eoff;
    int K = Kg;
    const int K1 = K1g;
    int kbeg = 0;
    if (!p.groups && p.k_splits > 1) {      // split-K over blockIdx.y (non-grouped launches are 1-D in x)
        const int per = ((p.K / p.k_splits + BK - 1) / BK) * BK;
        kbeg = blockIdx.y * per;
        K = min(p.K, kbeg + per);
        c_off = (long)blockIdx.y * M * ldc;
    }
    if (p.groups_ext) { a_row0 = 0; w_row0 = 0; }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Staging: a thread owns NA + NB chunks of 8 consecutive k (two 16-byte loads each) per k-tile.  TWO register sets alternate, so
    // a tile's loads are requested two k-steps before its split -- one k-step (~0.3 us of MFMA work) is less than an HBM round trip,
    // and with one or two 4- / 8-wave work-groups per CU nothing else covers it.  Every load is UNCONDITIONAL (row and k indices are
    // clamped, out-of-range values are zeroed by a select when they are split): the first form put the loads under per-lane bounds
    // branches and the prologue activation under a run-time switch, and the compiler answered with `s_waitcnt vmcnt(0)` at every join
    // -- 1,764 of them in the kernel, the prefetch distance collapsed to nothing (ISA of round 3's first build; 245 us at
    // M = 98,688, N = K = 512).  PRO: prologue activation as a compile-time variant.
    // thread -> (row, 4-float piece) of a k-tile: EIGHT consecutive lanes read the 128 contiguous bytes a row contributes to a k-tile, so a
    // wave's load instruction covers 8 whole 128-byte lines.  Three staging maps were measured at M = 98,688, N = K = 512, bias -> fp32
    // (profiles/r03_gemm_x3_pmc.txt): four lanes per row with two 16-byte loads per lane 245-262 us; sixteen consecutive lanes on sixteen
    // rows (conflict-free plane writes: LDS conflict cycles 2.6e7 -> 1.3e7, VALU instructions 4.1e7 -> 3.5e7) 301 us -- 64 quarter-used
    // lines per load instruction; this one 257 us (128-row tile) / 243 us (256-row tile).  None moves the launch: per SIMD it issues
    // ~148 k cycles of MFMA and ~146 k cycles of VALU (the split: five instructions per pair of elements) in ~455 k, LDS active 46 %.
    constexpr int NA4 = BM_ * 8 / NT, NB4 = BN * 8 / NT;          // float4 pieces per thread and k-tile
    const int lr = tid >> 3, lc = (tid & 7) * 4;                 // this thread's row (+ NT/8 per piece) and k offset in a k-tile
    const float* arow[NA4];
    const float* a2row[NA4];
    const float* wrow[NB4];
    bool aok[NA4], wok[NB4];
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
        const int gm = m0 + lr + (NT / 8) * i;
        aok[i] = gm < M;
        const size_t gmc = (size_t)min(gm, M - 1);
        arow[i] = A + gmc * lda_g;
        a2row[i] = A2 ? A2 + gmc * p.lda2 - K1 : arow[i];         // indexed with the global k as well
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
        const int gn = n0 + lr + (NT / 8) * i;
        wok[i] = gn < N;
        wrow[i] = W + (size_t)min(gn, N - 1) * p.ldw;
    }
    auto fetch = [&](float4 (&ra)[NA4], float4 (&rb)[NB4], int k0) {
        const int gk = k0 + lc;
        const bool inside = gk < K;                               // (K % 8 == 0: a 4-float piece is inside or outside as a whole)
        const bool second = inside && k0 >= K1;                   // k0 >= K1 is uniform: K1 is a multiple of the k-tile when A2 is given
        const int gkc = inside ? gk : min(lc, K1 - 4);            // a piece past the end re-reads a valid one of the first operand
#pragma unroll
        for (int i = 0; i < NA4; ++i) ra[i] = *(const float4*)((second ? a2row[i] : arow[i]) + gkc);
#pragma unroll
        for (int i = 0; i < NB4; ++i) rb[i] = *(const float4*)(wrow[i] + (inside ? gk : min(lc, K - 4)));
    };
    const bool tile_inside = (m0 + BM_ <= M) && (n0 + BN <= N) && (K % BK == 0);      // uniform: no element of any k-tile needs zeroing
    auto put4 = [&](bf16_t* hi, bf16_t* lo, int o, float4 x, bool ok) {
        if (PRO || !tile_inside) {
            float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PRO) v[q] = apply_act(v[q], p.act_pro);
                v[q] = ok ? v[q] : 0.f;
            }
            x = make_float4(v[0], v[1], v[2], v[3]);
        }
        const uint32_t h0 = pack2_bf16(x.x, x.y), h1 = pack2_bf16(x.z, x.w);
        const uint32_t l0 = pack2_bf16(x.x - __uint_as_float(h0 << 16), x.y - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t l1 = pack2_bf16(x.z - __uint_as_float(h1 << 16), x.w - __uint_as_float(h1 & 0xffff0000u));
        *(uint2*)&hi[o] = make_uint2(h0, h1);
        *(uint2*)&lo[o] = make_uint2(l0, l1);
    };
    auto split_store = [&](float4 (&ra)[NA4], float4 (&rb)[NB4], int st, int k0) {
        bf16_t* hi = sT + (size_t)st * 2 * PLANE;
        bf16_t* lo = hi + PLANE;
        const bool kok = tile_inside || k0 + lc < K;
#pragma unroll
        for (int i = 0; i < NA4; ++i) put4(hi, lo, (lr + (NT / 8) * i) * LS + lc, ra[i], kok && aok[i]);
#pragma unroll
        for (int i = 0; i < NB4; ++i) put4(hi, lo, (BM_ + lr + (NT / 8) * i) * LS + lc, rb[i], kok && wok[i]);
    };
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    auto compute = [&](int st) {
        const bf16_t* hi = sT + (size_t)st * 2 * PLANE;
        const bf16_t* lo = hi + PLANE;
        bf16x8_t bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = (BM_ + wn * 64 + j * 16 + fr) * LS + fk;
            bh[j] = *(const bf16x8_t*)&hi[o];
            bl[j] = *(const bf16x8_t*)&lo[o];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (wm * 64 + i * 16 + fr) * LS + fk;
            const bf16x8_t ah = *(const bf16x8_t*)&hi[o], al = *(const bf16x8_t*)&lo[o];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (K - kbeg + BK - 1) / BK;
    float4 ra0[NA4], rb0[NB4], ra1[NA4], rb1[NB4];
    // invariant at the top of step kt (even): stage kt & 1 holds tile kt; set 1 holds tile kt+1 (in flight), set 0 tile kt+2 (just requested)
    // (loads past the last k-tile are issued anyway -- clamped addresses, values never used -- so that the loop body has no branch
    //  around a load and the waits count instructions)
    if (nk > 0) {
        fetch(ra0, rb0, kbeg);
        fetch(ra1, rb1, kbeg + BK);
        split_store(ra0, rb0, 0, kbeg);
        fetch(ra0, rb0, kbeg + 2 * BK);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) split_store(ra1, rb1, 1, kbeg + (kt + 1) * BK);   // (stage 1 was last read in step kt-1, before its barrier)
        fetch(ra1, rb1, kbeg + (kt + 3) * BK);
        compute(0);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) split_store(ra0, rb0, 0, kbeg + (kt + 2) * BK);
            fetch(ra0, rb0, kbeg + (kt + 4) * BK);
            compute(1);
            __syncthreads();
        }
    }
    if (p.epi_fast && gemm_epilogue_dispatch<BM_, false>(p, acc, sDot, sT, M, N, ldc, m0, n0, (int)(x3_lds_bytes<WM>() - 2 * BM_ * sizeof(float)))) return;
    gemm_epilogue<BM_>(p, acc, sDot, sT, M, N, ldc, a_row0, w_row0, c_off, m0, n0);
}

// development probe: resident work-groups per CU of the split-bf16 kernel as the runtime computes it (its two LDS stages are exactly
// 80 KiB at the 128-row tile: do two fit beside each other in 160 KiB?)
extern "C" int fabind_gemm_x3_occupancy(int wm) {
    int n = -1;
    if (wm == 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_x3_kernel<4, 2, false>, 512, x3_lds_bytes<4>());
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_x3_kernel<2, 2, false>, 256, x3_lds_bytes<2>());
    return n;
}

static int g_x3_wm = 2;      // tile height of the split-bf16 kernel in units of 64 rows (2: two 4-wave work-groups per CU, 4: one 8-wave)
extern "C" void fabind_gemm_set_x3_tile(int wm) { g_x3_wm = (wm == 4) ? 4 : 2; }

template <int WM, int MINW, bool PRO>
static int launch_x3(const FabindGemmArgs& p, int maxM, int maxN, hipStream_t stream) {
    constexpr int BM_ = WM * 64;
    const size_t lds = x3_lds_bytes<WM>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<WM, MINW, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM_ - 1) / BM_, p.groups ? p.n_groups : 1);
    if (!p.groups) grid = dim3(grid.x * grid.y, p.k_splits > 1 ? p.k_splits : 1, 1);
    hipLaunchKernelGGL((gemm_x3_kernel<WM, MINW, PRO>), grid, dim3(WM * 128), lds, stream, p);
    return 0;
}

extern "C" int fabind_gemm(const FabindGemmArgs* args, hipStream_t stream) {
    FabindGemmArgs p = *args;
    FB_REQUIRE(p.groups_ext || p.K % 8 == 0, "fabind_gemm: K must be a multiple of 8");
    FB_REQUIRE(p.A2 == nullptr || (p.K1 % BK == 0), "fabind_gemm: K1 must be a multiple of 32 when A2 is given");
    if (p.A2 == nullptr) p.K1 = p.K;
    FB_REQUIRE(p.lda % 4 == 0 && p.ldw % 4 == 0, "fabind_gemm: lda/ldw must be multiples of 4");
    FB_REQUIRE(!(p.a_dtype == FB_DT_BF16 && (p.lda % 8 != 0)), "fabind_gemm: bf16 A needs lda % 8 == 0");
    FB_REQUIRE(!(p.w_dtype == FB_DT_BF16 && (p.ldw % 8 != 0)), "fabind_gemm: bf16 W needs ldw % 8 == 0");
    p.epi_fast = 0;
    FB_REQUIRE(p.p_drop >= 0.f && p.p_drop < 1.f, "fabind_gemm: p_drop in [0, 1)");
    const bool foldq = p.row_mu != nullptr;
    FB_REQUIRE(!foldq || (p.row_rs != nullptr && p.col_c != nullptr), "fabind_gemm: row_mu needs row_rs and col_c");
    const bool drop = p.p_drop > 0.f;    /* dropout: generic epilogue, or the fast ones without a second / pre-activation tile */
    /* C += A W^T on a plain fp32 C is the residual epilogue with R = C: every element is read and written by the same lane, so
       the fast fp32 epilogue (row-contiguous 16-B accesses) serves it (the generic one stores element by element) */
    if (p.accumulate && !p.R && !p.r_index && !p.groups && p.k_splits <= 1 && p.C != nullptr && p.c_dtype == FB_DT_F32) {
        p.R = p.C; p.ldr = p.ldc; p.accumulate = 0; p.r_dtype = FB_DT_F32;
    }
    if (!p.R) p.r_dtype = FB_DT_F32;
    FB_REQUIRE(p.r_dtype == FB_DT_F32 || p.r_dtype == FB_DT_BF16, "fabind_gemm: r_dtype");
    const bool r32 = p.r_dtype == FB_DT_F32;
    if (!p.groups && p.k_splits <= 1 && p.alpha == 1.0f && !p.aux && !p.accumulate && !p.r_index && p.C != nullptr && r32 &&
        p.c_dtype == FB_DT_F32 && p.act_epi == FB_ACT_NONE && !p.dotvec && !p.C2) {
        p.epi_fast = p.R ? 10 : 9;                /* (with or without epilogue dropout ahead of the residual) */
    } else if (!drop && !p.groups && p.k_splits <= 1 && p.alpha == 1.0f && !p.aux && !p.accumulate && p.R && p.r_index && p.C && r32 &&
               p.c_dtype == FB_DT_BF16 && p.act_epi == FB_ACT_NONE && !p.dotvec && !p.C2) {
        p.epi_fast = 11;
    } else if (!drop && !p.groups && p.k_splits <= 1 && p.alpha == 1.0f && !p.aux && !p.accumulate && p.R && !p.r_index && p.C && !r32 &&
               p.c_dtype == FB_DT_BF16 && p.act_epi == FB_ACT_NONE && !p.dotvec && !p.C2 && !p.store_preact) {
        p.epi_fast = 16;     /* bf16 C = A W^T + bias + bf16 R: the residual stream of the FABind+ pair embedding */
    } else if (!drop && !p.groups && p.k_splits <= 1 && p.aux && p.aux_dtype == FB_DT_BF16 &&      /* (any alpha: applied with act') */
               (p.dact_epi == FB_ACT_RELU || p.dact_epi == FB_ACT_STORED_DERIV) && !p.R && !p.accumulate && !p.r_index && p.C &&
               p.c_dtype == FB_DT_BF16 && p.act_epi == FB_ACT_NONE && !p.dotvec && !p.C2 && !p.bias && !p.store_preact) {
        p.epi_fast = 14;     /* C = (A W^T) * act'(aux): the activation adjoint of an MLP inside its input-gradient GEMM */
    } else
    if (!p.groups && p.k_splits <= 1 && p.alpha == 1.0f && !p.aux && !p.R && !p.accumulate && !p.r_index &&
        (p.C == nullptr || p.c_dtype == FB_DT_BF16) && (p.C2 == nullptr || p.C != nullptr) &&
        !(drop && (p.C2 != nullptr || p.store_preact))) {
        const bool hc = p.C != nullptr, hc2 = p.C2 != nullptr, hd = p.dotvec != nullptr;
        const bool pre = p.store_preact != 0;
        if (p.act_epi == FB_ACT_NONE && hc && !hc2 && !hd) p.epi_fast = 1;
        else if (p.act_epi == FB_ACT_SILU && hc && !hc2 && !hd && !pre) p.epi_fast = 2;
        else if (p.act_epi == FB_ACT_SILU && hc && hc2 && !hd && !pre) p.epi_fast = 3;
        else if (p.act_epi == FB_ACT_SILU && !hc && !hc2 && hd) p.epi_fast = 4;
        else if (p.act_epi == FB_ACT_RELU && hc && !hc2 && !hd && !pre) p.epi_fast = 5;
        else if (p.act_epi == FB_ACT_RELU && !hc && !hc2 && hd) p.epi_fast = 6;
        else if (p.act_epi == FB_ACT_SILU && hc && !hc2 && hd && pre) p.epi_fast = 7;
        else if (p.act_epi == FB_ACT_RELU && hc && !hc2 && hd && pre) p.epi_fast = 8;
        else if (p.act_epi == FB_ACT_RELU && hc && !hc2 && hd && !pre) p.epi_fast = 15;
    }
    if (p.epi_fast == 0 && !drop && !p.groups && p.k_splits <= 1 && p.alpha == 1.0f && !p.accumulate && !p.r_index && !p.R && p.C != nullptr &&
        p.c_dtype == FB_DT_F32 && !p.dotvec && !p.store_preact && (!p.C16 || !p.aux) && !foldq) {
        /* fp32 C with an activation (+ stored derivative) or with the activation adjoint through an fp32 aux tile */
        if (p.aux && p.aux_dtype == FB_DT_F32 && p.act_epi == FB_ACT_NONE && !p.C2 && (p.dact_epi == FB_ACT_RELU || p.dact_epi == FB_ACT_STORED_DERIV))
            p.epi_fast = p.dact_epi == FB_ACT_RELU ? 20 : 21;
        else if (!p.aux && p.act_epi == FB_ACT_SILU) p.epi_fast = p.C2 ? 22 : 23;
        else if (!p.aux && p.act_epi == FB_ACT_RELU) p.epi_fast = p.C2 ? 25 : 24;
    }
    if (foldq) {
        FB_REQUIRE((p.epi_fast == 5 || p.epi_fast == 6) && p.C2 == nullptr && p.a_dtype == FB_DT_BF16 && p.w_dtype == FB_DT_BF16 &&
                       p.K % 64 == 0,
                   "fabind_gemm: the row_mu / row_rs fold exists for relu epilogues writing bf16 C or a row-dot only (bf16 operands, "
                   "K % 64 == 0, no residual / second tile / split-K)");
        p.epi_fast += 7;        /* 5 -> 12, 6 -> 13: the FOLD instantiations */
    }
    FB_REQUIRE(p.C16 == nullptr || (p.C != nullptr && p.c_dtype == FB_DT_F32 && !p.groups && p.k_splits <= 1 && !p.store_preact &&
                                    p.ldc16 % 4 == 0 && ((uintptr_t)p.C16 & 7) == 0),
               "fabind_gemm: C16 (bf16 copy of C) needs a plain fp32 C (no groups / split-K / stored pre-activation), ldc16 % 4 == 0");
    int maxM = p.groups ? p.max_m : p.M, maxN = p.groups ? p.max_n : p.N;
    if (maxM <= 0 || maxN <= 0) return 0;
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, p.groups ? p.n_groups : (p.k_splits > 1 ? p.k_splits : 1));
    if (p.k_splits > 1) {
        FB_REQUIRE(!p.groups && p.c_dtype == FB_DT_F32 && !p.bias && !p.R && !p.dotvec && !p.aux && !p.accumulate &&
                       p.act_epi == FB_ACT_NONE && p.A2 == nullptr,
                   "fabind_gemm: split-K needs a plain fp32 epilogue");
        FB_REQUIRE(p.K % 64 == 0, "fabind_gemm: split-K needs K % 64 == 0");
    }
    if (p.w_dtype == FB_DT_F32 && p.split3) {
        FB_REQUIRE(p.a_dtype == FB_DT_F32, "fabind_gemm: the split-bf16 contraction takes fp32 A and fp32 W");
        FB_REQUIRE(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0) && (p.A2 == nullptr || ((uintptr_t)p.A2 % 16 == 0 && p.lda2 % 4 == 0)),
                   "fabind_gemm: split-bf16 operands must be 16-byte aligned");
        FB_REQUIRE(p.lda % 4 == 0 && p.ldw % 4 == 0, "fabind_gemm: split-bf16 operands are read as 16-byte rows (lda % 4 == 0, ldw % 4 == 0)");
        if (p.act_pro != FB_ACT_NONE) launch_x3<2, 2, true>(p, maxM, maxN, stream);
        else if (g_x3_wm == 4) launch_x3<4, 2, false>(p, maxM, maxN, stream);
        else launch_x3<2, 2, false>(p, maxM, maxN, stream);
    } else if (p.w_dtype == FB_DT_F32) {
        FB_REQUIRE(p.a_dtype == FB_DT_F32, "fabind_gemm: fp32 MMA needs fp32 A");
        hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, dim3(256), 0, stream, p);
    } else if (p.a_dtype == FB_DT_F32) {
        hipLaunchKernelGGL((gemm_nt_kernel<float, bf16_t>), grid, dim3(256), 0, stream, p);
    } else if (!p.groups_ext && p.act_pro == FB_ACT_NONE && p.K % FBK == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 &&
               ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0) &&
               (p.A2 == nullptr || (g_gemm_cfg != 0 && p.K1 % FBK == 0 && p.lda2 % 8 == 0 && ((uintptr_t)p.A2 % 16 == 0)))) {
        const long n_tiles = (long)((maxM + 255) / 256) * ((maxN + BN - 1) / BN);
        if (g_gemm_persist && !p.groups && p.k_splits <= 1 && n_tiles >= 2048 && p.K <= 2048) {
            launch_persist<4, 32, 3>(p, stream);
            FB_CHECK_LAUNCH();
            return 0;
        }
        int cfg = (p.k_splits > 1 && g_gemm_cfg == 0) ? 3 : g_gemm_cfg;
        int row_tile0 = 0;
        /* few row tiles (ligand rows, pocket-sized batches): the 256-row tile leaves most CUs idle -- 128-row tiles double the work-group
           count; same k-order, bitwise-equal results (tools/probes/gemm_small_m.py) */
        if (cfg == 13 && !p.groups && g_gemm_small_m && n_tiles * (p.k_splits > 1 ? p.k_splits : 1) < g_gemm_small_m) cfg = 6;
        switch (cfg) {
            case 1: launch_pipe<2, 32, 4>(p, maxM, maxN, stream); break;
            case 2: launch_pipe<2, 64, 3>(p, maxM, maxN, stream); break;
            case 3: launch_pipe<4, 32, 3>(p, maxM, maxN, stream); break;
            case 4: launch_pipe<4, 64, 3>(p, maxM, maxN, stream); break;
            case 5: launch_pipe<4, 32, 4>(p, maxM, maxN, stream); break;
            case 6: launch_pipe<2, 32, 3>(p, maxM, maxN, stream); break;
            case 7: launch_pipe<4, 32, 2>(p, maxM, maxN, stream); break;
            case 8: launch_pipe<2, 32, 2>(p, maxM, maxN, stream); break;
            case 9: launch_pipe<4, 64, 2>(p, maxM, maxN, stream); break;
            case 13: launch_pipe<4, 32, 3, 4>(p, maxM, maxN, stream, row_tile0); break;   /* cfg 3 under a 4-waves-per-SIMD bound (default) */
            default: hipLaunchKernelGGL(gemm_bf16_glds_kernel, grid, dim3(256), 0, stream, p);
        }
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, p);
    }
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out[C,R] = act(in[R,C])^T   (used to turn dW = dY^T X into NT GEMMs; fused activation on read)
// ------------------------------------------------------------------------------------------------
// 64x64 tiles through LDS; 16 contiguous elements per thread on both the read and the write side
__global__ __launch_bounds__(256) void transpose_act_kernel(const void* in, int in_dt, int ldi, void* out, int out_dt,
                                                            int ldo, int R, int C, int act) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;
    {
        const int r = r0 + tr;
        const bool vec = (r < R) && (c0 + tc + 16 <= C) && (ldi % 4 == 0);
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = ld4_any(in, in_dt, (size_t)r * ldi + c0 + tc + q * 4);
                tile[tr][tc + q * 4 + 0] = apply_act(v.x, act); tile[tr][tc + q * 4 + 1] = apply_act(v.y, act);
                tile[tr][tc + q * 4 + 2] = apply_act(v.z, act); tile[tr][tc + q * 4 + 3] = apply_act(v.w, act);
            }
        } else {
            for (int q = 0; q < 16; ++q) {
                const int c = c0 + tc + q;
                tile[tr][tc + q] = (r < R && c < C) ? apply_act(ld_any(in, in_dt, (size_t)r * ldi + c), act) : 0.f;
            }
        }
    }
    __syncthreads();
    {
        const int c = c0 + tr;                 // output row
        if (c >= C) return;
        const bool vec = (r0 + tc + 16 <= R) && (ldo % 4 == 0);
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                st4_any(out, out_dt, (size_t)c * ldo + r0 + tc + q * 4,
                        make_float4(tile[tc + q * 4 + 0][tr], tile[tc + q * 4 + 1][tr], tile[tc + q * 4 + 2][tr],
                                    tile[tc + q * 4 + 3][tr]));
        } else {
            for (int q = 0; q < 16; ++q) {
                const int r = r0 + tc + q;
                if (r < R) st_any(out, out_dt, (size_t)c * ldo + r, tile[tc + q][tr]);
            }
        }
    }
}

extern "C" int fabind_transpose_act(const void* in, int in_dt, int ldi, void* out, int out_dt, int ldo, int R, int C,
                                    int act, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((C + 63) / 64, (R + 63) / 64);
    hipLaunchKernelGGL(transpose_act_kernel, grid, dim3(256), 0, stream, in, in_dt, ldi, out, out_dt, ldo, R, C, act);
    FB_CHECK_LAUNCH();
    return 0;
}

// column sums of a [R,C] matrix (bias gradients): deterministic two-pass (chunk partials, then fixed-order sum)
__global__ __launch_bounds__(256) void colsum_part_kernel(const void* in, int in_dt, int ldi, float* scratch, int R, int C,
                                                          int rows_per) {
    // one block = 256 columns (64 lanes x 4 consecutive columns) x 4 row-lanes; fixed-order combine in LDS
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vec = (c + 3 < C) && (ldi % 4 == 0);
    if (vec) {
        int r = r0 + q;
        for (; r + 12 < r1; r += 16) {          // 4 independent loads in flight per lane
            float4 v0 = ld4_any(in, in_dt, (size_t)r * ldi + c), v1 = ld4_any(in, in_dt, (size_t)(r + 4) * ldi + c);
            float4 v2 = ld4_any(in, in_dt, (size_t)(r + 8) * ldi + c), v3 = ld4_any(in, in_dt, (size_t)(r + 12) * ldi + c);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; r < r1; r += 4) {
            float4 v = ld4_any(in, in_dt, (size_t)r * ldi + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    } else if (c < C) {
        for (int r = r0 + q; r < r1; r += 4) {
            s.x += ld_any(in, in_dt, (size_t)r * ldi + c);
            if (c + 1 < C) s.y += ld_any(in, in_dt, (size_t)r * ldi + c + 1);
            if (c + 2 < C) s.z += ld_any(in, in_dt, (size_t)r * ldi + c + 2);
            if (c + 3 < C) s.w += ld_any(in, in_dt, (size_t)r * ldi + c + 3);
        }
    }
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        float4 a = part[0][lane], b = part[1][lane], d = part[2][lane], e = part[3][lane];
        float o[4] = {(a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w)};
        for (int k = 0; k < 4 && c + k < C; ++k) scratch[(size_t)blockIdx.y * C + c + k] = o[k];
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* scratch, float* out, int C, int nchunk, int accumulate) {
    // 64 columns per block, the chunk rows split over 4 row-lanes, fixed-order combine
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C)
        for (int k = q; k < nchunk; k += 4) s += scratch[(size_t)k * C + c];
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// out[i] = sum_s part[s][i] in fixed split order, written as fp32 or bf16: the reduction of a split contraction's partial results
// straight into the dtype its consumer wants (a weight gradient that autograd carries as bf16 needed a cast kernel after the sum).
__global__ __launch_bounds__(256) void split_sum_kernel(const float* __restrict__ part, int splits, long n, void* out, int out_dt,
                                                        long n_head, float* out_tail) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;                                           // n % 4 == 0, n_head % 4 == 0
    float4 s = *(const float4*)(part + i);
    for (int k = 1; k < splits; ++k) {
        const float4 v = *(const float4*)(part + (size_t)k * n + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (i < n_head) st4_any(out, out_dt, (size_t)i, s);
    else *(float4*)(out_tail + (i - n_head)) = s;                 // the tail (a bias gradient behind a weight gradient) stays fp32
}

extern "C" int fabind_split_sum(const float* part, int splits, long n, void* out, int out_dt, long n_tail, float* out_tail,
                                hipStream_t stream) {
    if (n <= 0 || splits <= 0) return 0;
    FB_REQUIRE(n % 4 == 0 && ((uintptr_t)part & 15) == 0 && ((uintptr_t)out & 7) == 0, "fabind_split_sum: n % 4 == 0, aligned buffers");
    FB_REQUIRE(n_tail >= 0 && n_tail <= n && n_tail % 4 == 0 && (n_tail == 0 || (out_tail && ((uintptr_t)out_tail & 15) == 0)),
               "fabind_split_sum: n_tail % 4 == 0 with a 16-byte aligned fp32 out_tail");
    hipLaunchKernelGGL(split_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, part, splits, n, out, out_dt,
                       n - n_tail, out_tail);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_colsum(const void* in, int in_dt, int ldi, float* out, int R, int C, int accumulate,
                             float* scratch, int nchunk, hipStream_t stream) {
    if (C <= 0) return 0;
    FB_REQUIRE(nchunk >= 1, "fabind_colsum: nchunk >= 1");
    int rows_per = (R + nchunk - 1) / nchunk;
    if (rows_per < 1) rows_per = 1;
    if (nchunk == 1 && !accumulate) {      // a single chunk (split-K partial sums, R <= 256 rows): its "partial" IS the result
        hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 255) / 256, 1), dim3(256), 0, stream, in, in_dt, ldi, out, R, C, rows_per);
        FB_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 255) / 256, nchunk), dim3(256), 0, stream, in, in_dt, ldi, scratch, R,
                       C, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, scratch, out, C, nchunk,
                       accumulate);
    FB_CHECK_LAUNCH();
    return 0;
}
