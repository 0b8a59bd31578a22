// Protein<->ligand cross attention (RowAttentionBlock / Attention._attention, reference models/cross_att.py:118-134,
// models/model_utils.py:21-38,96-133) as MFMA bf16 tiles for gfx950 -- forward and both backward passes.
//
// The block is 4 heads x 32 channels, so one v_mfma_f32_16x16x32_bf16 contracts a whole head: a 16-key x 16-query
// score tile is ONE instruction per head.  All three kernels use the "swapped" orientation so that the softmax axis
// stays inside a lane and the probabilities feed the second contraction straight from the accumulator registers:
//
//   forward / pass Q (wave = 16 queries, loop over 32-key chunks):
//     S^T[key, query]  = K[key, :] . Q^T          A = K tile (LDS, row-major, swizzled),  B = Q fragment (registers)
//     lane (n = query, kq = lane>>4) then holds keys {t*16 + kq*4 + r}: bias add, exp, row statistics are lane-local
//     (+ two xor-shuffles across kq), and {p[t=0][0..3], p[t=1][0..3]} IS the B fragment of
//     O^T[dim, query] += V^T[dim, key'] . P^T[key', query]   with the LDS image of V^T stored in the permuted key order
//     key' = kq*8 + t*4 + r  (no cross-lane movement, no P round trip through LDS).
//   pass KV (wave = 16 keys, loop over 32-query chunks): the same with the roles of queries and keys exchanged.
//
// Scores, softmax statistics, the pair bias lin * sigmoid(gate) and every output stay fp32; bf16 only as MFMA operands
// (Q, K, V, P, dO, dS).  The fp32 VALU kernels (attn.hip / bwd.hip) remain the parity-mode path.  The work is bound by
// the pair-bias stream (32 B per (query, key) pair against 512 flop), not by the matrix cores: see DESIGN.md section 6.
#include "common.h"
#include "fabind_hip.h"

#define CM_KC 32                                   // keys (pass KV: queries) per chunk = K of the second contraction

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16x8_t cm_pack8(const float* f) {
    u32x4_t u;
    u[0] = pack2_bf16(f[0], f[1]); u[1] = pack2_bf16(f[2], f[3]); u[2] = pack2_bf16(f[4], f[5]); u[3] = pack2_bf16(f[6], f[7]);
    return *(bf16x8_t*)&u;
}
__device__ __forceinline__ float cm_round(float x) { return __uint_as_float(pack2_bf16(x, 0.f) << 16); }   // to bf16 and back
// sum of the eight bf16 values of a fragment (as fp32)
__device__ __forceinline__ float cm_sum8(const bf16x8_t f) {
    const u32x4_t u = *(const u32x4_t*)&f;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) t += __uint_as_float(u[k] << 16) + __uint_as_float(u[k] & 0xffff0000u);
    return t;
}
// position of chunk-local row `l` (0..31) in the permuted K order of the second contraction
__device__ __forceinline__ int cm_perm(int l) { return ((l >> 2) & 3) * 8 + (l >> 4) * 4 + (l & 3); }

// row-major image [32 rows][128 cols] bf16, 16-B chunks XOR-swizzled by the row: A operand of the first contraction
__device__ __forceinline__ void cm_store_rows(unsigned char* img, int row, int c4, float4 v) {
    *(uint2*)(img + row * 256 + ((((c4 >> 3) ^ (row & 15))) << 4) + (c4 & 4) * 2) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}
__device__ __forceinline__ bf16x8_t cm_load_rows(const unsigned char* img, int row, int chunk) {
    return *(const bf16x8_t*)(img + row * 256 + ((chunk ^ (row & 15)) << 4));
}
// transposed image [128 cols][32 permuted rows] bf16 (64 B per column, 16-B chunks XOR-swizzled by (col>>2)&3): A operand of the
// second contraction.  Staged by 256 threads from fp32 rows in global memory: thread = (column, pair of kq groups); the eight
// rows of one 16-B chunk are read as scalars (coalesced across the lanes' columns) and written with one ds_write_b128 --
// 16 lanes cover all 64 banks once (per-element 2-B stores at a 64-B column stride were 8-way conflicted).
__device__ __forceinline__ void cm_stage_cols(unsigned char* img, const float* __restrict__ src, int ld, long row0, int nrows,
                                              float mul, int tid) {
    const int col = tid & 127, g = tid >> 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = g * 2 + kk;
        float f[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + kq * 4 + r;
                f[t * 4 + r] = (row < nrows) ? src[(size_t)(row0 + row) * ld + col] * mul : 0.f;
            }
        *(bf16x8_t*)(img + col * 64 + ((kq ^ ((col >> 2) & 3)) << 4)) = cm_pack8(f);
    }
}
// row-major image of `nrows` (<= 32) fp32 rows [.,128] starting at row0, scaled by mul; rows beyond nrows are zero
__device__ __forceinline__ void cm_stage_rows(unsigned char* img, const float* __restrict__ src, int ld, long row0, int nrows,
                                              float mul, int tid) {
    for (int i = tid; i < CM_KC * 32; i += 256) {
        const int jr = i >> 5, c4 = (i & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jr < nrows) {
            v = *(const float4*)(src + (size_t)(row0 + jr) * ld + c4);
            v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        }
        cm_store_rows(img, jr, c4, v);
    }
}
__device__ __forceinline__ bf16x8_t cm_load_cols(const unsigned char* img, int col, int kq) {
    return *(const bf16x8_t*)(img + col * 64 + ((kq ^ ((col >> 2) & 3)) << 4));
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_attn_mfma_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ldkv,
                                                                  const float* __restrict__ gpre, int ldg,
                                                                  const float* __restrict__ bias, int bias_ld, int lin_col,
                                                                  int gate_col, const int* desc, float scale, float* out, int ldo,
                                                                  float* lse, int ksplit, float* part) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[CM_KC * 256];
    __shared__ __attribute__((aligned(16))) unsigned char sVt[128 * 64];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int qi = blockIdx.x * 64 + wave * 16 + n;
    const bool qvalid = qi < nq;
    const size_t qrow = (size_t)(q_off + (qvalid ? qi : 0));
    bf16x8_t bq[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float f[8];
        const float4 a = *(const float4*)(q + qrow * ldq + h * 32 + kq * 8), b = *(const float4*)(q + qrow * ldq + h * 32 + kq * 8 + 4);
        f[0] = a.x * scale; f[1] = a.y * scale; f[2] = a.z * scale; f[3] = a.w * scale;
        f[4] = b.x * scale; f[5] = b.y * scale; f[6] = b.z * scale; f[7] = b.w * scale;
        if (!qvalid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = 0.f;
        }
        bq[h] = cm_pack8(f);
    }
    f32x4_t o[4][2];
    float m[4], l[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        m[h] = -INFINITY; l[h] = 0.f;
        o[h][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o[h][1] = o[h][0];
    }
    int kb = 0, ke = nk;
    if (ksplit > 1) {
        const int per = ((nk + ksplit - 1) / ksplit + CM_KC - 1) / CM_KC * CM_KC;
        kb = blockIdx.z * per; ke = min(nk, kb + per);
    }
    for (int j0 = kb; j0 < ke; j0 += CM_KC) {
        __syncthreads();
        cm_stage_rows(sK, k, ldkv, (long)k_off + j0, ke - j0, 1.f, tid);
        cm_stage_cols(sVt, v, ldkv, (long)k_off + j0, ke - j0, 1.f, tid);
        __syncthreads();
        // pair bias of this lane's 8 (query, key) pairs: all four heads of a pair are one 32-B segment (issued before the MFMAs)
        float4 lin[2][4], gat[2][4];
        bool ok[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + t * 16 + kq * 4 + r;
                ok[t][r] = qvalid && j < ke;
                const size_t bi = (size_t)(pair_off + (long)(qvalid ? qi : 0) * sq + (long)min(j, ke - 1) * sk) * bias_ld;
                lin[t][r] = *(const float4*)(bias + bi + lin_col);
                gat[t][r] = *(const float4*)(bias + bi + gate_col);
            }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sK, t * 16 + n, h * 4 + kq), bq[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lv = ((const float*)&lin[t][r])[h], gv = ((const float*)&gat[t][r])[h];
                    const float sv = ok[t][r] ? s[t][r] + lv * sigmoid_f(gv) : -INFINITY;
                    s[t][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[h], mx);
            const float ms = (mn == -INFINITY) ? 0.f : mn;
            const float corr = __expf(m[h] - ms);
            float p[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[t * 4 + r] = __expf(s[t][r] - ms);
            const bf16x8_t pf = cm_pack8(p);
            // the normaliser sums the ROUNDED probabilities, i.e. exactly what P.V consumes: the output is an exact convex
            // combination of the value rows, and the backward (pass Q) can reproduce every weight bit for bit -- the row sums
            // of dS, which the pair-bias gradients are made of, then cancel to fp32 round-off instead of to 2^-9
            l[h] = l[h] * corr + cm_sum8(pf);
            m[h] = mn;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h][mi][r] *= corr;
                o[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sVt, h * 32 + mi * 16 + n, kq), pf, o[h][mi], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        l[h] += __shfl_xor(l[h], 16, 64);
        l[h] += __shfl_xor(l[h], 32, 64);
    }
    if (!qvalid) return;
    if (ksplit > 1) {      // un-normalised partial: [row][split][head][34] = o[32], m, l  (merged by cross_attn_mfma_combine_kernel)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float* pp = part + (((size_t)(q_off + qi) * ksplit + blockIdx.z) * 4 + h) * 34;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) pp[mi * 16 + kq * 4 + r] = o[h][mi][r];
            if (kq == 0) { pp[32] = m[h]; pp[33] = l[h]; }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float inv = l[h] > 0.f ? 1.f / l[h] : 0.f;     // no key at all (an empty ligand side): zeros, not 0 * inf
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int c = h * 32 + mi * 16 + kq * 4;
            const float4 g = *(const float4*)(gpre + qrow * ldg + c);
            *(float4*)(out + qrow * ldo + c) = make_float4(o[h][mi][0] * inv * sigmoid_f(g.x), o[h][mi][1] * inv * sigmoid_f(g.y),
                                                           o[h][mi][2] * inv * sigmoid_f(g.z), o[h][mi][3] * inv * sigmoid_f(g.w));
        }
        if (lse && kq == 0) lse[qrow * 4 + h] = m[h] + __logf(l[h]);
    }
}

__global__ __launch_bounds__(256) void cross_attn_mfma_combine_kernel(const float* __restrict__ part, int ksplit,
                                                                      const float* __restrict__ gpre, int ldg, int n_rows, float* out,
                                                                      int ldo, float* lse) {
    const int t = blockIdx.x * 256 + threadIdx.x;          // (row, head)
    if (t >= n_rows * 4) return;
    const int rowi = t >> 2, h = t & 3;
    float M = -INFINITY;
    for (int s_ = 0; s_ < ksplit; ++s_) M = fmaxf(M, part[(((size_t)rowi * ksplit + s_) * 4 + h) * 34 + 32]);
    float L = 0.f, o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    for (int s_ = 0; s_ < ksplit; ++s_) {
        const float* pp = part + (((size_t)rowi * ksplit + s_) * 4 + h) * 34;
        const float w = (pp[33] > 0.f) ? __expf(pp[32] - M) : 0.f;
        L += pp[33] * w;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] += pp[d] * w;
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d)
        out[(size_t)rowi * ldo + h * 32 + d] = o[d] * inv * sigmoid_f(gpre[(size_t)rowi * ldg + h * 32 + d]);
    if (lse) lse[(size_t)rowi * 4 + h] = M + __logf(L);
}

extern "C" int fabind_cross_attn_mfma_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* gpre,
                                          int ldg, const float* bias, int bias_ld, int lin_col, int gate_col, const int* desc, int B,
                                          int max_nq, float scale, float* out, int ldo, float* lse, int ksplit, float* part,
                                          int n_rows, hipStream_t stream) {
    FB_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0, "fabind_cross_attn_mfma_fwd: strides % 4");
    FB_REQUIRE(bias_ld % 4 == 0 && lin_col % 4 == 0 && gate_col % 4 == 0, "fabind_cross_attn_mfma_fwd: bias columns % 4");
    if (B <= 0 || max_nq <= 0) return 0;
    if (ksplit < 1) ksplit = 1;
    FB_REQUIRE(ksplit == 1 || part != nullptr, "fabind_cross_attn_mfma_fwd: ksplit > 1 needs the partials buffer");
    hipLaunchKernelGGL(cross_attn_mfma_fwd_kernel, dim3((max_nq + 63) / 64, B, ksplit), dim3(256), 0, stream, q, ldq, k, v, ldkv,
                       gpre, ldg, bias, bias_ld, lin_col, gate_col, desc, scale, out, ldo, lse, ksplit, part);
    if (ksplit > 1)
        hipLaunchKernelGGL(cross_attn_mfma_combine_kernel, dim3((n_rows * 4 + 255) / 256), dim3(256), 0, stream, part, ksplit, gpre,
                           ldg, n_rows, out, ldo, lse);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward, pass Q: d q, d gpre, d bias (lin, gate), and the per-query helpers dO = dout * sigmoid(gpre), D = dout . out
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_attn_mfma_bwd_q_kernel(const float* __restrict__ qg, int ldq, const float* __restrict__ kv,
                                                                    int ldkv, const float* __restrict__ bias, int bias_ld,
                                                                    int lin_col, int gate_col, const int* desc, float scale,
                                                                    const float* __restrict__ out, const float* lse,
                                                                    const float* __restrict__ dout, float* dqg, float* dbias,
                                                                    float* dO, float* Dv, int ksplit, float* part, int part_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[CM_KC * 256];
    __shared__ __attribute__((aligned(16))) unsigned char sV[CM_KC * 256];
    __shared__ __attribute__((aligned(16))) unsigned char sKt[128 * 64];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int qi = blockIdx.x * 64 + wave * 16 + n;
    const bool qvalid = qi < nq;
    const size_t qrow = (size_t)(q_off + (qvalid ? qi : 0));
    bf16x8_t bq[4], bdo[4];
    float D[4], L[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int c = h * 32 + kq * 8;
        float fq[8], fo[8], dsum = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 a = *(const float4*)(qg + qrow * ldq + c + half * 4);
            const float4 g = *(const float4*)(qg + qrow * ldq + 128 + c + half * 4);
            const float4 ov = *(const float4*)(out + qrow * 128 + c + half * 4);
            const float4 dv = *(const float4*)(dout + qrow * 128 + c + half * 4);
            const float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w}, o4[4] = {ov.x, ov.y, ov.z, ov.w},
                        d4[4] = {dv.x, dv.y, dv.z, dv.w};
            float dg[4], dov[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sg = sigmoid_f(gv[u]);
                fq[half * 4 + u] = qvalid ? av[u] * scale : 0.f;
                dov[u] = qvalid ? d4[u] * sg : 0.f;
                fo[half * 4 + u] = dov[u];
                // D = dO . o with dO as the contraction sees it (bf16) and o = out / gate the un-gated attention output: then
                // D = sum_j w_ij dP_ij holds to fp32 round-off for the weights w the forward used, and sum_j dS_ij = 0
                dsum += cm_round(dov[u]) * (sg > 1e-30f ? o4[u] * __builtin_amdgcn_rcpf(sg) : 0.f);
                dg[u] = d4[u] * o4[u] * (1.f - sg);
            }
            if (qvalid && blockIdx.z == 0) {
                *(float4*)(dqg + qrow * ldq + 128 + c + half * 4) = make_float4(dg[0], dg[1], dg[2], dg[3]);     // d gpre
                *(float4*)(dO + qrow * 128 + c + half * 4) = make_float4(dov[0], dov[1], dov[2], dov[3]);
            }
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        D[h] = dsum;
        L[h] = qvalid ? lse[qrow * 4 + h] : 0.f;
        if (qvalid && kq == 0 && blockIdx.z == 0) Dv[qrow * 4 + h] = dsum;
        bq[h] = cm_pack8(fq);
        bdo[h] = cm_pack8(fo);
    }
    f32x4_t dq[4][2];
    float mrun[4];                              // the forward's running row maximum, replayed chunk by chunk (same chunk order)
#pragma unroll
    for (int h = 0; h < 4; ++h) { dq[h][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[h][1] = dq[h][0]; mrun[h] = -INFINITY; }
    int kb = 0, ke = nk;
    if (ksplit > 1) {
        const int per = ((nk + ksplit - 1) / ksplit + CM_KC - 1) / CM_KC * CM_KC;
        kb = blockIdx.z * per; ke = min(nk, kb + per);
    }
    for (int j0 = kb; j0 < ke; j0 += CM_KC) {
        __syncthreads();
        cm_stage_rows(sK, kv, ldkv, (long)k_off + j0, ke - j0, 1.f, tid);
        cm_stage_rows(sV, kv + 128, ldkv, (long)k_off + j0, ke - j0, 1.f, tid);
        cm_stage_cols(sKt, kv, ldkv, (long)k_off + j0, ke - j0, 1.f, tid);
        __syncthreads();
        float4 lin[2][4], gat[2][4];
        bool ok[2][4];
        size_t bidx[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + t * 16 + kq * 4 + r;
                ok[t][r] = qvalid && j < ke;
                bidx[t][r] = (size_t)(pair_off + (long)(qvalid ? qi : 0) * sq + (long)min(j, ke - 1) * sk) * bias_ld;
                lin[t][r] = *(const float4*)(bias + bidx[t][r] + lin_col);
                gat[t][r] = *(const float4*)(bias + bidx[t][r] + gate_col);
            }
        float dl[2][4][4], dgt[2][4][4];                                  // [t][r][head]: the pair's two 16-B gradient segments
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sK, t * 16 + n, h * 4 + kq), bq[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sV, t * 16 + n, h * 4 + kq), bdo[h],
                                                                 f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // the weight of (query, key) exactly as the forward applied it: the probability relative to the running maximum
            // of ITS chunk, rounded to bf16 (the P operand of P.V), times exp(running max - log-sum-exp) in fp32
            float sgv[8], mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lv = ((const float*)&lin[t][r])[h], sg = sigmoid_f(((const float*)&gat[t][r])[h]);
                    sgv[t * 4 + r] = sg;
                    s[t][r] = ok[t][r] ? s[t][r] + lv * sg : -INFINITY;
                    mx = fmaxf(mx, s[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(mrun[h], mx);
            const float ms = (mn == -INFINITY) ? 0.f : mn;
            const float wsc = __expf(ms - L[h]);
            mrun[h] = mn;
            float dsv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lv = ((const float*)&lin[t][r])[h], sg = sgv[t * 4 + r];
                    const float w = cm_round(__expf(s[t][r] - ms)) * wsc;
                    const float dsj = w * (dp[t][r] - D[h]);
                    dsv[t * 4 + r] = dsj;
                    dl[t][r][h] = dsj * sg;
                    dgt[t][r][h] = dsj * lv * sg * (1.f - sg);
                }
            const bf16x8_t df = cm_pack8(dsv);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                dq[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sKt, h * 32 + mi * 16 + n, kq), df, dq[h][mi], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ok[t][r]) {
                    *(float4*)(dbias + bidx[t][r] + lin_col) = make_float4(dl[t][r][0], dl[t][r][1], dl[t][r][2], dl[t][r][3]);
                    *(float4*)(dbias + bidx[t][r] + gate_col) = make_float4(dgt[t][r][0], dgt[t][r][1], dgt[t][r][2], dgt[t][r][3]);
                }
    }
    if (!qvalid) return;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int c = h * 32 + mi * 16 + kq * 4;
            const float4 val = make_float4(dq[h][mi][0] * scale, dq[h][mi][1] * scale, dq[h][mi][2] * scale, dq[h][mi][3] * scale);
            if (ksplit > 1)     // partial over this key range -> part[split][complex][local query][128]
                *(float4*)(part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * part_rows + qi) * 128 + c) = val;
            else
                *(float4*)(dqg + qrow * ldq + c) = val;
        }
}

// ------------------------------------------------------------------------------------------------
// backward, pass KV: d k, d v (wave = 16 keys, loop over 32-query chunks; probabilities recomputed from the saved log-sum-exp)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_attn_mfma_bwd_kv_kernel(const float* __restrict__ qg, int ldq, const float* __restrict__ kv,
                                                                     int ldkv, const float* __restrict__ bias, int bias_ld,
                                                                     int lin_col, int gate_col, const int* desc, float scale,
                                                                     const float* lse, const float* __restrict__ dO, const float* Dv,
                                                                     float* dkv, int qsplit, float* part, int part_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char sQ[CM_KC * 256];
    __shared__ __attribute__((aligned(16))) unsigned char sDO[CM_KC * 256];
    __shared__ __attribute__((aligned(16))) unsigned char sQt[128 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char sDOt[128 * 64];
    __shared__ float sL[CM_KC * 4], sD[CM_KC * 4];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int kj = blockIdx.x * 64 + wave * 16 + n;
    const bool kvalid = kj < nk;
    const size_t krow = (size_t)(k_off + (kvalid ? kj : 0));
    bf16x8_t bk[4], bv[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float fk[8], fv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 a = *(const float4*)(kv + krow * ldkv + h * 32 + kq * 8 + half * 4);
            const float4 b = *(const float4*)(kv + krow * ldkv + 128 + h * 32 + kq * 8 + half * 4);
            fk[half * 4] = a.x; fk[half * 4 + 1] = a.y; fk[half * 4 + 2] = a.z; fk[half * 4 + 3] = a.w;
            fv[half * 4] = b.x; fv[half * 4 + 1] = b.y; fv[half * 4 + 2] = b.z; fv[half * 4 + 3] = b.w;
        }
        if (!kvalid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { fk[u] = 0.f; fv[u] = 0.f; }
        }
        bk[h] = cm_pack8(fk);
        bv[h] = cm_pack8(fv);
    }
    f32x4_t dk[4][2], dv[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) { dk[h][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[h][mi] = dk[h][mi]; }
    int qb = 0, qe = nq;
    if (qsplit > 1) {
        const int per = ((nq + qsplit - 1) / qsplit + CM_KC - 1) / CM_KC * CM_KC;
        qb = blockIdx.z * per; qe = min(nq, qb + per);
    }
    for (int i0 = qb; i0 < qe; i0 += CM_KC) {
        __syncthreads();
        // q is staged PRE-SCALED (like the forward's fragment), so that the recomputed scores round exactly as the ones whose
        // log-sum-exp was saved; d k = dS^T (scale q) then needs no further factor
        cm_stage_rows(sQ, qg, ldq, (long)q_off + i0, qe - i0, scale, tid);
        cm_stage_rows(sDO, dO, 128, (long)q_off + i0, qe - i0, 1.f, tid);
        cm_stage_cols(sQt, qg, ldq, (long)q_off + i0, qe - i0, scale, tid);
        cm_stage_cols(sDOt, dO, 128, (long)q_off + i0, qe - i0, 1.f, tid);
        if (tid < CM_KC * 4) {
            const int ir = tid >> 2;
            sL[tid] = (i0 + ir < qe) ? lse[(size_t)(q_off + i0 + ir) * 4 + (tid & 3)] : 0.f;
            sD[tid] = (i0 + ir < qe) ? Dv[(size_t)(q_off + i0 + ir) * 4 + (tid & 3)] : 0.f;
        }
        __syncthreads();
        // S[query, key] = Q . K^T: the lane holds queries {t*16 + kq*4 + r} of its key
        float4 lin[2][4], gat[2][4];
        bool ok[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + t * 16 + kq * 4 + r;
                ok[t][r] = kvalid && i < qe;
                const size_t bi = (size_t)(pair_off + (long)min(i, qe - 1) * sq + (long)(kvalid ? kj : 0) * sk) * bias_ld;
                lin[t][r] = *(const float4*)(bias + bi + lin_col);
                gat[t][r] = *(const float4*)(bias + bi + gate_col);
            }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sQ, t * 16 + n, h * 4 + kq), bk[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sDO, t * 16 + n, h * 4 + kq), bv[h],
                                                                 f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // NOTE the first contraction here has queries as M (rows of the LDS image) and this lane's key as N
            float pv[8], dsv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = t * 16 + kq * 4 + r;
                    const float lv = ((const float*)&lin[t][r])[h], sg = sigmoid_f(((const float*)&gat[t][r])[h]);
                    const float p = ok[t][r] ? __expf(s[t][r] + lv * sg - sL[il * 4 + h]) : 0.f;
                    pv[t * 4 + r] = p;
                    dsv[t * 4 + r] = p * (dp[t][r] - sD[il * 4 + h]);
                }
            const bf16x8_t pf = cm_pack8(pv), df = cm_pack8(dsv);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                dv[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sDOt, h * 32 + mi * 16 + n, kq), pf, dv[h][mi], 0, 0, 0);
                dk[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sQt, h * 32 + mi * 16 + n, kq), df, dk[h][mi], 0, 0, 0);
            }
        }
    }
    if (!kvalid) return;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int c = h * 32 + mi * 16 + kq * 4;
            const float4 a = make_float4(dk[h][mi][0], dk[h][mi][1], dk[h][mi][2], dk[h][mi][3]);
            const float4 b = make_float4(dv[h][mi][0], dv[h][mi][1], dv[h][mi][2], dv[h][mi][3]);
            if (qsplit > 1) {   // partial over this query range -> part[split][complex][local key][256] (dk | dv)
                float* pp = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * part_rows + kj) * 256;
                *(float4*)(pp + c) = a;
                *(float4*)(pp + 128 + c) = b;
            } else {
                *(float4*)(dkv + krow * ldkv + c) = a;
                *(float4*)(dkv + krow * ldkv + 128 + c) = b;
            }
        }
}

__global__ __launch_bounds__(256) void cross_attn_mfma_split_reduce_kernel(const float* __restrict__ part, int nsplit, int part_rows,
                                                                           int cols, const int* __restrict__ desc, int off_idx,
                                                                           int cnt_idx, float* __restrict__ out, int ldo) {
    const int* ds = desc + blockIdx.y * 8;
    const int off = ds[off_idx], n = ds[cnt_idx];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int r = idx / cols, c = idx % cols;
    if (r >= n) return;
    float acc = 0.f;
    for (int z = 0; z < nsplit; ++z) acc += part[(((size_t)z * gridDim.y + blockIdx.y) * part_rows + r) * cols + c];
    out[(size_t)(off + r) * ldo + c] = acc;
}

static void cm_splits(int max_nq, int max_nk, int* ksplit, int* qsplit) {
    *ksplit = (max_nq <= 256 && max_nk >= 512) ? max_nk / 128 : 1;
    *qsplit = (max_nk <= 256 && max_nq >= 512) ? max_nq / 128 : 1;
    if (*ksplit > 32) *ksplit = 32;
    if (*qsplit > 32) *qsplit = 32;
}

// Same contract, arguments and scratch size (fabind_cross_attn_bwd_scratch) as fabind_cross_attn_bwd.
extern "C" int fabind_cross_attn_mfma_bwd(const float* qg, int ldq, const float* kv, int ldkv, const float* bias, int bias_ld,
                                          int lin_col, int gate_col, const int* desc, int B, int max_nq, int max_nk, float scale,
                                          const float* out, const float* lse, const float* dout, float* dqg, float* dkv, float* dbias,
                                          float* dO, float* Dv, float* scratch, hipStream_t stream) {
    if (B <= 0) return 0;
    FB_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0, "fabind_cross_attn_mfma_bwd: strides % 4");
    FB_REQUIRE(bias_ld % 4 == 0 && lin_col % 4 == 0 && gate_col % 4 == 0, "fabind_cross_attn_mfma_bwd: bias columns % 4");
    int ksplit, qsplit;
    cm_splits(max_nq, max_nk, &ksplit, &qsplit);
    FB_REQUIRE((ksplit == 1 && qsplit == 1) || scratch != nullptr,
               "fabind_cross_attn_mfma_bwd: this shape splits a pass and needs fabind_cross_attn_bwd_scratch() floats of scratch");
    hipLaunchKernelGGL(cross_attn_mfma_bwd_q_kernel, dim3((max_nq + 63) / 64, B, ksplit), dim3(256), 0, stream, qg, ldq, kv, ldkv,
                       bias, bias_ld, lin_col, gate_col, desc, scale, out, lse, dout, dqg, dbias, dO, Dv, ksplit, scratch, max_nq);
    if (ksplit > 1)
        hipLaunchKernelGGL(cross_attn_mfma_split_reduce_kernel, dim3((max_nq * 128 + 255) / 256, B), dim3(256), 0, stream, scratch,
                           ksplit, max_nq, 128, desc, 0, 1, dqg, ldq);
    hipLaunchKernelGGL(cross_attn_mfma_bwd_kv_kernel, dim3((max_nk + 63) / 64, B, qsplit), dim3(256), 0, stream, qg, ldq, kv, ldkv,
                       bias, bias_ld, lin_col, gate_col, desc, scale, lse, dO, Dv, dkv, qsplit, scratch, max_nk);
    if (qsplit > 1)
        hipLaunchKernelGGL(cross_attn_mfma_split_reduce_kernel, dim3((max_nk * 256 + 255) / 256, B), dim3(256), 0, stream, scratch,
                           qsplit, max_nk, 256, desc, 2, 3, dkv, ldkv);
    FB_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================================
// Cross attention with the pair bias RECOMPUTED IN THE KERNEL (round 3): the block's matrix-core form.
//
// In the v1 stack every layer's RowAttention pair bias is  bias[h,i,j] = lin_h(z0_ij) * sigmoid(gate_h(z0_ij))  of the SAME factored pair
// embedding z0_ij = W_o (a_i . b_j) + b_o (att_model.py:198-206, cross_att.py:118-134), i.e. with the composed weights Wc = W_{lin|gate} W_o
//     lin_h / gate_h (i, j) = sum_k a_i[k] * (b_j[k] Wc[o,k]) + c_o          -- a K = H contraction per head-column o (8 of them).
// Rounds 1-2 evaluated it as one ragged GEMM per layer and block into eight [pairs, 8] fp32 tensors (126 MB each at the headline
// shape), which the three attention kernels then stream at 32 B per pair against 512 flop: HBM / latency-bound at 1 % matrix-core
// utilisation.  Here a work-group owns 64 PROTEIN rows of one complex:
//   A. stage their a0 rows (bf16) into a swizzled [64][H] LDS tile and contract them with the complex's packed ligand-side operand
//      Bo[(atom, slot), k] = b_atom[k] Wc[slot, k] (fragments streamed from L2, built once per layer and block by pair_bo_pack_kernel;
//      slots per atom: lin0, gate0, lin1, gate1, lin2, gate2, lin3, gate3) in the operand-swapped form: a lane's accumulator quad is
//      (lin_h, gate_h, lin_h+1, gate_h+1) of ONE (protein row, atom) pair -- the bias lin * sigmoid(gate) is formed in registers and
//      written as a [.][.][4 heads] fp32 LDS tile over the dead a0 tile: 8.2 kFLOP per pair on the matrix cores instead of 32 B of HBM;
//   B. runs the attention of cross_attn_mfma_fwd_kernel with the bias read from that tile:
//      MODE 0 (protein-query block): the 64 rows are the queries of 4 waves, keys = the complex's ligand atoms;
//      MODE 1 (ligand-query block):  the 64 rows are the KEYS (two 32-key chunks), queries = the ligand atoms (16 per wave); the
//              un-normalised partial (o, m, l) of every (atom, head) goes to `part` and cross_attn_mfma_combine_kernel merges the
//              ceil(P / 64) row tiles of a complex.
// Ligands of up to 62 atoms (the bias tile of 64 rows x 62 atoms x 4 heads is what fits beside the K / V images at two work-groups
// per CU); larger ones take the tensor path.  Forward only so far: under autograd the tensor path runs (its backward reads the bias
// tensors).  Reference: cross_att.py:118-134, model_utils.py:21-38.
// =====================================================================================================================
// K / V staging of the fused kernel with ALL global loads of a chunk in flight before the first LDS store (cm_stage_rows / cm_stage_cols
// above interleave load and store per iteration; the phase counters of tools/probes/attn_fused_phases.py showed the fused kernel spending
// 11 k cycles per 32-key chunk there -- a dozen serialised L2 round trips -- against ~8 k of matrix-core work in its whole contraction)
__device__ __forceinline__ void cf_stage_kv(unsigned char* sK, unsigned char* sVt, const float* __restrict__ k, const float* __restrict__ v,
                                            int ld, long row0, int nrows, int tid) {
    float4 kr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, jr = i >> 5, c4 = (i & 31) * 4;
        kr[u] = (jr < nrows) ? *(const float4*)(k + (size_t)(row0 + jr) * ld + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int col = tid & 127, g = tid >> 7;
    float f[2][8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = g * 2 + kk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + kq * 4 + r;
                f[kk][t * 4 + r] = (row < nrows) ? v[(size_t)(row0 + row) * ld + col] : 0.f;
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256;
        cm_store_rows(sK, i >> 5, (i & 31) * 4, kr[u]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = g * 2 + kk;
        *(bf16x8_t*)(sVt + col * 64 + ((kq ^ ((col >> 2) & 3)) << 4)) = cm_pack8(f[kk]);
    }
}

#define CF_ROWS 64
#define CF_KEYS 62                                  // atoms per bias tile; MODE 1 row stride = 62 * 4 + 4 floats (bank-conflict-free)

// Bo fragments of one layer / block: out[(toff[b] * NKS + ks * T_b + tl) * 512 + lane * 8 + e], T_b = ceil(C_b / 2) column tiles
__global__ __launch_bounds__(64) void pair_bo_pack_kernel(const float* __restrict__ b0, int ldb, const float* __restrict__ wcomp, int H,
                                                         const int* __restrict__ c_index, const int* __restrict__ desc,
                                                         const int* __restrict__ toff, const int* __restrict__ tile_b, bf16_t* out,
                                                         long blk_stride) {
    wcomp += (size_t)blockIdx.z * 8 * H;                            // blockIdx.z: layer / block (all packs of a model call in one launch)
    out += (size_t)blockIdx.z * blk_stride;
    const int t = blockIdx.x, ks = blockIdx.y, lane = threadIdx.x;
    const int b = tile_b[t];
    const int tl = t - toff[b], T = toff[b + 1] - toff[b];
    const int fr = lane & 15, fq = lane >> 4;
    const int c = tl * 16 + fr, key = c >> 3, slot = c & 7;
    const int C = desc[b * 8 + 3], lig0 = desc[b * 8 + 2];
    const int wrow = (slot & 1) * 4 + (slot >> 1);                // slot order lin0, gate0, lin1, gate1, ...: wcomp rows are lin0..3, gate0..3
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (key < C) {
        const size_t node = (size_t)c_index[lig0 + key];
        const int k0 = ks * 32 + fq * 8;
        const float4 x0 = *(const float4*)(b0 + node * ldb + k0), x1 = *(const float4*)(b0 + node * ldb + k0 + 4);
        const float4 w0 = *(const float4*)(wcomp + (size_t)wrow * H + k0), w1 = *(const float4*)(wcomp + (size_t)wrow * H + k0 + 4);
        f[0] = x0.x * w0.x; f[1] = x0.y * w0.y; f[2] = x0.z * w0.z; f[3] = x0.w * w0.w;
        f[4] = x1.x * w1.x; f[5] = x1.y * w1.y; f[6] = x1.z * w1.z; f[7] = x1.w * w1.w;
    }
    const int NKS = H / 32;
    *(bf16x8_t*)(out + ((size_t)toff[b] * NKS + (size_t)ks * T + tl) * 512 + lane * 8) = cm_pack8(f);
}

extern "C" int fabind_pair_bo_pack(const float* b0, int ldb, const float* wcomp, int H, const int* c_index, const int* desc,
                                   const int* toff, const int* tile_b, int n_tiles, void* out, int n_blocks, long blk_stride,
                                   hipStream_t stream) {
    if (n_tiles <= 0 || n_blocks <= 0) return 0;
    FB_REQUIRE(H % 32 == 0 && ldb % 4 == 0 && ((uintptr_t)b0 & 15) == 0 && ((uintptr_t)wcomp & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "fabind_pair_bo_pack: H % 32 == 0, 16-byte aligned rows");
    hipLaunchKernelGGL(pair_bo_pack_kernel, dim3(n_tiles, H / 32, n_blocks), dim3(64), 0, stream, b0, ldb, wcomp, H, c_index, desc, toff, tile_b,
                       (bf16_t*)out, blk_stride);
    FB_CHECK_LAUNCH();
    return 0;
}

// The bias contraction of one wave: acc[i][j] += sA tile (64 rows) x the wave's column tiles t0 .. t0 + NJ - 1 (tiles past `nt` repeat the
// last one; their results are not used).  Operand-swapped form, rolled two k-steps per trip with ping-pong fragments (fused_common.h).
// NJ = 8: both halves of a wave's tile range in ONE k-loop -- 8 fragment loads in flight against 32 MFMAs per half-trip; with the two
// halves run one after the other (NJ = 4 twice) a half-trip's 16 MFMAs (256 cycles) covered less than an L2 round trip at two waves
// per SIMD, and the kernel sat at 14 % matrix-core utilisation waiting for weight fragments (profiles/r03_pmc_util.txt).
template <int H, int SWZ, int NJ>
__device__ __forceinline__ void cf_gemm_pass(const bf16_t* sA, const bf16_t* __restrict__ bo_c, int T, int t0, int nt, int lane,
                                             f32x4_t (&acc)[4][NJ]) {
    constexpr int NKS = H / 32;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* wp[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wp[j] = bo_c + (size_t)(t0 + min(j, nt - 1)) * 512 + lane * 8;
    const size_t kstride = (size_t)T * 512;
    bf16x8_t b0[NJ], b1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b0[j] = *(const bf16x8_t*)wp[j];
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) b1[j] = *(const bf16x8_t*)(wp[j] + (size_t)(ks + 1) * kstride);
        bf16x8_t a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 16 + fr;
            a[i] = *(const bf16x8_t*)&sA[r * H + (((ks * 4 + fq) ^ (r & SWZ)) * 8)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a[i], acc[i][j], 0, 0, 0);
        {
            const int kn = min(ks + 2, NKS - 2);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b0[j] = *(const bf16x8_t*)(wp[j] + (size_t)kn * kstride);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 16 + fr;
            a[i] = *(const bf16x8_t*)&sA[r * H + ((((ks + 1) * 4 + fq) ^ (r & SWZ)) * 8)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a[i], acc[i][j], 0, 0, 0);
    }
}

template <int H, int MODE>
__global__ __launch_bounds__(256, 2) void cross_attn_fused_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                                      const float* __restrict__ v, int ldkv,
                                                                      const float* __restrict__ gpre, int ldg,
                                                                      const bf16_t* __restrict__ a0, int lda0,
                                                                      const bf16_t* __restrict__ bo, const int* __restrict__ toff,
                                                                      const float* __restrict__ bconst, const int* __restrict__ desc,
                                                                      float scale, float* out, int ldo, float* part, int nsplit, int B,
                                                                      long long* dbg, float* lse) {
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    constexpr int TILE_BYTES = (CF_ROWS * H * 2 > 65536) ? CF_ROWS * H * 2 : 65536;
    // XCD-aware work-group order (1-D grid; consecutive ids go round-robin to the 8 XCDs): ALL row tiles of a complex run on the XCD
    // (complex % 8), so the complex's packed operand Bo (336 KiB at 41 ligand-side nodes) is fetched into that XCD's L2 once and hit by
    // the other tiles -- with tile-major ids every XCD streamed every complex's Bo from HBM and the k-loop ran at HBM latency
    const int cplx = (int)(blockIdx.x >> 3) / nsplit * 8 + (int)(blockIdx.x & 7);
    const int rtile = (int)(blockIdx.x >> 3) % nsplit;
    if (cplx >= B) return;
    // bias tile: MODE 0 [atom][row][4] (lanes of a wave = rows: consecutive 16-byte reads), MODE 1 [row][CF_KEYS + 1][4] (lanes = atoms)
    constexpr int LDB1 = (CF_KEYS + 1) * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sA = (bf16_t*)smem;
    float* sB = (float*)smem;                                       // over the a0 tile once the contraction is done
    unsigned char* sK = smem + TILE_BYTES;
    unsigned char* sVt = sK + CM_KC * 256;
    const int* ds = desc + cplx * 8;
    const int prow0 = ds[0], P = ds[1], lig0 = ds[2], C = ds[3];
    const int m0 = rtile * CF_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    if (m0 >= P) {
        if (MODE == 1) {                                            // neutral partials of a row tile past the end of this complex
            for (int t = tid; t < C * 4; t += 256) {
                float* pp = part + (((size_t)(lig0 + (t >> 2)) * nsplit + rtile) * 4 + (t & 3)) * 34;
#pragma unroll
                for (int d = 0; d < 32; ++d) pp[d] = 0.f;
                pp[32] = -INFINITY; pp[33] = 0.f;
            }
        }
        return;
    }
    const int nrow = min(CF_ROWS, P - m0);
    // phase cycle counters of one work-group's wave 0 (tools/probes/attn_fused_phases.py): t[0] start, [1] a0 tile staged, [2] contraction
    // done, [3] bias tile written, [4] attention done
    long long tph[5] = {0, 0, 0, 0, 0};
    const bool timed = dbg != nullptr && blockIdx.x == 8 * 3 && wave == 0;      // a work-group in the middle of a complex
    if (timed) tph[0] = __builtin_readcyclecounter();

    // ---- A1: a0 rows of the tile -> swizzled LDS image (rows past the end repeat the last row; their results are not used); every load
    // is requested before the first store (a rolled load / store loop cost 12 k cycles of serialised round trips per work-group), and the
    // first K / V chunk of the attention phase is staged here as well, under the same latency
    const int ke = MODE == 0 ? C : nrow;                            // keys: the ligand-side nodes / the rows of the tile
    const long krow0 = MODE == 0 ? (long)lig0 : (long)prow0 + m0;
    {
        constexpr int CH = H / 8, NLD = CF_ROWS * CH / 256;
        uint4 val[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c = tid + u * 256, r = c / CH, ch = c % CH;
            val[u] = *(const uint4*)(a0 + (size_t)(prow0 + m0 + min(r, nrow - 1)) * lda0 + ch * 8);
        }
        cf_stage_kv(sK, sVt, k, v, ldkv, krow0, ke, tid);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c = tid + u * 256, r = c / CH, ch = c % CH;
            *(uint4*)&sA[r * H + ((ch ^ (r & SWZ)) * 8)] = val[u];
        }
    }
    __syncthreads();
    if (timed) tph[1] = __builtin_readcyclecounter();

    // ---- A2: bias contraction.  T column tiles (2 atoms each) are dealt to the 4 waves in contiguous ranges of <= 8 (two passes of 4)
    const int T = toff[cplx + 1] - toff[cplx];
    const bf16_t* bo_c = bo + (size_t)toff[cplx] * (H / 32) * 512;
    const int per = (T + 3) / 4;
    const int t0 = wave * per, t1 = min(T, t0 + per);
    const int nw = t1 - t0;                                         // this wave's tiles (<= 8)
    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (nw > 4) cf_gemm_pass<H, SWZ, 8>(sA, bo_c, T, t0, nw, lane, acc);
    else if (nw > 0) {                                              // small ligands (<= 32 ligand-side nodes): four tiles per wave at most
        f32x4_t acc4[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        cf_gemm_pass<H, SWZ, 4>(sA, bo_c, T, t0, nw, lane, acc4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = acc4[i][j];
    }
    __syncthreads();                                                // every wave has finished reading the a0 tile
    if (timed) tph[2] = __builtin_readcyclecounter();

    // ---- A3: bias = (lin + c_lin) * sigmoid(gate + c_gate) for the lane's two heads -> LDS tile
    {
        const int hb = (kq & 1) * 2;                                // this lane's quad: (lin_hb, gate_hb, lin_hb+1, gate_hb+1)
        const float cl0 = bconst[hb], cg0 = bconst[4 + hb], cl1 = bconst[hb + 1], cg1 = bconst[4 + hb + 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nw) {
                const int atom = (t0 + j) * 2 + (kq >> 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 16 + n;
                    const float2 bv = make_float2((acc[i][j][0] + cl0) * sigmoid_f(acc[i][j][1] + cg0),
                                                  (acc[i][j][2] + cl1) * sigmoid_f(acc[i][j][3] + cg1));
                    if (atom < CF_KEYS) {
                        if (MODE == 0) *(float2*)&sB[(atom * CF_ROWS + row) * 4 + hb] = bv;
                        else *(float2*)&sB[row * LDB1 + atom * 4 + hb] = bv;
                    }
                }
            }
        }
    }
    // (the barrier in front of the first K / V staging below orders these writes before the reads)
    if (timed) tph[3] = __builtin_readcyclecounter();

    // ---- B: attention (cross_attn_mfma_fwd_kernel with the bias from the tile)
    const int nq_total = MODE == 0 ? nrow : C;                      // queries this work-group serves
    const int qi = wave * 16 + n;                                   // MODE 0: row of the tile; MODE 1: atom
    const bool qvalid = qi < nq_total;
    const size_t qrow = MODE == 0 ? (size_t)(prow0 + m0 + (qvalid ? qi : 0)) : (size_t)(lig0 + (qvalid ? qi : 0));
    bf16x8_t bq[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float f[8];
        const float4 a = *(const float4*)(q + qrow * ldq + h * 32 + kq * 8), b = *(const float4*)(q + qrow * ldq + h * 32 + kq * 8 + 4);
        f[0] = a.x * scale; f[1] = a.y * scale; f[2] = a.z * scale; f[3] = a.w * scale;
        f[4] = b.x * scale; f[5] = b.y * scale; f[6] = b.z * scale; f[7] = b.w * scale;
        if (!qvalid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = 0.f;
        }
        bq[h] = cm_pack8(f);
    }
    f32x4_t o[4][2];
    float m[4], l[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        m[h] = -INFINITY; l[h] = 0.f;
        o[h][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o[h][1] = o[h][0];
    }
    for (int j0 = 0; j0 < ke; j0 += CM_KC) {
        __syncthreads();                                            // (chunk 0: orders the bias-tile writes before their reads)
        if (j0 > 0) {                                               // chunk 0 was staged with the a0 tile
            cf_stage_kv(sK, sVt, k, v, ldkv, krow0 + j0, ke - j0, tid);
            __syncthreads();
        }
        float4 bia[2][4];
        bool ok[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + t * 16 + kq * 4 + r;
                ok[t][r] = qvalid && j < ke;
                const int jc = min(j, ke - 1), qc = qvalid ? qi : 0;
                bia[t][r] = MODE == 0 ? *(const float4*)&sB[(jc * CF_ROWS + qc) * 4] : *(const float4*)&sB[jc * LDB1 + qc * 4];
            }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sK, t * 16 + n, h * 4 + kq), bq[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = ok[t][r] ? s[t][r] + ((const float*)&bia[t][r])[h] : -INFINITY;
                    s[t][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[h], mx);
            const float ms = (mn == -INFINITY) ? 0.f : mn;
            const float corr = __expf(m[h] - ms);
            float p[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[t * 4 + r] = __expf(s[t][r] - ms);
            const bf16x8_t pf = cm_pack8(p);
            l[h] = l[h] * corr + cm_sum8(pf);                     // (the normaliser sums the ROUNDED probabilities: see the tensor-path kernel)
            m[h] = mn;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h][mi][r] *= corr;
                o[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sVt, h * 32 + mi * 16 + n, kq), pf, o[h][mi], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        l[h] += __shfl_xor(l[h], 16, 64);
        l[h] += __shfl_xor(l[h], 32, 64);
    }
    if (timed && lane == 0) {
        tph[4] = __builtin_readcyclecounter();
#pragma unroll
        for (int u = 0; u < 5; ++u) dbg[u] = tph[u];
    }
    if (!qvalid) return;
    if (MODE == 1) {       // un-normalised partial of this row tile: [atom][tile][head][34] = o[32], m, l
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float* pp = part + (((size_t)(lig0 + qi) * nsplit + rtile) * 4 + h) * 34;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) pp[mi * 16 + kq * 4 + r] = o[h][mi][r];
            if (kq == 0) { pp[32] = m[h]; pp[33] = l[h]; }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float inv = l[h] > 0.f ? 1.f / l[h] : 0.f;     // no key at all (an empty ligand side): zeros, not 0 * inf
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int c = h * 32 + mi * 16 + kq * 4;
            const float4 g = *(const float4*)(gpre + qrow * ldg + c);
            *(float4*)(out + qrow * ldo + c) = make_float4(o[h][mi][0] * inv * sigmoid_f(g.x), o[h][mi][1] * inv * sigmoid_f(g.y),
                                                           o[h][mi][2] * inv * sigmoid_f(g.z), o[h][mi][3] * inv * sigmoid_f(g.w));
        }
        if (lse && kq == 0) lse[qrow * 4 + h] = m[h] + __logf(l[h]);      // (training: the fused backward replays the weights from it)
    }
}

static long long* g_cf_dbg = nullptr;      // development probe: device buffer of 5 int64 receiving one work-group's phase cycle counters
extern "C" void fabind_cross_attn_fused_set_dbg(void* p) { g_cf_dbg = (long long*)p; }

// development probe: resident work-groups per CU of the fused forward kernel (80 KiB of LDS each at H = 512)
extern "C" int fabind_cross_attn_fused_occupancy(int lds_bytes) {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)cross_attn_fused_fwd_kernel<512, 0>, 256, (size_t)lds_bytes);
    return n;
}

extern "C" int fabind_cross_attn_fused_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* gpre, int ldg,
                                           const void* a0, int lda0, const void* bo, const int* toff, const float* bconst,
                                           const int* desc, int B, int max_P, int max_C, int H, int mode, float scale, float* out,
                                           int ldo, float* part, int n_lig_rows, float* lse, hipStream_t stream) {
    FB_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && lda0 % 8 == 0, "fabind_cross_attn_fused_fwd: strides");
    FB_REQUIRE(max_C <= CF_KEYS, "fabind_cross_attn_fused_fwd: at most 62 ligand atoms per complex (larger ligands take the tensor path)");
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_cross_attn_fused_fwd: H must be 64, 128, 256 or 512");
    // (layout contract, not checkable from pointers: q / k / v / gpre rows are 4 heads x 32 channels, bconst8 and the packed operand's
    //  eight rows per ligand atom are ordered lin0..3 | gate0..3 -- fabind_pair_bo_pack's wcomp [8, H])
    FB_REQUIRE(mode == 0 || part != nullptr, "fabind_cross_attn_fused_fwd: the ligand-query block needs the partials buffer");
    if (B <= 0 || max_P <= 0) return 0;
    const int nsplit = (max_P + CF_ROWS - 1) / CF_ROWS;
    const size_t lds = (size_t)((CF_ROWS * H * 2 > 65536) ? CF_ROWS * H * 2 : 65536) + CM_KC * 256 + 128 * 64;
#define CF_LAUNCH(HH, MM)                                                                                                          \
    do {                                                                                                                           \
        static bool set_ = false;                                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)cross_attn_fused_fwd_kernel<HH, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((cross_attn_fused_fwd_kernel<HH, MM>), dim3(8 * nsplit * ((B + 7) / 8)), dim3(256), lds, stream, q, ldq, k, v, \
                           ldkv, gpre, ldg, (const bf16_t*)a0, lda0, (const bf16_t*)bo, toff, bconst, desc, scale, out, ldo, part, nsplit, B, \
                           g_cf_dbg, mode == 0 ? lse : (float*)nullptr);                                                         \
    } while (0)
#define CF_LAUNCH_H(MM) do { if (H == 512) CF_LAUNCH(512, MM); else if (H == 256) CF_LAUNCH(256, MM); else if (H == 128) CF_LAUNCH(128, MM); else CF_LAUNCH(64, MM); } while (0)
    if (mode == 0) CF_LAUNCH_H(0);
    else {
        CF_LAUNCH_H(1);
        hipLaunchKernelGGL(cross_attn_mfma_combine_kernel, dim3((n_lig_rows * 4 + 255) / 256), dim3(256), 0, stream, part, nsplit, gpre,
                           ldg, n_lig_rows, out, ldo, lse);
    }
#undef CF_LAUNCH_H
#undef CF_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================================
// Fused BACKWARD of the cross attention with the pair bias recomputed in the kernel (round 4, VERDICT r3 missing item 1).
//
// The tensor path's adjoint needs eight [pairs, 8] fp32 bias tensors in, eight [pairs, 8] fp32 gradients out, a concatenation pass, a
// ragged d a0 GEMM and two batched transposes before its T_k GEMMs.  Here a work-group again owns 64 protein rows of one complex:
//   Q pass  (cross_attn_fused_bwd_q_kernel):  recompute lin / gate on the matrix cores (the forward's contraction), keep
//            bias = lin * sigmoid(gate) as the fp32 LDS tile and sigmoid(gate) as a 16-bit fixed-point tile; run the tensor path's pass Q with the bias read
//            from LDS (same replay of the forward's weights from the saved log-sum-exp); every pair's gradient
//            D = (d lin_h, d gate_h)_h = (dS_h sg_h, dS_h bias_h (1 - sg_h))_h OVERWRITES its bias entry as eight bf16 (16 bytes in place);
//            then  d a0[row, :] += sum_(atom, slot) D[row, (atom, slot)] * Bo[(atom, slot), :]  is a second contraction on the matrix
//            cores, D fragments straight from that tile, Bo^T fragments from the packed operand `boT` (pair_bot_pack_kernel); the D tile
//            also goes out as bf16 rows of the tensor path's `Acat` matrix, whose transposed copy feeds the T_k = D_k^T a0 GEMM + finishing
//            pass of the ligand-side / weight gradients (sums over ALL row tiles of a complex: not a per-tile job).
//   KV pass (cross_attn_fused_bwd_kv_kernel): recompute the bias tile the same way and run the tensor path's pass KV on it.
// MODE 0: queries = the 64 protein rows, keys = the ligand-side nodes (d k / d v are per-tile partials, reduced by
// cross_attn_mfma_split_reduce_kernel);  MODE 1: queries = the ligand-side nodes (d q per-tile partials), keys = the rows.
// One work-group per CU (LDS: 64 KiB tile + 32 KiB sigmoid tile + 24 KiB K / V images).  Reference: autograd of cross_att.py:118-134,
// model_utils.py:21-38,96-159.
// =====================================================================================================================
#define CFB_SG_BYTES 32768

// BoT fragments of one layer / block: out[((koff[b] + ks) * (H / 16) + nt) * 512 + lane * 8 + e] = b0[atom][h] * wcomp[wrow(e)][h],
// atom = ks * 4 + (lane >> 4), h = nt * 16 + (lane & 15); atoms past C are zero rows
__global__ __launch_bounds__(64) void pair_bot_pack_kernel(const float* __restrict__ b0, int ldb, const float* __restrict__ wcomp, int H,
                                                          const int* __restrict__ c_index, const int* __restrict__ desc,
                                                          const int* __restrict__ koff, const int* __restrict__ kstep_b, bf16_t* out,
                                                          long blk_stride) {
    wcomp += (size_t)blockIdx.z * 8 * H;
    out += (size_t)blockIdx.z * blk_stride;
    const int kg = blockIdx.x, nt = blockIdx.y, lane = threadIdx.x;
    const int b = kstep_b[kg];
    const int ks = kg - koff[b];
    const int fr = lane & 15, fq = lane >> 4;
    const int atom = ks * 4 + fq, h = nt * 16 + fr;
    const int C = desc[b * 8 + 3], lig0 = desc[b * 8 + 2];
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (atom < C) {
        const float x = b0[(size_t)c_index[lig0 + atom] * ldb + h];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = x * wcomp[(size_t)((e & 1) * 4 + (e >> 1)) * H + h];
    }
    *(bf16x8_t*)(out + ((size_t)kg * (H / 16) + nt) * 512 + lane * 8) = cm_pack8(f);
}

extern "C" int fabind_pair_bot_pack(const float* b0, int ldb, const float* wcomp, int H, const int* c_index, const int* desc,
                                    const int* koff, const int* kstep_b, int n_ksteps, void* out, int n_blocks, long blk_stride,
                                    hipStream_t stream) {
    if (n_ksteps <= 0 || n_blocks <= 0) return 0;
    FB_REQUIRE(H % 16 == 0 && ((uintptr_t)out & 15) == 0, "fabind_pair_bot_pack: H % 16 == 0, 16-byte aligned output");
    hipLaunchKernelGGL(pair_bot_pack_kernel, dim3(n_ksteps, H / 16, n_blocks), dim3(64), 0, stream, b0, ldb, wcomp, H, c_index, desc, koff, kstep_b,
                       (bf16_t*)out, blk_stride);
    FB_CHECK_LAUNCH();
    return 0;
}

// Images of one 32-row chunk for the backward passes with EVERY global load requested before the first LDS store (cm_stage_rows /
// cm_stage_cols interleave load and store per iteration: a dozen serialised L2 round trips per image): row-major images of srcA and srcB
// (each [.,128] fp32 at row stride lda / ldb, scaled by mulA / 1), transposed image of srcA, and -- NT = 2 -- of srcB as well.
template <int NT>
__device__ __forceinline__ void cfb_stage(unsigned char* sRa, unsigned char* sRb, unsigned char* sTa, unsigned char* sTb,
                                          const float* __restrict__ srcA, int lda, float mulA, const float* __restrict__ srcB, int ldb,
                                          long row0, int nrows, int tid) {
    float4 ra[4], rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, jr = i >> 5, c4 = (i & 31) * 4;
        const bool okr = jr < nrows;
        ra[u] = okr ? *(const float4*)(srcA + (size_t)(row0 + jr) * lda + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[u] = okr ? *(const float4*)(srcB + (size_t)(row0 + jr) * ldb + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int col = tid & 127, g = tid >> 7;
    float fa[2][8], fb[2][8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = g * 2 + kk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + kq * 4 + r;
                fa[kk][t * 4 + r] = (row < nrows) ? srcA[(size_t)(row0 + row) * lda + col] * mulA : 0.f;
                if (NT == 2) fb[kk][t * 4 + r] = (row < nrows) ? srcB[(size_t)(row0 + row) * ldb + col] : 0.f;
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256;
        float4 a = ra[u];
        a.x *= mulA; a.y *= mulA; a.z *= mulA; a.w *= mulA;
        cm_store_rows(sRa, i >> 5, (i & 31) * 4, a);
        cm_store_rows(sRb, i >> 5, (i & 31) * 4, rb[u]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = g * 2 + kk;
        *(bf16x8_t*)(sTa + col * 64 + ((kq ^ ((col >> 2) & 3)) << 4)) = cm_pack8(fa[kk]);
        if (NT == 2) *(bf16x8_t*)(sTb + col * 64 + ((kq ^ ((col >> 2) & 3)) << 4)) = cm_pack8(fb[kk]);
    }
}

// phases A1-A3 of the forward kernel for the backward passes: a0 tile -> LDS, lin / gate contraction, bias (fp32) -> sB, sigmoid(gate)
// (bf16) -> sG (same index as sB, in elements).  Ends with every wave past its writes (the caller's next barrier orders them).
template <int H, int MODE>
__device__ __forceinline__ void cfb_bias_tile(const FabindAttnFusedBwdArgs& p, int cplx, int prow0, int m0, int nrow, bf16_t* sA, float* sB,
                                              bf16_t* sG, int tid) {
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    constexpr int LDB1 = (CF_KEYS + 1) * 4;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
    {
        constexpr int CH = H / 8, NLD = CF_ROWS * CH / 256;
        uint4 val[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c = tid + u * 256, r = c / CH, ch = c % CH;
            val[u] = *(const uint4*)((const bf16_t*)p.a0 + (size_t)(prow0 + m0 + min(r, nrow - 1)) * p.lda0 + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c = tid + u * 256, r = c / CH, ch = c % CH;
            *(uint4*)&sA[r * H + ((ch ^ (r & SWZ)) * 8)] = val[u];
        }
    }
    __syncthreads();
    const int T = p.toff[cplx + 1] - p.toff[cplx];
    const bf16_t* bo_c = (const bf16_t*)p.bo + (size_t)p.toff[cplx] * (H / 32) * 512;
    const int per = (T + 3) / 4;
    const int t0 = wave * per, t1 = min(T, t0 + per);
    const int nw = t1 - t0;
    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (nw > 4) cf_gemm_pass<H, SWZ, 8>(sA, bo_c, T, t0, nw, lane, acc);
    else if (nw > 0) {
        f32x4_t acc4[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        cf_gemm_pass<H, SWZ, 4>(sA, bo_c, T, t0, nw, lane, acc4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = acc4[i][j];
    }
    __syncthreads();                                                // every wave has finished reading the a0 tile
    if (MODE == 1)                                                  // the tail of the tile past row 63 (read as a zero-weighted k entry)
        for (int i = CF_ROWS * LDB1 + tid; i < 65536 / 4; i += 256) sB[i] = 0.f;
    const int hb = (kq & 1) * 2;
    const float* bconst = p.bconst;
    const float cl0 = bconst[hb], cg0 = bconst[4 + hb], cl1 = bconst[hb + 1], cg1 = bconst[4 + hb + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < nw) {
            const int atom = (t0 + j) * 2 + (kq >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + n;
                const float sg0 = sigmoid_f(acc[i][j][1] + cg0), sg1 = sigmoid_f(acc[i][j][3] + cg1);
                const float2 bv = make_float2((acc[i][j][0] + cl0) * sg0, (acc[i][j][2] + cl1) * sg1);
                if (atom < CF_KEYS) {
                    const int idx = MODE == 0 ? (atom * CF_ROWS + row) * 4 + hb : row * LDB1 + atom * 4 + hb;
                    *(float2*)&sB[idx] = bv;
                    // sigmoid(gate) as 16-bit FIXED point (absolute error 8e-6): as bf16 its 2^-9 relative error came back as 2 % in the
                    // constants' gradients, which are sums over all pairs with heavy cancellation
                    *(uint32_t*)&sG[idx] = __float2uint_rn(sg0 * 65535.f) | (__float2uint_rn(sg1 * 65535.f) << 16);
                }
            }
        }
    }
}

template <int H, int MODE>
__global__ __launch_bounds__(256, 1) void cross_attn_fused_bwd_q_kernel(const FabindAttnFusedBwdArgs p) {
    constexpr int LDB1 = (CF_KEYS + 1) * 4;
    const int nsplit = p.nsplit;
    const int cplx = (int)(blockIdx.x >> 3) / nsplit * 8 + (int)(blockIdx.x & 7);
    const int rtile = (int)(blockIdx.x >> 3) % nsplit;
    if (cplx >= p.B) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sA = (bf16_t*)smem;
    float* sB = (float*)smem;
    bf16_t* sG = (bf16_t*)(smem + 65536);
    unsigned char* sK = smem + 65536 + CFB_SG_BYTES;
    unsigned char* sV = sK + CM_KC * 256;
    unsigned char* sKt = sV + CM_KC * 256;
    const int* ds = p.desc + cplx * 8;
    const int prow0 = ds[0], P = ds[1], lig0 = ds[2], C = ds[3];
    const int ppos0 = p.desc_p[cplx * 8];                           // first row of the complex in the compact protein list (Acat rows)
    const int m0 = rtile * CF_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    float* colp = p.colpart + (size_t)(cplx * nsplit + rtile) * p.ldcolpart;
    if (m0 >= P) {
        if (tid < 8) colp[tid] = 0.f;
        if (MODE == 1)                                              // neutral d q partials of a row tile past the end of this complex
            for (int t = tid; t < C * 32; t += 256)
                *(float4*)(p.part + (((size_t)(lig0 + (t >> 5)) * nsplit + rtile) * 128) + (t & 31) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int nrow = min(CF_ROWS, P - m0);
    const int ke = MODE == 0 ? C : nrow;
    const long krow0 = MODE == 0 ? (long)lig0 : (long)prow0 + m0;
    cfb_stage<1>(sK, sV, sKt, nullptr, p.kv, p.ldkv, 1.f, p.kv + 128, p.ldkv, krow0, ke, tid);      // chunk 0, under the bias phase's latency
    cfb_bias_tile<H, MODE>(p, cplx, prow0, m0, nrow, sA, sB, sG, tid);

    // ---- pass Q on the tile
    const int nq_total = MODE == 0 ? nrow : C;
    const int qi = wave * 16 + n;
    const bool qvalid = qi < nq_total;
    const size_t qrow = MODE == 0 ? (size_t)(prow0 + m0 + (qvalid ? qi : 0)) : (size_t)(lig0 + (qvalid ? qi : 0));
    const bool first = MODE == 0 || rtile == 0;                     // per-query outputs of a ligand-side query: written by its first tile only
    const float scale = p.scale;
    const float* qg = p.qg;
    const int ldq = p.ldq;
    bf16x8_t bq[4], bdo[4];
    float D[4], L[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int c = h * 32 + kq * 8;
        float fq_[8], fo[8], dsum = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 a = *(const float4*)(qg + qrow * ldq + c + half * 4);
            const float4 g = *(const float4*)(qg + qrow * ldq + 128 + c + half * 4);
            const float4 ov = *(const float4*)(p.out + qrow * 128 + c + half * 4);
            const float4 dv = *(const float4*)(p.dout + qrow * 128 + c + half * 4);
            const float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w}, o4[4] = {ov.x, ov.y, ov.z, ov.w},
                        d4[4] = {dv.x, dv.y, dv.z, dv.w};
            float dg[4], dov[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sg = sigmoid_f(gv[u]);
                fq_[half * 4 + u] = qvalid ? av[u] * scale : 0.f;
                dov[u] = qvalid ? d4[u] * sg : 0.f;
                fo[half * 4 + u] = dov[u];
                dsum += cm_round(dov[u]) * (sg > 1e-30f ? o4[u] * __builtin_amdgcn_rcpf(sg) : 0.f);
                dg[u] = d4[u] * o4[u] * (1.f - sg);
            }
            if (qvalid && first) {
                *(float4*)(p.dqg + qrow * ldq + 128 + c + half * 4) = make_float4(dg[0], dg[1], dg[2], dg[3]);     // d gpre
                *(float4*)(p.dO + qrow * 128 + c + half * 4) = make_float4(dov[0], dov[1], dov[2], dov[3]);
            }
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        D[h] = dsum;
        L[h] = qvalid ? p.lse[qrow * 4 + h] : 0.f;
        if (qvalid && kq == 0 && first) p.Dv[qrow * 4 + h] = dsum;
        bq[h] = cm_pack8(fq_);
        bdo[h] = cm_pack8(fo);
    }
    f32x4_t dq[4][2];
    float mrun[4], csum[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) { dq[h][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[h][1] = dq[h][0]; mrun[h] = -INFINITY; }
#pragma unroll
    for (int u = 0; u < 8; ++u) csum[u] = 0.f;
    const float* kvp = p.kv;
    const int ldkv = p.ldkv;
    for (int j0 = 0; j0 < ke; j0 += CM_KC) {
        __syncthreads();                                            // (chunk 0: orders the bias / sigmoid tile writes before their reads)
        if (j0 > 0) {
            cfb_stage<1>(sK, sV, sKt, nullptr, kvp, ldkv, 1.f, kvp + 128, ldkv, krow0 + j0, ke - j0, tid);
            __syncthreads();
        }
        float4 bia[2][4];
        uint2 sgp[2][4];
        bool ok[2][4];
        int bix[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + t * 16 + kq * 4 + r;
                ok[t][r] = qvalid && j < ke;
                const int jc = min(j, ke - 1), qc = qvalid ? qi : 0;
                bix[t][r] = MODE == 0 ? (jc * CF_ROWS + qc) * 4 : jc * LDB1 + qc * 4;
                bia[t][r] = *(const float4*)&sB[bix[t][r]];
                sgp[t][r] = *(const uint2*)&sG[bix[t][r]];
            }
        float dl[2][4][4], dgt[2][4][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sK, t * 16 + n, h * 4 + kq), bq[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sV, t * 16 + n, h * 4 + kq), bdo[h],
                                                                 f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            float bvv[8], sgv[8], mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bb = ((const float*)&bia[t][r])[h];
                    const uint32_t w2 = (h < 2) ? sgp[t][r].x : sgp[t][r].y;
                    sgv[t * 4 + r] = (float)((h & 1) ? (w2 >> 16) : (w2 & 0xffffu)) * (1.0f / 65535.f);
                    bvv[t * 4 + r] = bb;
                    s[t][r] = ok[t][r] ? s[t][r] + bb : -INFINITY;
                    mx = fmaxf(mx, s[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(mrun[h], mx);
            const float ms = (mn == -INFINITY) ? 0.f : mn;
            const float wsc = __expf(ms - L[h]);
            mrun[h] = mn;
            float dsv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = sgv[t * 4 + r];
                    const float w = cm_round(__expf(s[t][r] - ms)) * wsc;
                    const float dsj = ok[t][r] ? w * (dp[t][r] - D[h]) : 0.f;
                    dsv[t * 4 + r] = dsj;
                    dl[t][r][h] = dsj * sg;
                    dgt[t][r][h] = dsj * bvv[t * 4 + r] * (1.f - sg);          // d gate = dS lin sg (1 - sg), lin sg = bias
                }
            const bf16x8_t df = cm_pack8(dsv);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                dq[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sKt, h * 32 + mi * 16 + n, kq), df, dq[h][mi], 0, 0, 0);
        }
        // D = (d lin_0, d gate_0, d lin_1, d gate_1, ...) as eight bf16 OVER the pair's bias entry (this lane is its only reader)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + t * 16 + kq * 4 + r;
                const bool slot = MODE == 0 ? (j < 64) : (qi < CF_KEYS + 1 && j < CF_ROWS);   // the entry exists in the tile
                if (!slot) continue;
                const int ix = MODE == 0 ? (j * CF_ROWS + qi) * 4 : j * LDB1 + qi * 4;
                uint4 o = make_uint4(0u, 0u, 0u, 0u);
                if (ok[t][r]) {
                    o = make_uint4(pack2_bf16(dl[t][r][0], dgt[t][r][0]), pack2_bf16(dl[t][r][1], dgt[t][r][1]),
                                   pack2_bf16(dl[t][r][2], dgt[t][r][2]), pack2_bf16(dl[t][r][3], dgt[t][r][3]));
#pragma unroll
                    for (int h = 0; h < 4; ++h) { csum[2 * h] += dl[t][r][h]; csum[2 * h + 1] += dgt[t][r][h]; }
                }
                *(uint4*)&sB[ix] = o;
            }
    }
    // ---- d q (MODE 0: complete; MODE 1: this tile's partial)
    if (qvalid) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int c = h * 32 + mi * 16 + kq * 4;
                const float4 val = make_float4(dq[h][mi][0] * scale, dq[h][mi][1] * scale, dq[h][mi][2] * scale, dq[h][mi][3] * scale);
                if (MODE == 1) *(float4*)(p.part + (((size_t)(lig0 + qi) * nsplit + rtile) * 128) + c) = val;
                else *(float4*)(p.dqg + qrow * ldq + c) = val;
            }
    }
    // ---- column sums of D over the tile (-> d bconst): lanes -> wave -> work-group
#pragma unroll
    for (int u = 0; u < 8; ++u) csum[u] = wave_sum(csum[u]);
    __syncthreads();                                                // the D tile is complete; the K / V images are free
    float* sCs = (float*)sK;
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sCs[wave * 8 + u] = csum[u];
    }
    __syncthreads();
    if (tid < 8) colp[tid] = (sCs[tid] + sCs[8 + tid]) + (sCs[16 + tid] + sCs[24 + tid]);
    // ---- D rows -> Acat (bf16, slot order lin0, gate0, lin1, gate1, ...: the order of the packed operands)
    {
        bf16_t* ac = (bf16_t*)p.acat;
        const int na = max(C, p.kp / 8);                            // entries per row incl. the block's zero padding (columns [C * 8, kp):
        for (int idx = tid; idx < nrow * na; idx += 256) {          // the ragged GEMMs over Acat contract them against zero rows, so they must be finite)
            const int r = idx / na, a = idx - r * na;
            const uint4 v = a < C ? *(const uint4*)&sB[MODE == 0 ? (a * CF_ROWS + r) * 4 : r * LDB1 + a * 4] : make_uint4(0u, 0u, 0u, 0u);
            *(uint4*)(ac + (size_t)(ppos0 + m0 + r) * p.ldacat + p.kcol0 + a * 8) = v;
        }
    }
    // ---- d a0[row, h] += sum_(atom, slot) D[row, (atom, slot)] Bo[(atom, slot), h]: M = h (16 per tile), N = row, K = 4 atoms x 8 slots.
    // OPTIONAL (da0 != NULL).  Measured at the headline shape: eight blocks each read-modify-write the 64 x H fp32 gradient rows of a tile
    // (400 MB per launch, 3.2 GB per step) where ONE ragged GEMM over the stored bf16 gradient rows of all blocks (K = 8 x C x 8) writes
    // them once: the host takes that GEMM by default and leaves da0 NULL (ops._CrossAttnFused).
    if (p.da0 != nullptr) {
        constexpr int NJ2 = H / 64;                                 // h-tiles per wave
        const int KS = (C + 3) / 4;
        const bf16_t* bt = (const bf16_t*)p.boT + ((size_t)p.koff[cplx] * (H / 16) + wave * NJ2) * 512 + lane * 8;
        f32x4_t acc2[4][NJ2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ2; ++j) acc2[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t bf[NJ2], af[4];
#pragma unroll
            for (int j = 0; j < NJ2; ++j) bf[j] = *(const bf16x8_t*)(bt + ((size_t)ks * (H / 16) + j) * 512);
            const int atom = ks * 4 + kq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + n;
                af[i] = *(const bf16x8_t*)&sB[MODE == 0 ? (atom * CF_ROWS + row) * 4 : row * LDB1 + atom * 4];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc2[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + n;
            if (row < nrow) {
                float* dst = p.da0 + (size_t)(prow0 + m0 + row) * p.ldda0 + (wave * NJ2) * 16 + kq * 4;
#pragma unroll
                for (int j = 0; j < NJ2; ++j) {
                    float4 v = *(float4*)(dst + j * 16);
                    v.x += acc2[i][j][0]; v.y += acc2[i][j][1]; v.z += acc2[i][j][2]; v.w += acc2[i][j][3];
                    *(float4*)(dst + j * 16) = v;
                }
            }
        }
    }
}

// KV pass on the recomputed bias tile: d k, d v.  MODE 0: keys = the ligand-side nodes (one wave per 16), queries = the tile's rows in
// 32-row chunks: per-tile partials part[tile][complex][local key][256];  MODE 1: keys = the tile's rows, queries = the ligand-side nodes.
template <int H, int MODE>
__global__ __launch_bounds__(256, 1) void cross_attn_fused_bwd_kv_kernel(const FabindAttnFusedBwdArgs p) {
    constexpr int LDB1 = (CF_KEYS + 1) * 4;
    const int nsplit = p.nsplit;
    const int cplx = (int)(blockIdx.x >> 3) / nsplit * 8 + (int)(blockIdx.x & 7);
    const int rtile = (int)(blockIdx.x >> 3) % nsplit;
    if (cplx >= p.B) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sA = (bf16_t*)smem;
    float* sB = (float*)smem;
    bf16_t* sG = (bf16_t*)(smem + 65536);                           // (written by the shared bias phase; unused here)
    unsigned char* sQ = smem + 65536 + CFB_SG_BYTES;
    unsigned char* sDO = sQ + CM_KC * 256;
    unsigned char* sQt = sDO + CM_KC * 256;
    unsigned char* sDOt = sQt + 128 * 64;
    float* sL = (float*)(sDOt + 128 * 64);
    float* sD = sL + CM_KC * 4;
    const int* ds = p.desc + cplx * 8;
    const int prow0 = ds[0], P = ds[1], lig0 = ds[2], C = ds[3];
    const int m0 = rtile * CF_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    if (m0 >= P) {
        if (MODE == 0)                                              // neutral d k | d v partials of a row tile past the end of this complex
            for (int t = tid; t < C * 64; t += 256)
                *(float4*)(p.part + (((size_t)rtile * p.B + cplx) * p.part_rows + (t >> 6)) * 256 + (t & 63) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int nrow = min(CF_ROWS, P - m0);
    const int qe = MODE == 0 ? nrow : C;                            // queries of this work-group
    const long q0 = MODE == 0 ? (long)prow0 + m0 : (long)lig0;
    auto stage_q = [&](int i0) {
        cfb_stage<2>(sQ, sDO, sQt, sDOt, p.qg, p.ldq, p.scale, p.dO, 128, q0 + i0, qe - i0, tid);
        if (tid < CM_KC * 4) {
            const int ir = tid >> 2;
            sL[tid] = (i0 + ir < qe) ? p.lse[(size_t)(q0 + i0 + ir) * 4 + (tid & 3)] : 0.f;
            sD[tid] = (i0 + ir < qe) ? p.Dv[(size_t)(q0 + i0 + ir) * 4 + (tid & 3)] : 0.f;
        }
    };
    stage_q(0);                                                     // chunk 0, under the bias phase's latency
    cfb_bias_tile<H, MODE>(p, cplx, prow0, m0, nrow, sA, sB, sG, tid);
    const int nk_total = MODE == 0 ? C : nrow;                      // keys this work-group serves
    const int kj = wave * 16 + n;
    const bool kvalid = kj < nk_total;
    const size_t krow = MODE == 0 ? (size_t)(lig0 + (kvalid ? kj : 0)) : (size_t)(prow0 + m0 + (kvalid ? kj : 0));
    const float* kv = p.kv;
    const int ldkv = p.ldkv;
    bf16x8_t bk[4], bv[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float fk[8], fv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 a = *(const float4*)(kv + krow * ldkv + h * 32 + kq * 8 + half * 4);
            const float4 b = *(const float4*)(kv + krow * ldkv + 128 + h * 32 + kq * 8 + half * 4);
            fk[half * 4] = a.x; fk[half * 4 + 1] = a.y; fk[half * 4 + 2] = a.z; fk[half * 4 + 3] = a.w;
            fv[half * 4] = b.x; fv[half * 4 + 1] = b.y; fv[half * 4 + 2] = b.z; fv[half * 4 + 3] = b.w;
        }
        if (!kvalid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { fk[u] = 0.f; fv[u] = 0.f; }
        }
        bk[h] = cm_pack8(fk);
        bv[h] = cm_pack8(fv);
    }
    f32x4_t dk[4][2], dv[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) { dk[h][mi] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[h][mi] = dk[h][mi]; }
    for (int i0 = 0; i0 < qe; i0 += CM_KC) {
        __syncthreads();                                            // (chunk 0: orders the bias tile writes before their reads)
        if (i0 > 0) {
            stage_q(i0);
            __syncthreads();
        }
        float4 bia[2][4];
        bool ok[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + t * 16 + kq * 4 + r;
                ok[t][r] = kvalid && i < qe;
                const int ic = min(i, qe - 1), kc = kvalid ? kj : 0;
                // bias of (query, key): MODE 0 tile [atom = key][row = query], MODE 1 tile [row = key][atom = query]
                bia[t][r] = *(const float4*)&sB[MODE == 0 ? (kc * CF_ROWS + ic) * 4 : kc * LDB1 + ic * 4];
            }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sQ, t * 16 + n, h * 4 + kq), bk[h],
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_rows(sDO, t * 16 + n, h * 4 + kq), bv[h],
                                                                 f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            float pv[8], dsv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = t * 16 + kq * 4 + r;
                    const float pr = ok[t][r] ? __expf(s[t][r] + ((const float*)&bia[t][r])[h] - sL[il * 4 + h]) : 0.f;
                    pv[t * 4 + r] = pr;
                    dsv[t * 4 + r] = pr * (dp[t][r] - sD[il * 4 + h]);
                }
            const bf16x8_t pf = cm_pack8(pv), df = cm_pack8(dsv);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                dv[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sDOt, h * 32 + mi * 16 + n, kq), pf, dv[h][mi], 0, 0, 0);
                dk[h][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm_load_cols(sQt, h * 32 + mi * 16 + n, kq), df, dk[h][mi], 0, 0, 0);
            }
        }
    }
    if (!kvalid) return;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int c = h * 32 + mi * 16 + kq * 4;
            const float4 a = make_float4(dk[h][mi][0], dk[h][mi][1], dk[h][mi][2], dk[h][mi][3]);
            const float4 b = make_float4(dv[h][mi][0], dv[h][mi][1], dv[h][mi][2], dv[h][mi][3]);
            if (MODE == 0) {    // partial over this row tile -> part[tile][complex][local key][256] (dk | dv)
                float* pp = p.part + (((size_t)rtile * p.B + cplx) * p.part_rows + kj) * 256;
                *(float4*)(pp + c) = a;
                *(float4*)(pp + 128 + c) = b;
            } else {
                *(float4*)(p.dkv + krow * ldkv + c) = a;
                *(float4*)(p.dkv + krow * ldkv + 128 + c) = b;
            }
        }
}

// d q of the ligand-side queries: sum of the row tiles' partials part[(lig row * nsplit + tile) * 128 + c]
__global__ __launch_bounds__(256) void cfb_reduce_dq_kernel(const float* __restrict__ part, int nsplit, int n_rows, float* __restrict__ dqg, int ldq) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    if (r >= n_rows) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
        const float4 v = *(const float4*)(part + ((size_t)r * nsplit + z) * 128 + c4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(dqg + (size_t)r * ldq + c4) = acc;
}

extern "C" long fabind_cross_attn_fused_bwd_scratch(int B, int max_P, int max_C, int n_lig_rows, int mode) {
    const long nsplit = (max_P + CF_ROWS - 1) / CF_ROWS;
    return mode == 0 ? nsplit * B * (long)max_C * 256 : (long)n_lig_rows * nsplit * 128;
}
extern "C" int fabind_cross_attn_fused_bwd_parts(int B, int max_P) { return B * ((max_P + CF_ROWS - 1) / CF_ROWS); }

extern "C" int fabind_cross_attn_fused_bwd(const FabindAttnFusedBwdArgs* a_in, int B, int max_P, int max_C, int H, int mode,
                                           int n_lig_rows, hipStream_t stream) {
    FabindAttnFusedBwdArgs a = *a_in;
    if (B <= 0 || max_P <= 0) return 0;
    FB_REQUIRE(max_C <= CF_KEYS, "fabind_cross_attn_fused_bwd: at most 62 ligand-side nodes per complex");
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_cross_attn_fused_bwd: H must be 64, 128, 256 or 512");
    FB_REQUIRE(a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.lda0 % 8 == 0 && a.ldda0 % 4 == 0 && a.ldacat % 8 == 0 && a.kcol0 % 8 == 0 &&
               a.kp % 8 == 0 && a.kp >= max_C * 8 && a.kcol0 + a.kp <= a.ldacat, "fabind_cross_attn_fused_bwd: strides / acat block width kp");
    FB_REQUIRE(a.part != nullptr && a.colpart != nullptr && a.acat != nullptr && (a.da0 == nullptr || a.boT != nullptr) && a.ldcolpart >= 8,
               "fabind_cross_attn_fused_bwd: scratch / operand pointers");
    a.nsplit = (max_P + CF_ROWS - 1) / CF_ROWS;
    a.B = B;
    a.part_rows = max_C;
    const int grid = 8 * a.nsplit * ((B + 7) / 8);
    const size_t lds_q = 65536 + CFB_SG_BYTES + 2 * CM_KC * 256 + 128 * 64;
    const size_t lds_kv = 65536 + CFB_SG_BYTES + 2 * CM_KC * 256 + 2 * 128 * 64 + 2 * CM_KC * 4 * sizeof(float);
#define CFB_LAUNCH(HH, MM)                                                                                                         \
    do {                                                                                                                           \
        static bool set_ = false;                                                                                                  \
        if (!set_) {                                                                                                               \
            (void)hipFuncSetAttribute((const void*)cross_attn_fused_bwd_q_kernel<HH, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);   \
            (void)hipFuncSetAttribute((const void*)cross_attn_fused_bwd_kv_kernel<HH, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv); \
            set_ = true;                                                                                                           \
        }                                                                                                                          \
        hipLaunchKernelGGL((cross_attn_fused_bwd_q_kernel<HH, MM>), dim3(grid), dim3(256), lds_q, stream, a);                      \
        if (MM == 1)                                                                                                               \
            hipLaunchKernelGGL(cfb_reduce_dq_kernel, dim3((n_lig_rows * 32 + 255) / 256), dim3(256), 0, stream, a.part, a.nsplit,  \
                               n_lig_rows, a.dqg, a.ldq);                                                                          \
        hipLaunchKernelGGL((cross_attn_fused_bwd_kv_kernel<HH, MM>), dim3(grid), dim3(256), lds_kv, stream, a);                    \
        if (MM == 0)                                                                                                               \
            hipLaunchKernelGGL(cross_attn_mfma_split_reduce_kernel, dim3((max_C * 256 + 255) / 256, B), dim3(256), 0, stream, a.part, \
                               a.nsplit, max_C, 256, a.desc, 2, 3, a.dkv, a.ldkv);                                                 \
    } while (0)
#define CFB_LAUNCH_H(MM) do { if (H == 512) CFB_LAUNCH(512, MM); else if (H == 256) CFB_LAUNCH(256, MM); else if (H == 128) CFB_LAUNCH(128, MM); else CFB_LAUNCH(64, MM); } while (0)
    if (mode == 0) CFB_LAUNCH_H(0); else CFB_LAUNCH_H(1);
#undef CFB_LAUNCH_H
#undef CFB_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}
