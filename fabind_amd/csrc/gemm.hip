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
    u.x = (uint32_t)f32_to_bf16(r.v[0]) | ((uint32_t)f32_to_bf16(r.v[1]) << 16);
    u.y = (uint32_t)f32_to_bf16(r.v[2]) | ((uint32_t)f32_to_bf16(r.v[3]) << 16);
    u.z = (uint32_t)f32_to_bf16(r.v[4]) | ((uint32_t)f32_to_bf16(r.v[5]) << 16);
    u.w = (uint32_t)f32_to_bf16(r.v[6]) | ((uint32_t)f32_to_bf16(r.v[7]) << 16);
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

template <typename TA, typename TM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(FabindGemmArgs p) {
    constexpr int LS = LdsStride<TM>::v;
    __shared__ __attribute__((aligned(16))) TM sA[BM * LS];
    __shared__ __attribute__((aligned(16))) TM sB[BN * LS];
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
    const TA* A = (const TA*)p.A + a_row0 * p.lda;
    const TA* A2 = p.A2 ? (const TA*)p.A2 + a_row0 * p.lda2 : nullptr;
    const TM* W = (const TM*)p.W + w_row0 * p.ldw;
    const int K = p.K, K1 = p.K1;

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
                const TA* src = (gk < K1) ? A + (size_t)gm * p.lda + gk : A2 + (size_t)gm * p.lda2 + (gk - K1);
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

    const int nk = (K + BK - 1) / BK;
    fetch(0);
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        stash();
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * BK);
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

    // ---- epilogue: C/D layout of mfma 16x16: col = lane&15, row = (lane>>4)*4 + reg
    const int cq = lane >> 4;
    const bool want_dot = p.dotvec != nullptr;
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i) {
        float ds0 = 0.f, ds1 = 0.f, ds2 = 0.f, ds3 = 0.f;
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j) {
            const f32x4_t a4 = acc[i][j];
            const float av[4] = {a4[0], a4[1], a4[2], a4[3]};
            const int col = n0 + wn * 64 + j * 16 + fr;
            const float dv = (want_dot && col < N) ? p.dotvec[col] : 0.f;
            const float bv = (p.bias && col < N) ? p.bias[w_row0 + col] : 0.f;
#pragma clang loop unroll(full)
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + cq * 4 + r;
                float v = 0.f;
                if (row < M && col < N) {
                    v = av[r] * p.alpha + bv;
                    const float vpre = v;
                    v = apply_act(v, p.act_epi);
                    if (p.aux) v *= apply_dact(ld_any(p.aux, p.aux_dtype, (size_t)(a_row0 + row) * p.ldaux + col), p.dact_epi);
                    if (p.R) {
                        long rr = p.r_index ? (long)p.r_index[a_row0 + row] : (a_row0 + row);
                        v += ((const float*)p.R)[(size_t)rr * p.ldr + col];
                    }
                    if (p.C) {
                        size_t ci = (size_t)c_off + (size_t)row * ldc + col;
                        if (p.accumulate) v += ld_any(p.C, p.c_dtype, ci);
                        st_any(p.C, p.c_dtype, ci, p.store_preact ? vpre : v);
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
                if (fr == 0) sDot[wn][wm * 64 + i * 16 + cq * 4 + r] = dsum;
            }
        }
    }
    if (want_dot) {
        __syncthreads();
        if (tid < BM && m0 + tid < M)
            p.dot_out[(size_t)(a_row0 + m0 + tid) * p.dot_ld + blockIdx.x] = sDot[0][tid] + sDot[1][tid];
    }
}

extern "C" int fabind_gemm(const FabindGemmArgs* args, hipStream_t stream) {
    FabindGemmArgs p = *args;
    FB_REQUIRE(p.K % 8 == 0, "fabind_gemm: K must be a multiple of 8");
    FB_REQUIRE(p.A2 == nullptr || (p.K1 % BK == 0), "fabind_gemm: K1 must be a multiple of 32 when A2 is given");
    if (p.A2 == nullptr) p.K1 = p.K;
    FB_REQUIRE(p.lda % 4 == 0 && p.ldw % 4 == 0, "fabind_gemm: lda/ldw must be multiples of 4");
    FB_REQUIRE(!(p.a_dtype == FB_DT_BF16 && (p.lda % 8 != 0)), "fabind_gemm: bf16 A needs lda % 8 == 0");
    FB_REQUIRE(!(p.w_dtype == FB_DT_BF16 && (p.ldw % 8 != 0)), "fabind_gemm: bf16 W needs ldw % 8 == 0");
    int maxM = p.groups ? p.max_m : p.M, maxN = p.groups ? p.max_n : p.N;
    if (maxM <= 0 || maxN <= 0) return 0;
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, p.groups ? p.n_groups : 1);
    if (p.w_dtype == FB_DT_F32) {
        FB_REQUIRE(p.a_dtype == FB_DT_F32, "fabind_gemm: fp32 MMA needs fp32 A");
        hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, dim3(256), 0, stream, p);
    } else if (p.a_dtype == FB_DT_F32) {
        hipLaunchKernelGGL((gemm_nt_kernel<float, bf16_t>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, p);
    }
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out[C,R] = act(in[R,C])^T   (used to turn dW = dY^T X into NT GEMMs; fused activation on read)
// ------------------------------------------------------------------------------------------------
__global__ void transpose_act_kernel(const void* in, int in_dt, int ldi, void* out, int out_dt, int ldo, int R, int C,
                                     int act) {
    __shared__ float tile[32][33];
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? apply_act(ld_any(in, in_dt, (size_t)r * ldi + c), act) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) st_any(out, out_dt, (size_t)c * ldo + r, tile[tx][i]);
    }
}

extern "C" int fabind_transpose_act(const void* in, int in_dt, int ldi, void* out, int out_dt, int ldo, int R, int C,
                                    int act, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    dim3 grid((C + 31) / 32, (R + 31) / 32);
    hipLaunchKernelGGL(transpose_act_kernel, grid, dim3(256), 0, stream, in, in_dt, ldi, out, out_dt, ldo, R, C, act);
    FB_CHECK_LAUNCH();
    return 0;
}

// column sums of a [R,C] matrix (bias gradients): deterministic two-pass (chunk partials, then fixed-order sum)
__global__ void colsum_part_kernel(const void* in, int in_dt, int ldi, float* scratch, int R, int C, int rows_per) {
    __shared__ float part[4][64];
    int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + q; r < r1; r += 4) s += ld_any(in, in_dt, (size_t)r * ldi + c);
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        int l = threadIdx.x;
        scratch[(size_t)blockIdx.y * C + c] = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
    }
}
__global__ void colsum_final_kernel(const float* scratch, float* out, int C, int nchunk, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += scratch[(size_t)k * C + c];
    out[c] = accumulate ? out[c] + s : s;
}

extern "C" int fabind_colsum(const void* in, int in_dt, int ldi, float* out, int R, int C, int accumulate,
                             float* scratch, int nchunk, hipStream_t stream) {
    if (C <= 0) return 0;
    FB_REQUIRE(nchunk >= 1, "fabind_colsum: nchunk >= 1");
    int rows_per = (R + nchunk - 1) / nchunk;
    if (rows_per < 1) rows_per = 1;
    hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 63) / 64, nchunk), dim3(256), 0, stream, in, in_dt, ldi, scratch, R,
                       C, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, scratch, out, C, nchunk,
                       accumulate);
    FB_CHECK_LAUNCH();
    return 0;
}
