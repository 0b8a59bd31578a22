// Forward-only node-level chains: Linear -> activation -> Linear (+ residual) over 64-row tiles with the hidden activation kept in LDS.
//
//   kind 0 (the node MLP of MC_E_GCL, egnn.py:99-109):          out = act([X1 | X2] W1^T + b1) W2^T + b2 + R      (hidden = H)
//   kind 1 (the Transition blocks, model_utils.py:162-175):     out = act(X1 W1^T + b1) W2^T + b2 + R             (hidden = 2 H)
//
// As two GEMM launches the hidden activation is written to HBM and read back (100 MB each way at 98,688 rows x 512, 200 MB each way for
// the 1,024-wide Transition): a fifth to two fifths of the pair's traffic, on launches that are HBM-bound.  Here a work-group of H / 64
// waves owns 64 rows: X tiles are staged into ONE swizzled [64][H] bf16 LDS tile (two work-groups per CU), every H x H weight block is a
// fragment-packed operand streamed from L2 (fused_common.h: fe_gemm_rolled, the operand-swapped contraction of the fused edge kernels),
// the hidden activation overwrites the tile in place, and only the fp32 output (+ its bf16 operand copy) leaves the CU.
// Inference only: under autograd the hidden activation and its gradient are the operands of the weight-gradient contractions and have
// to exist in HBM anyway.
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

typedef float nc_f2 __attribute__((ext_vector_type(2)));
template <int ACT> __device__ __forceinline__ nc_f2 nc_act_pair(const nc_f2 z) {
    if constexpr (ACT == FB_ACT_SILU) {
        const nc_f2 t = z * -1.44269504f;
        const nc_f2 o = nc_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
        return z * nc_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
    } else {
        return nc_f2{fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)};
    }
}

struct NodeChainArgs {
    const bf16_t* X1; int ld1;
    const bf16_t* X2; int ld2;        // kind 0 only
    const bf16_t* W1a; const bf16_t* W1b; const float* b1;        // kind 0: the two K-halves of W1; kind 1: its two hidden halves
    const bf16_t* W2a; const bf16_t* W2b; const float* b2;        // kind 0: W2 (W2b unused); kind 1: the two K-halves of W2
    const float* R; int ldr;
    float* out; int ldo;
    bf16_t* out16; int ldo16;
    int M;
};

template <int H, int KIND, int ACT>
__global__ __launch_bounds__(H, ((KIND == 0 && H == 512) ? 4 : 2)) void node_chain_fwd_kernel(const NodeChainArgs p) {
    constexpr int BM = 64, MI = BM / 16, CH = H / 8;
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [64][H] swizzled: X1 -> (X2) -> hidden activation
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nr = min(BM, p.M - m0);
    // LDS element offset of this lane's accumulator quad (i, j): row i*16 + fr, columns wave*64 + j*16 + cq*4 .. +3 (fused_edge_fwd2.hip)
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
#define NC_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    const int fcol = wave * 64 + cq * 4;

    constexpr int NLD = BM * CH / H;                              // 16-byte chunks per thread and tile (= 8)
    auto stage = [&](const bf16_t* X, int ld) {                   // every load of the tile in flight before the first LDS store;
        uint4 v[NLD];                                             // rows past the end are zero rows
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int q = tid + u * H, rw = q / CH, ch = q % CH;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (rw < nr) v[u] = *(const uint4*)(X + (size_t)(m0 + rw) * ld + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int q = tid + u * H, rw = q / CH, ch = q % CH;
            *(uint4*)&sX[rw * H + ((ch ^ (rw & SWZ)) * 8)] = v[u];
        }
    };
    auto hidden_to_tile = [&](f32x4_t (&acc)[MI][4], const float* bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bq = *(const float4*)(bias + fcol + j * 16);
            const nc_f2 b01 = nc_f2{bq.x, bq.y}, b23 = nc_f2{bq.z, bq.w};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const nc_f2 t01 = nc_act_pair<ACT>(nc_f2{acc[i][j][0], acc[i][j][1]} + b01);
                const nc_f2 t23 = nc_act_pair<ACT>(nc_f2{acc[i][j][2], acc[i][j][3]} + b23);
                *(uint2*)&sX[NC_QOFF(i, j)] = make_uint2(pack2_bf16(t01.x, t01.y), pack2_bf16(t23.x, t23.y));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4_t acc[MI][4], out[MI][4];
    fe_zero(out);
    stage(p.X1, p.ld1);
    __syncthreads();
    if constexpr (KIND == 0) {
        fe_zero(acc);
        fe_gemm_rolled<H, MI, true, SWZ>(sX, p.W1a, wave, lane, acc);
        __syncthreads();
        stage(p.X2, p.ld2);
        __syncthreads();
        fe_gemm_rolled<H, MI, true, SWZ>(sX, p.W1b, wave, lane, acc);
        __syncthreads();                                          // every wave has finished reading X2
        hidden_to_tile(acc, p.b1);
        __syncthreads();
        fe_gemm_rolled<H, MI, true, SWZ>(sX, p.W2a, wave, lane, out);
    } else {
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            if (c) {                                              // the tile held the first hidden half: X1 comes back from L2
                __syncthreads();
                stage(p.X1, p.ld1);
                __syncthreads();
            }
            fe_zero(acc);
            fe_gemm_rolled<H, MI, true, SWZ>(sX, c ? p.W1b : p.W1a, wave, lane, acc);
            __syncthreads();
            hidden_to_tile(acc, p.b1 + c * H);
            __syncthreads();
            fe_gemm_rolled<H, MI, true, SWZ>(sX, c ? p.W2b : p.W2a, wave, lane, out);
        }
    }
    // out = acc + b2 (+ R), fp32 (+ bf16 copy).  A lane's accumulator quad is four consecutive columns of one row: stored from the
    // registers that is 64-byte pieces of sixteen rows per instruction (the two-GEMM form's epilogue moves 256-byte row segments and was
    // as fast as this whole kernel).  The tile is dead after the last contraction: the fp32 result goes through it in two column halves
    // ([64][H / 2] fp32 = the tile's 64 x H x 2 bytes), and leaves as whole rows -- a wave per row, 16 bytes per lane.
    {
        float* sO = (float*)smem;                                 // [64][H / 2] fp32, row-major
        constexpr int HW = H / 2;                                 // columns per half
        constexpr int WPH = (H / 64) / 2;                         // waves whose columns lie in one half
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                      // the contraction (or the previous half's readers) are done with the tile
            if (wave / WPH == half) {
                const int lc = (wave % WPH) * 64 + cq * 4;        // this lane's first column inside the half
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 bq = *(const float4*)(p.b2 + fcol + j * 16);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        *(float4*)&sO[(i * 16 + fr) * HW + lc + j * 16] =
                            make_float4(out[i][j][0] + bq.x, out[i][j][1] + bq.y, out[i][j][2] + bq.z, out[i][j][3] + bq.w);
                }
            }
            __syncthreads();
            constexpr int C4 = HW / 4;                            // float4 chunks per row of the half
            for (int q = tid; q < BM * C4; q += H) {
                const int rw = q / C4, c4 = q % C4;
                if (rw >= nr) continue;
                const size_t row = (size_t)(m0 + rw);
                const int col = half * HW + c4 * 4;
                float4 o = *(const float4*)&sO[rw * HW + c4 * 4];
                if (p.R) {
                    const float4 r = *(const float4*)(p.R + row * p.ldr + col);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *(float4*)(p.out + row * p.ldo + col) = o;
                if (p.out16) *(uint2*)(p.out16 + row * p.ldo16 + col) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));
            }
        }
    }
#undef NC_QOFF
}

extern "C" int fabind_node_chain_fwd(const void* X1, int ld1, const void* X2, int ld2, const void* W1a, const void* W1b, const float* b1,
                                     const void* W2a, const void* W2b, const float* b2, int act, int kind, const float* R, int ldr,
                                     float* out, int ldo, void* out16, int ldo16, int M, int H, hipStream_t stream) {
    if (M <= 0) return 0;
    FB_REQUIRE(H == 512 || H == 256 || H == 128, "fabind_node_chain_fwd: H in {128, 256, 512}");
    FB_REQUIRE(kind == 0 || kind == 1, "fabind_node_chain_fwd: kind 0 (K = 2H, hidden H) or 1 (K = H, hidden 2H)");
    FB_REQUIRE(act == FB_ACT_SILU || act == FB_ACT_RELU, "fabind_node_chain_fwd: silu or relu");
    FB_REQUIRE(X1 && W1a && W1b && W2a && b1 && b2 && out && (kind == 0 ? X2 != nullptr : W2b != nullptr), "fabind_node_chain_fwd: operands");
    FB_REQUIRE(ld1 % 8 == 0 && (kind == 1 || ld2 % 8 == 0) && ldo % 4 == 0 && (!R || ldr % 4 == 0) && (!out16 || ldo16 % 4 == 0),
               "fabind_node_chain_fwd: leading dimensions (bf16 rows % 8, fp32 rows % 4)");
    FB_REQUIRE((((uintptr_t)X1 | (uintptr_t)X2 | (uintptr_t)out | (uintptr_t)R | (uintptr_t)out16 | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0,
               "fabind_node_chain_fwd: 16-byte aligned buffers");
    NodeChainArgs a;
    a.X1 = (const bf16_t*)X1; a.ld1 = ld1; a.X2 = (const bf16_t*)X2; a.ld2 = ld2;
    a.W1a = (const bf16_t*)W1a; a.W1b = (const bf16_t*)W1b; a.b1 = b1;
    a.W2a = (const bf16_t*)W2a; a.W2b = (const bf16_t*)W2b; a.b2 = b2;
    a.R = R; a.ldr = ldr; a.out = out; a.ldo = ldo; a.out16 = (bf16_t*)out16; a.ldo16 = ldo16; a.M = M;
    const dim3 grid((M + 63) / 64);
#define NC_LAUNCH_(HH, KK, AA)                                                                                                  \
    do {                                                                                                                        \
        const size_t lds = (size_t)64 * HH * 2;                                                                                 \
        static bool set_ = false;                                                                                               \
        if (!set_) { (void)hipFuncSetAttribute((const void*)node_chain_fwd_kernel<HH, KK, AA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((node_chain_fwd_kernel<HH, KK, AA>), grid, dim3(HH), lds, stream, a);                                 \
    } while (0)
#define NC_LAUNCH(HH)                                                                                                           \
    do {                                                                                                                        \
        if (kind == 0) { if (act == FB_ACT_SILU) NC_LAUNCH_(HH, 0, FB_ACT_SILU); else NC_LAUNCH_(HH, 0, FB_ACT_RELU); }          \
        else { if (act == FB_ACT_SILU) NC_LAUNCH_(HH, 1, FB_ACT_SILU); else NC_LAUNCH_(HH, 1, FB_ACT_RELU); }                    \
    } while (0)
    if (H == 512) NC_LAUNCH(512); else if (H == 256) NC_LAUNCH(256); else NC_LAUNCH(128);
#undef NC_LAUNCH
#undef NC_LAUNCH_
    FB_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the SPLIT-PRECISION form of the chain (bf16 mode, config.set_split_sites(3): both Linears of a node MLP / Transition contract
// fp32 activations with fp32 master weights as split bf16, three MFMAs per product term, and the hidden layer never gets rounded to
// bf16).  fp32 X rows are split while they are staged into TWO swizzled [64][H] bf16 planes (hi = bf16(x), lo = bf16(x - hi)), the
// weights arrive as hi | lo fragment packs (kernels.pack_frag_split), fe_gemm_x3 (fused_common.h) contracts, the fp32 hidden accumulators
// get bias + activation and are split back into the planes in place.  One work-group of H / 64 waves per 64 rows, 2 x 64 x H x 2 bytes of
// LDS (128 KiB at H = 512: one work-group per CU).  As two gemm_x3 launches the fp32 hidden layer travels through HBM (404 MB each way for
// the Transition at 98,688 rows) and every k-tile of both operands is split in every work-group.
// ------------------------------------------------------------------------------------------------------------------------------------------
struct NodeChainX3Args {
    const float* X1; int ld1;
    const float* X2; int ld2;
    const bf16_t* W1ah; const bf16_t* W1al; const bf16_t* W1bh; const bf16_t* W1bl; const float* b1;
    const bf16_t* W2ah; const bf16_t* W2al; const bf16_t* W2bh; const bf16_t* W2bl; const float* b2;
    const float* R; int ldr;
    float* out; int ldo;
    bf16_t* out16; int ldo16;
    int M;
    bf16_t* T16; int ldt;             // SAVE (training forward): bf16 of the activated hidden layer [M, hidden] ...
    bf16_t* D16; int ldd;             // ... and of the activation derivative (SiLU only; NULL for ReLU): the operands of the unchanged bf16 backward
};

__device__ __forceinline__ void nc_split4(const float4 x, uint2& hi, uint2& lo) {
    const uint32_t h0 = pack2_bf16(x.x, x.y), h1 = pack2_bf16(x.z, x.w);
    hi = make_uint2(h0, h1);
    lo = make_uint2(pack2_bf16(x.x - __uint_as_float(h0 << 16), x.y - __uint_as_float(h0 & 0xffff0000u)),
                    pack2_bf16(x.z - __uint_as_float(h1 << 16), x.w - __uint_as_float(h1 & 0xffff0000u)));
}

template <int ACT> __device__ __forceinline__ nc_f2 nc_dact_pair(const nc_f2 z) {     // act'(z), SiLU: s (1 + z (1 - s))
    const nc_f2 t = z * -1.44269504f;
    const nc_f2 o = nc_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    const nc_f2 sg = nc_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
    return sg * (1.0f + z * (1.0f - sg));
}

template <int H, int KIND, int ACT, bool SAVE = false>
__global__ __launch_bounds__(H, 1) void node_chain_x3_kernel(const NodeChainX3Args p) {
    constexpr int BM = 64, MI = BM / 16, CH = H / 8;
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sXh = (bf16_t*)smem;                                  // [64][H] swizzled, hi plane
    bf16_t* sXl = sXh + BM * H;                                   // lo plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nr = min(BM, p.M - m0);
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
#define NC_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    const int fcol = wave * 64 + cq * 4;

    constexpr int NLD = BM * CH / H;                              // 8-element chunks per thread and tile (= 8): two float4 loads each
    auto stage = [&](const float* X, int ld) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                    // two rounds of four chunks: 8 float4 loads in flight per thread
            float4 v[NLD / 2][2];
#pragma unroll
            for (int u = 0; u < NLD / 2; ++u) {
                const int q = tid + (half * (NLD / 2) + u) * H, rw = q / CH, ch = q % CH;
                v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rw < nr) {
                    const float* src = X + (size_t)(m0 + rw) * ld + ch * 8;
                    v[u][0] = *(const float4*)src;
                    v[u][1] = *(const float4*)(src + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < NLD / 2; ++u) {
                const int q = tid + (half * (NLD / 2) + u) * H, rw = q / CH, ch = q % CH;
                uint2 h0, l0, h1, l1;
                nc_split4(v[u][0], h0, l0);
                nc_split4(v[u][1], h1, l1);
                const int o = rw * H + ((ch ^ (rw & SWZ)) * 8);
                *(uint4*)&sXh[o] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                *(uint4*)&sXl[o] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    };
    auto hidden_to_tile = [&](f32x4_t (&acc)[MI][4], const float* bias, int hcol0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bq = *(const float4*)(bias + fcol + j * 16);
            const nc_f2 b01 = nc_f2{bq.x, bq.y}, b23 = nc_f2{bq.z, bq.w};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const nc_f2 z01 = nc_f2{acc[i][j][0], acc[i][j][1]} + b01, z23 = nc_f2{acc[i][j][2], acc[i][j][3]} + b23;
                const nc_f2 t01 = nc_act_pair<ACT>(z01);
                const nc_f2 t23 = nc_act_pair<ACT>(z23);
                uint2 hi, lo;
                nc_split4(make_float4(t01.x, t01.y, t23.x, t23.y), hi, lo);
                *(uint2*)&sXh[NC_QOFF(i, j)] = hi;
                *(uint2*)&sXl[NC_QOFF(i, j)] = lo;
                if constexpr (SAVE && ACT == FB_ACT_SILU) {       // the stored derivative of the backward's activation adjoint: four columns of one row
                    const int rw = i * 16 + fr;
                    if (rw < nr && p.D16) {
                        const nc_f2 d01 = nc_dact_pair<ACT>(z01), d23 = nc_dact_pair<ACT>(z23);
                        *(uint2*)(p.D16 + (size_t)(m0 + rw) * p.ldd + hcol0 + fcol + j * 16) = make_uint2(pack2_bf16(d01.x, d01.y), pack2_bf16(d23.x, d23.y));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // SAVE: the hi plane IS bf16(hidden) -- copied out as whole rows (un-swizzled) once every wave has written its columns
    auto flush_hidden = [&](int hcol0) {
        if constexpr (SAVE) {
            for (int q = tid; q < BM * CH; q += H) {
                const int rw = q / CH, ch = q % CH;
                if (rw < nr) *(uint4*)(p.T16 + (size_t)(m0 + rw) * p.ldt + hcol0 + ch * 8) = *(const uint4*)&sXh[rw * H + ((ch ^ (rw & SWZ)) * 8)];
            }
        }
    };

    f32x4_t acc[MI][4], out[MI][4];
    fe_zero(out);
    stage(p.X1, p.ld1);
    __syncthreads();
    if constexpr (KIND == 0) {
        fe_zero(acc);
        fe_gemm_x3<H, MI, SWZ>(sXh, sXl, p.W1ah, p.W1al, wave, lane, acc);
        __syncthreads();
        stage(p.X2, p.ld2);
        __syncthreads();
        fe_gemm_x3<H, MI, SWZ>(sXh, sXl, p.W1bh, p.W1bl, wave, lane, acc);
        __syncthreads();
        hidden_to_tile(acc, p.b1, 0);
        __syncthreads();
        flush_hidden(0);
        fe_gemm_x3<H, MI, SWZ>(sXh, sXl, p.W2ah, p.W2al, wave, lane, out);
    } else {
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            if (c) {
                __syncthreads();
                stage(p.X1, p.ld1);
                __syncthreads();
            }
            fe_zero(acc);
            fe_gemm_x3<H, MI, SWZ>(sXh, sXl, c ? p.W1bh : p.W1ah, c ? p.W1bl : p.W1al, wave, lane, acc);
            __syncthreads();
            hidden_to_tile(acc, p.b1 + c * H, c * H);
            __syncthreads();
            flush_hidden(c * H);
            fe_gemm_x3<H, MI, SWZ>(sXh, sXl, c ? p.W2bh : p.W2ah, c ? p.W2bl : p.W2al, wave, lane, out);
        }
    }
    {   // out = acc + b2 (+ R): as in node_chain_fwd_kernel, through the (dead) planes in two column halves, whole rows out
        float* sO = (float*)smem;
        constexpr int HW = H / 2;
        constexpr int WPH = (H / 64) / 2;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if (wave / WPH == half) {
                const int lc = (wave % WPH) * 64 + cq * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 bq = *(const float4*)(p.b2 + fcol + j * 16);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        *(float4*)&sO[(i * 16 + fr) * HW + lc + j * 16] =
                            make_float4(out[i][j][0] + bq.x, out[i][j][1] + bq.y, out[i][j][2] + bq.z, out[i][j][3] + bq.w);
                }
            }
            __syncthreads();
            constexpr int C4 = HW / 4;
            for (int q = tid; q < BM * C4; q += H) {
                const int rw = q / C4, c4 = q % C4;
                if (rw >= nr) continue;
                const size_t row = (size_t)(m0 + rw);
                const int col = half * HW + c4 * 4;
                float4 o = *(const float4*)&sO[rw * HW + c4 * 4];
                if (p.R) {
                    const float4 r = *(const float4*)(p.R + row * p.ldr + col);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *(float4*)(p.out + row * p.ldo + col) = o;
                if (p.out16) *(uint2*)(p.out16 + row * p.ldo16 + col) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));
            }
        }
    }
#undef NC_QOFF
}

extern "C" int fabind_node_chain_x3_fwd(const float* X1, int ld1, const float* X2, int ld2, const void* W1ah, const void* W1al,
                                        const void* W1bh, const void* W1bl, const float* b1, const void* W2ah, const void* W2al,
                                        const void* W2bh, const void* W2bl, const float* b2, int act, int kind, const float* R, int ldr,
                                        float* out, int ldo, void* out16, int ldo16, void* t16, int ldt, void* d16, int ldd, int M, int H,
                                        hipStream_t stream) {
    if (M <= 0) return 0;
    FB_REQUIRE(H == 512 || H == 256 || H == 128, "fabind_node_chain_x3_fwd: H in {128, 256, 512}");
    FB_REQUIRE(!t16 || (ldt % 8 == 0 && ((uintptr_t)t16 & 15) == 0 && (!d16 || (ldd % 4 == 0 && ((uintptr_t)d16 & 7) == 0))),
               "fabind_node_chain_x3_fwd: saved hidden tile: ldt % 8, 16-byte aligned; derivative tile: ldd % 4, 8-byte aligned");
    FB_REQUIRE(!d16 || (t16 && act == FB_ACT_SILU), "fabind_node_chain_x3_fwd: the derivative tile goes with a saved SiLU hidden tile");
    FB_REQUIRE(kind == 0 || kind == 1, "fabind_node_chain_x3_fwd: kind 0 (K = 2H, hidden H) or 1 (K = H, hidden 2H)");
    FB_REQUIRE(act == FB_ACT_SILU || act == FB_ACT_RELU, "fabind_node_chain_x3_fwd: silu or relu");
    FB_REQUIRE(X1 && W1ah && W1al && W1bh && W1bl && W2ah && W2al && b1 && b2 && out && (kind == 0 ? X2 != nullptr : (W2bh && W2bl)),
               "fabind_node_chain_x3_fwd: operands");
    FB_REQUIRE(ld1 % 4 == 0 && (kind == 1 || ld2 % 4 == 0) && ldo % 4 == 0 && (!R || ldr % 4 == 0) && (!out16 || ldo16 % 4 == 0),
               "fabind_node_chain_x3_fwd: leading dimensions % 4");
    FB_REQUIRE((((uintptr_t)X1 | (uintptr_t)X2 | (uintptr_t)out | (uintptr_t)R | (uintptr_t)out16 | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0,
               "fabind_node_chain_x3_fwd: 16-byte aligned buffers");
    NodeChainX3Args a;
    a.X1 = X1; a.ld1 = ld1; a.X2 = X2; a.ld2 = ld2;
    a.W1ah = (const bf16_t*)W1ah; a.W1al = (const bf16_t*)W1al; a.W1bh = (const bf16_t*)W1bh; a.W1bl = (const bf16_t*)W1bl; a.b1 = b1;
    a.W2ah = (const bf16_t*)W2ah; a.W2al = (const bf16_t*)W2al; a.W2bh = (const bf16_t*)W2bh; a.W2bl = (const bf16_t*)W2bl; a.b2 = b2;
    a.R = R; a.ldr = ldr; a.out = out; a.ldo = ldo; a.out16 = (bf16_t*)out16; a.ldo16 = ldo16; a.M = M;
    a.T16 = (bf16_t*)t16; a.ldt = ldt; a.D16 = (bf16_t*)d16; a.ldd = ldd;
    const dim3 grid((M + 63) / 64);
#define NCX_LAUNCH__(HH, KK, AA, SS)                                                                                            \
    do {                                                                                                                        \
        const size_t lds = (size_t)2 * 64 * HH * 2;                                                                             \
        static bool set_ = false;                                                                                               \
        if (!set_) { (void)hipFuncSetAttribute((const void*)node_chain_x3_kernel<HH, KK, AA, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((node_chain_x3_kernel<HH, KK, AA, SS>), grid, dim3(HH), lds, stream, a);                              \
    } while (0)
#define NCX_LAUNCH_(HH, KK, AA) do { if (t16) NCX_LAUNCH__(HH, KK, AA, true); else NCX_LAUNCH__(HH, KK, AA, false); } while (0)
#define NCX_LAUNCH(HH)                                                                                                          \
    do {                                                                                                                        \
        if (kind == 0) { if (act == FB_ACT_SILU) NCX_LAUNCH_(HH, 0, FB_ACT_SILU); else NCX_LAUNCH_(HH, 0, FB_ACT_RELU); }        \
        else { if (act == FB_ACT_SILU) NCX_LAUNCH_(HH, 1, FB_ACT_SILU); else NCX_LAUNCH_(HH, 1, FB_ACT_RELU); }                  \
    } while (0)
    if (H == 512) NCX_LAUNCH(512); else if (H == 256) NCX_LAUNCH(256); else NCX_LAUNCH(128);
#undef NCX_LAUNCH
#undef NCX_LAUNCH_
#undef NCX_LAUNCH__
    FB_CHECK_LAUNCH();
    return 0;
}
