// Fused pair-embedding update of FABind+'s CrossAttentionModule (reference FABind_plus/fabind/models/cross_att.py:42-44,
// model_utils.py MLPwithLastAct), bf16 inference path, one work-group per tile of 64 consecutive pairs:
//
//   z1  = z + (a32[p] * b32[c]) Wo^T + bo                      (InteractionModule, 32-wide Hadamard + linear_out)
//   y   = LayerNorm(z1) * ln_w + ln_b                          (fp32 statistics of the unrounded z1)
//   t   = drop( relu( y  W1^T + b1 ) )
//   z'  = drop( relu( t  W2^T + b2 ) )                         -> z_out (bf16)
//   bias' = z' Wb^T + bb                                        (optional: the NEXT layer's 16 row-attention biases)
//
// The [pairs, H] tensors z1, y and t never leave the CU: they live in one swizzled 64 KiB LDS tile.  W1 / W2 stream
// from L2 straight into MFMA B fragments (pre-packed in fragment order), one wave owns 64 output columns.  HBM traffic
// per pair: H bf16 in + H bf16 out + 64 B of biases, instead of the seven [pairs, H] round trips of the separate
// Hadamard / GEMM / LayerNorm / GEMM / GEMM / GEMM launches.  Two work-groups per CU (72 KiB LDS, <= 128 VGPRs).
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

#define PF_ROT4(a_) { auto t_ = a_[0]; a_[0] = a_[1]; a_[1] = a_[2]; a_[2] = a_[3]; a_[3] = t_; }

// sum over the 16 lanes that share a C-fragment row (fr = lane & 15)
__device__ __forceinline__ float pf_rowsum16(float t) {
    t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
    return t;
}

template <int H>
__global__ __launch_bounds__(H, 4) void pair_update_fused_kernel(const FabindPairUpdateArgs p) {
    constexpr int NW = H / 64;                       // waves; wave w owns output columns [64w, 64w+64)
    constexpr int CH = H / 8;                        // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                      // [64][H] bf16, 16-B chunk index XOR (row & 7)
    bf16_t* sH = sX + FE_BM * H;                     // [64][32] bf16: the Hadamard block of the tile
    float* sStat = (float*)(sH + FE_BM * 32);        // [2][NW][64]: per-wave row partials (sum, then squared deviations)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int e0 = blockIdx.x * FE_BM;
    const int ne = min(FE_BM, p.n_pairs - e0);
    const uint32_t thr16 = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);

    // ---- phase 0a: z tile -> sX (coalesced 16-byte loads), Hadamard block -> sH
    {
        const bf16_t* zt = (const bf16_t*)p.z_in + (size_t)e0 * H;
        for (unsigned q = tid; q < (unsigned)FE_BM * CH; q += H) {
            const unsigned rw = q / CH, ch = q % CH;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((int)rw < ne) v = *(const uint4*)&zt[rw * H + ch * 8];
            *(uint4*)&sX[rw * H + ((ch ^ (rw & 7)) * 8)] = v;
        }
        for (int t = tid; t < FE_BM * 4; t += H) {
            const int pr = t >> 2, ck = t & 3;
            float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (pr < ne) {
                const float* ap = p.T + (size_t)p.p_node[e0 + pr] * p.ldt + ck * 8;
                const float* bp = p.T + (size_t)p.c_node[e0 + pr] * p.ldt + p.b_off + ck * 8;
                const float4 a0 = *(const float4*)ap, a1 = *(const float4*)(ap + 4);
                const float4 b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 4);
                o[0] = a0.x * b0.x; o[1] = a0.y * b0.y; o[2] = a0.z * b0.z; o[3] = a0.w * b0.w;
                o[4] = a1.x * b1.x; o[5] = a1.y * b1.y; o[6] = a1.z * b1.z; o[7] = a1.w * b1.w;
            }
            *(uint4*)&sH[pr * 32 + ck * 8] = pack8(o);
        }
    }
    __syncthreads();

    // ---- phase 0b: acc = hd Wo^T (one k-step), + bo + z; LayerNorm over the H columns (spread over the NW waves)
    f32x4_t acc[4][4];
    fe_zero(acc);
    {
        const bf16x8_t* wp = (const bf16x8_t*)p.Wop + ((size_t)wave * 4) * 64 + lane;
        bf16x8_t b[4], a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = wp[(size_t)j * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)&sH[(i * 16 + fr) * 32 + cq * 8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // both epilogue loops run rolled over the four 16-row blocks (block 0 is processed, the blocks rotate): only the
    // statistics of 4 pair rows are live at a time next to the 64 accumulator registers
#define PF_ROTI(a_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { auto t_ = a_[0][j_]; a_[0][j_] = a_[1][j_]; a_[1][j_] = a_[2][j_]; a_[2][j_] = a_[3][j_]; a_[3][j_] = t_; }
    {
        float* sSq = sStat + NW * FE_BM;
        float bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = p.bo[wave * 64 + j * 16 + fr];
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int colj = wave * 64 + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = i * 16 + cq * 4 + r;
                    const float v = acc[0][j][r] + bj[j] + bf16_to_f32(sX[rw * H + (((colj >> 3) ^ (rw & 7)) * 8) + (colj & 7)]);
                    acc[0][j][r] = v;
                    s1[r] += v;
                    s2[r] += v * v;
                }
            }
            // fp32 sums of x and x^2 over the H columns: the 16 lanes of a fragment row here, the NW waves through LDS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = pf_rowsum16(s1[r]), u = pf_rowsum16(s2[r]);
                if (fr == 0) { sStat[wave * FE_BM + i * 16 + cq * 4 + r] = t; sSq[wave * FE_BM + i * 16 + cq * 4 + r] = u; }
            }
            PF_ROTI(acc)
        }
        __syncthreads();
        float2* sMR = (float2*)(sStat + 2 * NW * FE_BM);          // [64] (mean, rstd) per pair row
        if (tid < FE_BM) {
            float t = 0.f, u = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { t += sStat[w * FE_BM + tid]; u += sSq[w * FE_BM + tid]; }
            const float m = t * (1.0f / H);
            sMR[tid] = make_float2(m, __builtin_amdgcn_rsqf(fmaxf(u * (1.0f / H) - m * m, 0.f) + p.eps));
        }
        __syncthreads();
        float gw[4], gb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { gw[j] = p.ln_w[wave * 64 + j * 16 + fr]; gb[j] = p.ln_b[wave * 64 + j * 16 + fr]; }
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            float2 mr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mr[r] = sMR[i * 16 + cq * 4 + r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int colj = wave * 64 + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = i * 16 + cq * 4 + r;
                    // each lane overwrites exactly the z elements it read above: no cross-lane hazard on sX
                    sX[rw * H + (((colj >> 3) ^ (rw & 7)) * 8) + (colj & 7)] = f32_to_bf16((acc[0][j][r] - mr[r].x) * mr[r].y * gw[j] + gb[j]);
                }
            }
            PF_ROTI(acc)
        }
    }
#undef PF_ROTI
    __syncthreads();

    // ---- phases 1, 2: the two Linear + ReLU (+ dropout) of the transition, tile -> tile through LDS
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        const bf16_t* Wp = (const bf16_t*)(layer == 0 ? p.W1p : p.W2p);
        const float* bias = layer == 0 ? p.b1 : p.b2;
        const uint32_t seed = p.seed + (uint32_t)layer * 0x9e3779b9U;
        fe_zero(acc);
        fe_gemm_rolled<H, 4>(sX, Wp, wave, lane, acc);
        __syncthreads();                             // every wave has finished reading the tile
        float bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = bias[wave * 64 + j * 16 + fr];
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int colj = wave * 64 + j * 16 + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = i * 16 + cq * 4 + r;
                    float m = fmaxf(acc[i][0][r] + bj[0], 0.f);
                    if (thr16) m *= fe_keep(seed, (uint32_t)(e0 + rw), (uint32_t)colj, H, thr16, dscale);
                    sX[rw * H + (((colj >> 3) ^ (rw & 7)) * 8) + (colj & 7)] = f32_to_bf16(m);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) PF_ROT4(acc[i])
            PF_ROT4(bj)
        }
        __syncthreads();
    }

    // ---- phase 3: z' -> HBM; the next layer's 16 bias columns by waves 0..3 (16 pair rows each, full K)
    fe_tile_store<H>(sX, (bf16_t*)p.z_out, e0, ne, tid);
    if (p.Wbp != nullptr)
    for (int rb = wave; rb < 4; rb += NW) {
        f32x4_t c = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bf16x8_t* wb = (const bf16x8_t*)p.Wbp + lane;
        const int r = rb * 16 + fr;
#pragma unroll 4
        for (int ks = 0; ks < H / 32; ++ks) {
            const bf16x8_t a = *(const bf16x8_t*)&sX[r * H + (((ks * 4 + cq) ^ (r & 7)) * 8)];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wb[(size_t)ks * 64], c, 0, 0, 0);
        }
        const float bb = p.bb[fr];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rw = rb * 16 + cq * 4 + q;
            if (rw < ne) p.bias_out[(size_t)(e0 + rw) * 16 + fr] = c[q] + bb;
        }
    }
}

extern "C" int fabind_pair_update_fused(const FabindPairUpdateArgs* a, int H, hipStream_t stream) {
    if (a->n_pairs <= 0) return 0;
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_pair_update_fused: H must be 64, 128, 256 or 512");
    FB_REQUIRE(a->p_drop >= 0.f && a->p_drop < 1.f, "fabind_pair_update_fused: p_drop in [0, 1)");
    FB_REQUIRE(a->ldt % 4 == 0 && a->b_off % 4 == 0, "fabind_pair_update_fused: ldt, b_off must be multiples of 4");
    FB_REQUIRE(a->Wbp == nullptr || (a->bb != nullptr && a->bias_out != nullptr), "fabind_pair_update_fused: bb / bias_out");
    const dim3 grid((a->n_pairs + FE_BM - 1) / FE_BM);
    const size_t lds = (size_t)FE_BM * H * 2 + FE_BM * 32 * 2 + (size_t)2 * (H / 64) * FE_BM * sizeof(float) + FE_BM * 2 * sizeof(float);
#define PF_LAUNCH(HH)                                                                                              \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)pair_update_fused_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((pair_update_fused_kernel<HH>), grid, dim3(HH), lds, stream, *a);                       \
    } while (0)
    if (H == 512) PF_LAUNCH(512); else if (H == 256) PF_LAUNCH(256); else if (H == 128) PF_LAUNCH(128); else PF_LAUNCH(64);
#undef PF_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}
