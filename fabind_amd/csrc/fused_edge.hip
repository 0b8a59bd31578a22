// Fused intra-graph edge pipeline of MC_E_GCL (reference models/egnn.py:68-128), forward-only bf16 path:
//
//   S1  = silu( A[row] + Bc[col] + rhohat * w_r )            (first edge_mlp Linear, evaluated per node: AB = h W1'^T)
//   M   = silu( S1 W2^T + b2 )                               (second edge_mlp Linear)      -> messages m_e
//   agg[row] += M                                            (unsorted_segment_sum)
//   s   = w3 . silu( M Wc^T + bc )                           (coord_mlp)                   -> per-edge scalar
//
// for a tile of 64 consecutive (row-sorted) edges per work-group.  The edge tensors S1 and M never
// leave the CU: they live in one swizzled 64 KiB LDS tile; the two H x H weight matrices stream from
// L2 straight into MFMA B-fragments (pre-packed in fragment order, 1 KiB contiguous per wave load), so
// the k-loops need no LDS traffic for weights and no barriers.  One wave owns 64 output columns.
// HBM traffic per edge: one gathered bf16 row (the sender's half of AB) + 12 bytes, instead of four
// [E,H] round trips.  Two work-groups are resident per CU (66 KiB LDS, <= 128 VGPRs).
#include "common.h"
#include "fabind_hip.h"

#define FE_BM 64

__device__ __forceinline__ float fe_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float fe_silu(float x) { return x * fe_sigmoid(x); }

__device__ __forceinline__ void unpack8(const uint4 u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = pack2_bf16(f[0], f[1]);
    u.y = pack2_bf16(f[2], f[3]);
    u.z = pack2_bf16(f[4], f[5]);
    u.w = pack2_bf16(f[6], f[7]);
    return u;
}

// acc[i][j] (+)= X[64 x H] (LDS, swizzled) * Wp (packed [H/32][H/16][64 lanes][8]) for this wave's 64 columns
template <int H>
__device__ __forceinline__ void fe_gemm(const bf16_t* sX, const bf16_t* __restrict__ Wp, int wave, int lane,
                                        f32x4_t (&acc)[4][4]) {
    constexpr int NKS = H / 32, NG = H / 16;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16x8_t* wp = (const bf16x8_t*)Wp + ((size_t)wave * 4) * 64 + lane;    // + (ks*NG + j)*64
    bf16x8_t b0[4], b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = wp[(size_t)j * 64];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b1[j] = wp[((size_t)(ks + 1) * NG + j) * 64];
        }
        bf16x8_t a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 16 + fr;
            a[i] = *(const bf16x8_t*)&sX[r * H + (((ks * 4 + fq) ^ (r & 7)) * 8)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = b1[j];
    }
}

template <int H>
__global__ __launch_bounds__(H, 4) void gcl_edge_fused_kernel(const bf16_t* __restrict__ AB, int ldab, const int* __restrict__ row,
                                                           const int* __restrict__ col, const float* __restrict__ rhohat,
                                                           const float* __restrict__ w_r, const bf16_t* __restrict__ W2p,
                                                           const float* __restrict__ b2, const bf16_t* __restrict__ Wcp,
                                                           const float* __restrict__ bc, const float* __restrict__ w3, int E,
                                                           float* agg, float* s_out) {
    constexpr int NW = H / 64;                      // waves; wave w owns output columns [64w, 64w+64)
    constexpr int CPT = FE_BM * H / H;              // gather: elements per thread = 64 (one 64-column chunk of one edge)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                     // [64][H] bf16, 16-B chunk index XOR (row & 7)
    int* sRow = (int*)(smem + (size_t)FE_BM * H * 2);
    float* sDot = (float*)(sRow + FE_BM);           // [NW][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = blockIdx.x * FE_BM;
    const int ne = min(FE_BM, E - e0);

    // ---- phase 0: gather + first Linear + SiLU -> sX
    {
        constexpr int TPE = H / 64;                 // threads per edge
        const int el = tid / TPE, c64 = tid % TPE;
        if (tid < FE_BM) sRow[tid] = (tid < ne) ? row[e0 + tid] : -1;
        if (el < ne) {
            const int r = row[e0 + el], c = col[e0 + el];
            const float rh = rhohat[e0 + el];
            const bf16_t* ap = AB + (size_t)r * ldab + c64 * 64;
            const bf16_t* bp = AB + (size_t)c * ldab + H + c64 * 64;
            const float* wp = w_r + c64 * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float fa[8], fb[8], o[8];
                unpack8(*(const uint4*)(ap + q * 8), fa);
                unpack8(*(const uint4*)(bp + q * 8), fb);
                const float4 w0 = *(const float4*)(wp + q * 8), w1 = *(const float4*)(wp + q * 8 + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = fe_silu(fa[k] + fb[k] + rh * wv[k]);
                const int ch = c64 * 8 + q;
                *(uint4*)&sX[el * H + ((ch ^ (el & 7)) * 8)] = pack8(o);
            }
        } else if (el < FE_BM) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int q = 0; q < 8; ++q) *(uint4*)&sX[el * H + (((c64 * 8 + q) ^ (el & 7)) * 8)] = z;
        }
    }
    __syncthreads();

    // ---- phase 1: M = silu(S1 W2^T + b2)
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    fe_gemm<H>(sX, W2p, wave, lane, acc);
    __syncthreads();                                // every wave has finished reading S1
    const int fr = lane & 15, cq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int colj = wave * 64 + j * 16 + fr;
        const float bj = b2[colj];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = i * 16 + cq * 4 + r;
                sX[rw * H + (((colj >> 3) ^ (rw & 7)) * 8) + (colj & 7)] = f32_to_bf16(fe_silu(acc[i][j][r] + bj));
            }
    }
    __syncthreads();

    // ---- phase 2: agg[row] += M  (run-length scan down this thread's column; rows of a node are contiguous)
    {
        const int c = tid;                          // H threads <-> H columns
        float run = 0.f;
        int cur = sRow[0];
        bool first = true;                          // a run that starts at the tile's first row may continue a neighbour tile's
        for (int rw = 0; rw < ne; ++rw) {
            const int rr = sRow[rw];
            if (rr != cur) {
                // interior runs cover ALL edges of their node (rows are node-contiguous): plain store, no read-modify-write
                if (first) atomicAdd(&agg[(size_t)cur * H + c], run); else agg[(size_t)cur * H + c] = run;
                run = 0.f; cur = rr; first = false;
            }
            run += bf16_to_f32(sX[rw * H + (((c >> 3) ^ (rw & 7)) * 8) + (c & 7)]);
        }
        if (ne > 0) atomicAdd(&agg[(size_t)cur * H + c], run);   // last run may continue in the next tile
    }

    // ---- phase 3: s = w3 . silu(M Wc^T + bc)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    fe_gemm<H>(sX, Wcp, wave, lane, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ds[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int colj = wave * 64 + j * 16 + fr;
            const float bj = bc[colj], wj = w3[colj];
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] += fe_silu(acc[i][j][r] + bj) * wj;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = ds[r];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
            if (fr == 0) sDot[wave * FE_BM + i * 16 + cq * 4 + r] = t;
        }
    }
    __syncthreads();
    if (tid < ne) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sDot[w * FE_BM + tid];
        s_out[e0 + tid] = s;
    }
}

extern "C" int fabind_gcl_edge_fused(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                                     const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                                     const float* w3, int E, float* agg, float* s_out, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_gcl_edge_fused: H must be 64, 128, 256 or 512");
    FB_REQUIRE(ldab % 8 == 0, "fabind_gcl_edge_fused: ldab % 8");
    const dim3 grid((E + FE_BM - 1) / FE_BM);
    const size_t lds = (size_t)FE_BM * H * 2 + FE_BM * sizeof(int) + (size_t)(H / 64) * FE_BM * sizeof(float);
#define FE_LAUNCH(HH)                                                                                              \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)gcl_edge_fused_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((gcl_edge_fused_kernel<HH>), grid, dim3(HH), lds, stream, (const bf16_t*)AB, ldab, row, col, rhohat, \
                           w_r, (const bf16_t*)W2p, b2, (const bf16_t*)Wcp, bc, w3, E, agg, s_out);                 \
    } while (0)
    if (H == 512) FE_LAUNCH(512); else if (H == 256) FE_LAUNCH(256); else if (H == 128) FE_LAUNCH(128); else FE_LAUNCH(64);
#undef FE_LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}
