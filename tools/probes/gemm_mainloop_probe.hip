// Probe: main loop of the node-level NT GEMM C[M,N] = A[M,K] W[N,K]^T (bf16, fp32 accumulate) at the shapes that are main-loop-bound in
// the headline step ([98,688 x 1024] -> 512, [98,688 x 512] -> 1024), for wave layouts of the SAME 256 x 128 work-group tile:
//   <8 waves, 64 x 64 per wave>   the library's gemm_bf16_pipe_kernel<4, 32, 3, 4> layout (0.5 fragment reads per MFMA, 4 waves per SIMD)
//   <4 waves, 128 x 64 per wave>  0.375 fragment reads per MFMA, 2 waves per SIMD, still two work-groups per CU
// and BK = 32 / 64, 3- / 4-stage LDS-DMA rings.  The epilogue is the same minimal one for all (bf16 tile staged through LDS, 16-byte stores).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_mainloop_probe tools/probes/gemm_mainloop_probe.hip && /tmp/gemm_mainloop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__host__ __device__ inline bf16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

// NWV waves: NWV / 2 across M (256 / (NWV / 2) rows each), 2 across N (64 columns each)
template <int NWV, int BK_, int NSTAGE, int MINW>
__global__ __launch_bounds__(NWV * 64, MINW) void gemm_probe_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                                 int M, int N, int K) {
    constexpr int BM_ = 256, BN_ = 128, ROWS = BM_ + BN_;
    constexpr int RPW = BM_ / (NWV / 2), MI = RPW / 16;          // rows per wave, 16-row blocks per wave
    constexpr int RPP = 1024 / (BK_ * 2), PT = ROWS / RPP, PPW = PT / NWV, LPR = 64 / RPP;
    static_assert(PT % NWV == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sT = (bf16_t*)smem;
    const int nbx = N / BN_, total = gridDim.x;
    const int L = blockIdx.x, q = total / 8, r = total % 8, xcd = L % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + L / 8;
    const int bx = t % nbx, by = t / nbx;
    const int m0 = by * BM_, n0 = bx * BN_;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wave * PPW + j) * RPP + lane / LPR, cl = lane % LPR;
        const int chunk = (BK_ == 64) ? (cl ^ (row & 7)) : (cl ^ ((row >> 2) & 3));
        src[j] = row < BM_ ? A + (size_t)min(m0 + row, M - 1) * K + chunk * 8 : W + (size_t)(n0 + row - BM_) * K + chunk * 8;
    }
    auto stage = [&](int st, int k0) {
        bf16_t* base = sT + (size_t)st * ROWS * BK_;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + k0), (lptr_t)(base + (size_t)(wave * PPW + j) * RPP * BK_), 16, 0, 0);
    };
    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK_, fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s * BK_);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSTAGE - 2, nk - 1 - kt);
        if (ahead >= 2) wait_vmcnt<2 * PPW>(); else if (ahead == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < nk) stage((kt + NSTAGE - 1) % NSTAGE, (kt + NSTAGE - 1) * BK_);
        const bf16_t* tA = sT + (size_t)(kt % NSTAGE) * ROWS * BK_;
        const bf16_t* tB = tA + BM_ * BK_;
#pragma unroll
        for (int kk = 0; kk < BK_ / 32; ++kk) {
            bf16x8_t af[MI], bfr[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ra = wm * RPW + i * 16 + fr;
                const int ca = (BK_ == 64) ? ((kk * 4 + fq) ^ (ra & 7)) : (fq ^ ((ra >> 2) & 3));
                af[i] = *(const bf16x8_t*)&tA[ra * BK_ + ca * 8];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = wn * 64 + j * 16 + fr;
                const int cb = (BK_ == 64) ? ((kk * 4 + fq) ^ (rb & 7)) : (fq ^ ((rb >> 2) & 3));
                bfr[j] = *(const bf16x8_t*)&tB[rb * BK_ + cb * 8];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    // minimal epilogue: bf16 tile through LDS ([256][128 + 8]), 16-byte row-contiguous stores
    constexpr int OLD = BN_ + 8;
    bf16_t* sOut = sT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                sOut[(wm * RPW + i * 16 + (lane >> 4) * 4 + rr) * OLD + wn * 64 + j * 16 + fr] = f2b(acc[i][j][rr]);
    __syncthreads();
    for (int idx = tid; idx < BM_ * (BN_ / 8); idx += NWV * 64) {
        const int rowl = idx / (BN_ / 8), ch = idx % (BN_ / 8);
        if (m0 + rowl < M) *(uint4*)(C + (size_t)(m0 + rowl) * N + n0 + ch * 8) = *(const uint4*)&sOut[rowl * OLD + ch * 8];
    }
}

template <int NWV, int BK_, int NSTAGE, int MINW>
static float run(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int N, int K, int reps, double* checksum) {
    constexpr int ROWS = 384;
    size_t lds = (size_t)NSTAGE * ROWS * BK_ * 2;
    const size_t outb = (size_t)256 * 136 * 2;
    if (lds < outb) lds = outb;
    (void)hipFuncSetAttribute((const void*)gemm_probe_kernel<NWV, BK_, NSTAGE, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = ((M + 255) / 256) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_probe_kernel<NWV, BK_, NSTAGE, MINW>), dim3(grid), dim3(NWV * 64), lds, 0, A, W, C, M, N, K);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_probe_kernel<NWV, BK_, NSTAGE, MINW>), dim3(grid), dim3(NWV * 64), lds, 0, A, W, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<bf16_t> h(4096);
    hipMemcpy(h.data(), C + (size_t)(M - 9) * N, 4096 * 2, hipMemcpyDeviceToHost);
    double s = 0; for (bf16_t v : h) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); s += f; }
    *checksum = s;
    return ms / reps * 1e3f;
}

int main() {
    const int M = 98688;
    const int shapes[4][2] = {{512, 1024}, {1024, 512}, {512, 512}, {1536, 512}};
    bf16_t *A, *W, *C;
    hipMalloc(&A, (size_t)M * 1536 * 2); hipMalloc(&W, (size_t)1536 * 1536 * 2); hipMalloc(&C, (size_t)M * 1536 * 2);
    std::vector<bf16_t> ha((size_t)M * 1536), hw((size_t)1536 * 1536);
    uint32_t x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return f2b(((int)(x >> 9) % 2001 - 1000) * 1e-3f); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd();
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        const double gf = 2.0 * M * N * K * 1e-9;
        double c0, c;
        float t;
        printf("M=%d N=%d K=%d\n", M, N, K);
        t = run<8, 32, 3, 4>(A, W, C, M, N, K, 20, &c0); printf("  8 waves 64x64  BK32 3 stages 2 WG/CU (library layout): %7.1f us  %6.1f TFLOP/s\n", t, gf / t * 1e3);
        t = run<4, 32, 3, 2>(A, W, C, M, N, K, 20, &c);  printf("  4 waves 128x64 BK32 3 stages 2 WG/CU                  : %7.1f us  %6.1f TFLOP/s  %s\n", t, gf / t * 1e3, c == c0 ? "same" : "DIFFERENT");
        t = run<4, 32, 4, 2>(A, W, C, M, N, K, 20, &c);  printf("  4 waves 128x64 BK32 4 stages                          : %7.1f us  %6.1f TFLOP/s  %s\n", t, gf / t * 1e3, c == c0 ? "same" : "DIFFERENT");
        t = run<4, 64, 3, 2>(A, W, C, M, N, K, 20, &c);  printf("  4 waves 128x64 BK64 3 stages (1 WG/CU: 144 KiB)       : %7.1f us  %6.1f TFLOP/s  %s\n", t, gf / t * 1e3, c == c0 ? "same" : "DIFFERENT");
        t = run<8, 64, 3, 2>(A, W, C, M, N, K, 20, &c);  printf("  8 waves 64x64  BK64 3 stages (1 WG/CU)                : %7.1f us  %6.1f TFLOP/s  %s\n", t, gf / t * 1e3, c == c0 ? "same" : "DIFFERENT");
        t = run<8, 32, 4, 4>(A, W, C, M, N, K, 20, &c);  printf("  8 waves 64x64  BK32 4 stages (1 WG/CU: 96 KiB)        : %7.1f us  %6.1f TFLOP/s  %s\n", t, gf / t * 1e3, c == c0 ? "same" : "DIFFERENT");
    }
    return 0;
}
