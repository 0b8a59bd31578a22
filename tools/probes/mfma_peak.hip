// Probe: issue-bound rate of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 on MI355X with 1, 2 and 4 waves per SIMD
// (16 independent accumulators per wave, operands constant): what "100 % matrix-core busy" is in TFLOP/s on this box and clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probes/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}
__global__ void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (s == 123.456f) out[0] = s;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        for (int which = 0; which < 2; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (which == 0) hipLaunchKernelGGL(k16, dim3(256), dim3(256 * wps), 0, 0, out, iters);
                else hipLaunchKernelGGL(k32, dim3(256), dim3(256 * wps), 0, 0, out, iters);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double n_mfma = (double)256 * 4 * wps * iters * (which == 0 ? 16 : 4);
            const double flop = n_mfma * (which == 0 ? 16384.0 : 32768.0);
            printf("%s  %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", which == 0 ? "16x16x32" : "32x32x16", wps,
                   best, flop / best / 1e9, best * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
        }
    }
    return 0;
}
