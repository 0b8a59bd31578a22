// Probe: what bounds the CSR segment-sum?  Variants of "read E rows of 512 bf16, reduce runs of DEG rows".
// hipcc --offload-arch=gfx950 -O3 -o /tmp/segsum_probe segsum_probe.hip && /tmp/segsum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct F8 { float v[8]; };
__device__ __forceinline__ F8 ld8(const uint16_t* p) {
    const uint4 u = *(const uint4*)p; F8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
// V0: fixed DEG rows per wave, WPB waves per block, unroll U
template <int WPB, int U>
__global__ __launch_bounds__(WPB * 64) void v0(const uint16_t* Z, float* out, int n_rows, int deg) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    F8 acc; for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
    const uint16_t* p = Z + (size_t)r * deg * 512 + lane * 8;
    int e = 0;
    for (; e + U <= deg; e += U) {
        F8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld8(p + (size_t)(e + u) * 512);
#pragma unroll
        for (int u = 0; u < U; ++u) for (int k = 0; k < 8; ++k) acc.v[k] += v[u].v[k];
    }
    for (; e < deg; ++e) { F8 v = ld8(p + (size_t)e * 512); for (int k = 0; k < 8; ++k) acc.v[k] += v.v[k]; }
    float* o = out + (size_t)r * 512 + lane * 8;
    *(float4*)o = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
    *(float4*)(o + 4) = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
}
// V1: same with rowptr indirection
template <int WPB, int U>
__global__ __launch_bounds__(WPB * 64) void v1(const uint16_t* Z, const int* rp, float* out, int n_rows) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int e0 = rp[r], e1 = rp[r + 1];
    F8 acc; for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
    const uint16_t* p = Z + lane * 8;
    int e = e0;
    for (; e + U <= e1; e += U) {
        F8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld8(p + (size_t)(e + u) * 512);
#pragma unroll
        for (int u = 0; u < U; ++u) for (int k = 0; k < 8; ++k) acc.v[k] += v[u].v[k];
    }
    for (; e < e1; ++e) { F8 v = ld8(p + (size_t)e * 512); for (int k = 0; k < 8; ++k) acc.v[k] += v.v[k]; }
    float* o = out + (size_t)r * 512 + lane * 8;
    *(float4*)o = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
    *(float4*)(o + 4) = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
}
// V2: grid-stride persistent waves over rows (fewer, longer-lived waves)
template <int U>
__global__ __launch_bounds__(256) void v2(const uint16_t* Z, const int* rp, float* out, int n_rows) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += nw) {
        const int e0 = rp[r], e1 = rp[r + 1];
        F8 acc; for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
        const uint16_t* p = Z + lane * 8;
        int e = e0;
        for (; e + U <= e1; e += U) {
            F8 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld8(p + (size_t)(e + u) * 512);
#pragma unroll
            for (int u = 0; u < U; ++u) for (int k = 0; k < 8; ++k) acc.v[k] += v[u].v[k];
        }
        for (; e < e1; ++e) { F8 v = ld8(p + (size_t)e * 512); for (int k = 0; k < 8; ++k) acc.v[k] += v.v[k]; }
        float* o = out + (size_t)r * 512 + lane * 8;
        *(float4*)o = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
        *(float4*)(o + 4) = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
    }
}
// copy baseline
__global__ void cp(const uint4* a, uint4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i];
}
// read-only baseline: each thread sums 16 uint4 strided by grid
__global__ void rd(const uint4* a, float* o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x;
    unsigned s = 0;
    for (; i < n; i += st) { uint4 v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) o[0] = 1.f;
}
template <class F> float timeit(F f, int it = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a, 0); for (int i = 0; i < it; ++i) f(); hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / it;
}
int main() {
    const int N = 98688, DEG = 16; const size_t E = (size_t)N * DEG;
    uint16_t* Z; float* out; int* rp; uint4* Zc;
    hipMalloc(&Z, E * 512 * 2); hipMalloc(&Zc, E * 512 * 2); hipMalloc(&out, (size_t)N * 512 * 4); hipMalloc(&rp, (N + 1) * 4);
    hipMemset(Z, 0, E * 512 * 2);
    std::vector<int> h(N + 1); for (int i = 0; i <= N; ++i) h[i] = i * DEG;
    hipMemcpy(rp, h.data(), (N + 1) * 4, hipMemcpyHostToDevice);
    const double by = (double)E * 1024 + (double)N * 2048;
    auto rep = [&](const char* nm, float ms) { printf("%-28s %.3f ms  %.2f TB/s\n", nm, ms, by / ms / 1e9); };
    size_t n16 = E * 64;
    float ms = timeit([&] { cp<<<(n16 + 255) / 256, 256>>>((const uint4*)Z, Zc, n16); });
    printf("%-28s %.3f ms  %.2f TB/s (read+write)\n", "copy", ms, 2.0 * E * 1024 / ms / 1e9);
    ms = timeit([&] { rd<<<256 * 32, 256>>>((const uint4*)Z, out, n16); });
    printf("%-28s %.3f ms  %.2f TB/s (read only)\n", "read grid-stride", ms, 1.0 * E * 1024 / ms / 1e9);
    rep("v0 fixed deg WPB4 U4", timeit([&] { v0<4, 4><<<(N + 3) / 4, 256>>>(Z, out, N, DEG); }));
    rep("v0 fixed deg WPB4 U8", timeit([&] { v0<4, 8><<<(N + 3) / 4, 256>>>(Z, out, N, DEG); }));
    rep("v0 fixed deg WPB4 U16", timeit([&] { v0<4, 16><<<(N + 3) / 4, 256>>>(Z, out, N, DEG); }));
    rep("v0 fixed deg WPB1 U8", timeit([&] { v0<1, 8><<<N, 64>>>(Z, out, N, DEG); }));
    rep("v0 fixed deg WPB8 U8", timeit([&] { v0<8, 8><<<(N + 7) / 8, 512>>>(Z, out, N, DEG); }));
    rep("v1 rowptr WPB4 U4", timeit([&] { v1<4, 4><<<(N + 3) / 4, 256>>>(Z, rp, out, N); }));
    rep("v1 rowptr WPB4 U8", timeit([&] { v1<4, 8><<<(N + 3) / 4, 256>>>(Z, rp, out, N); }));
    rep("v2 persistent 2048blk U4", timeit([&] { v2<4><<<2048, 256>>>(Z, rp, out, N); }));
    rep("v2 persistent 2048blk U8", timeit([&] { v2<8><<<2048, 256>>>(Z, rp, out, N); }));
    rep("v2 persistent 1024blk U8", timeit([&] { v2<8><<<1024, 256>>>(Z, rp, out, N); }));
    rep("v2 persistent 4096blk U8", timeit([&] { v2<8><<<4096, 256>>>(Z, rp, out, N); }));
    return 0;
}
