// Probe: what a burst of tile stores costs a CU on MI355X.  One work-group of 512 threads per CU walks `tiles` tiles; per tile it
// writes 128 KiB (16 x 16-byte coalesced stores per thread) to fresh memory, optionally spins `work` cycles (stand-in for the
// contraction + epilogue between two tile stores of the fused edge backward), optionally issues ONE dependent 16-byte load right
// after the burst and waits for it (a wave's loads return behind its earlier stores).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/probes/store_rate.hip && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void burst_kernel(uint4* out, const uint4* in, int tiles, int work, int dep_load, int do_store,
                                                     unsigned long long* sink) {
    const int tid = threadIdx.x;
    uint4 v = make_uint4(tid, blockIdx.x, 3u, 4u);
    unsigned acc = 0;
    for (int t = 0; t < tiles; ++t) {
        uint4* dst = out + ((size_t)t * gridDim.x + blockIdx.x) * (128 * 1024 / 16);
        if (do_store) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dst[q * 512 + tid] = v;
        }
        if (dep_load) {
            const uint4 x = in[(size_t)(t & 63) * 512 + tid];          // L2-resident 512 KiB table
            acc += x.x;
            v.z += x.y & 1u;                                              // the next burst depends on it
        }
        if (work) {
            const long long until = __builtin_readcyclecounter() + work;
            while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
        }
        v.x += 1;
    }
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

// Role split: waves 0-7 do `loads` dependent 16-byte L2-hit loads spread over the work time, wave 8 writes the 128 KiB tile.
// Do another wave's stores (and the back-pressure of a chip-wide write burst) delay THIS wave's loads?
__global__ __launch_bounds__(576) void split_kernel(uint4* out, const uint4* in, int tiles, int work, int loads, int do_store,
                                                     unsigned long long* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint4 v = make_uint4(tid, blockIdx.x, 3u, 4u);
    unsigned acc = 0;
    for (int t = 0; t < tiles; ++t) {
        if (wave == 8) {
            if (do_store) {
                uint4* dst = out + ((size_t)t * gridDim.x + blockIdx.x) * (128 * 1024 / 16);
#pragma unroll 16
                for (int q = 0; q < 128; ++q) dst[q * 64 + lane] = v;
            }
        } else {
            unsigned idx = tid;
            for (int l = 0; l < loads; ++l) {
                const uint4 x = in[(size_t)((t + l) & 63) * 512 + (idx & 511)];
                acc += x.x;
                idx += x.y;                                               // dependent chain (table holds 0x01010101)
                const long long until = __builtin_readcyclecounter() + work / loads;
                while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        v.x += 1;
    }
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

int main() {
    const int tiles = 200;
    size_t bytes = (size_t)tiles * 256 * 128 * 1024;
    uint4 *out, *in;
    unsigned long long* sink;
    hipMalloc(&out, bytes); hipMalloc(&in, 64 * 512 * 16); hipMalloc(&sink, 8);
    hipMemset(in, 1, 64 * 512 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int grid, int work, int dep, int st) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(burst_kernel, dim3(grid), dim3(512), 0, 0, out, in, tiles, work, dep, st, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double us_tile = best * 1e3 / tiles;
        printf("grid %3d work %6d dep_load %d store %d : %8.3f ms  %7.2f us/tile  (%.1f B/clk/CU at 2.4 GHz, %.2f TB/s chip)\n", grid, work, dep,
               st, best, us_tile, st ? 131072.0 / (us_tile * 2400.0) : 0.0, st ? grid * 131072.0 / (us_tile * 1e6) : 0.0);
    };
    for (int grid : {8, 32, 64, 128, 256}) run(grid, 0, 0, 1);
    for (int work : {10000, 30000, 60000}) {
        run(256, work, 0, 0);
        run(256, work, 0, 1);
        run(256, work, 1, 0);
        run(256, work, 1, 1);
    }
    auto run2 = [&](int grid, int work, int loads, int st) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(576), 0, 0, out, in, tiles, work, loads, st, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("split: grid %3d work %6d loads/tile %3d store-wave %d : %8.3f ms  %7.2f us/tile\n", grid, work, loads, st, best, best * 1e3 / tiles);
    };
    for (int loads : {1, 8, 32}) {
        run2(256, 30000, loads, 0);
        run2(256, 30000, loads, 1);
    }
    run2(256, 12000, 8, 0);
    run2(256, 12000, 8, 1);
    run(32, 30000, 1, 1);
    run(32, 30000, 0, 1);
    return 0;
}
