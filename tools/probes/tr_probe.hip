// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS elements each lane receives).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(int mode, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint16_t)i;
    __syncthreads();
    int lane = threadIdx.x;
    uint32_t addr;
    if (mode == 0) addr = lane * 8;                              // consecutive 8-byte pieces
    else if (mode == 1) addr = (lane & 15) * 64 + (lane >> 4) * 8; // 16 rows of 64 B, 4 pieces per row
    else addr = (lane & 15) * 128 + (lane >> 4) * 8;
    uint32_t base = (uint32_t)(uintptr_t)s;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(v & 0xffff);
    out[lane * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
    out[lane * 4 + 2] = (uint16_t)((v >> 32) & 0xffff);
    out[lane * 4 + 3] = (uint16_t)((v >> 48) & 0xffff);
}
int main() {
    uint16_t* d; hipMalloc(&d, 512);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d (element index = byte address / 2)\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
