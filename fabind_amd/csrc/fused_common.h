// Device helpers shared by the fused tile kernels (fused_edge.hip, pair_fused.hip): swizzled [rows][H] bf16 LDS
// tiles, packed-weight MFMA contraction, counter-based dropout.
#pragma once
#include "common.h"

#define FE_BM 64
// wave priority around the MFMA clusters of the tile contractions: -2 % per launch on the fused edge backward (same-box A/B)
#define FE_PRIO(x) __builtin_amdgcn_s_setprio(x)

__device__ __forceinline__ float fe_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float fe_silu(float x) { return x * fe_sigmoid(x); }

// Counter-based dropout mask of the messages (train mode, egnn.py:82): one 32-bit hash per (edge pair, column)
// gives two 16-bit uniforms, element (e, c) is kept iff its half is >= thr16 = round(p * 65536).  Forward and
// backward kernels (and tests/helpers.py) evaluate the same function, nothing is stored.
__device__ __forceinline__ uint32_t fe_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float fe_keep(uint32_t seed, uint32_t e, uint32_t c, uint32_t H, uint32_t thr16, float scale) {
    const uint32_t h = fe_hash(seed + (e >> 1) * H + c);
    return (((e & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16) ? scale : 0.f;
}

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

// acc[i][j] (+)= X[BM x H] (LDS, swizzled) * Wp (packed [H/32][H/16][64 lanes][8]) for this wave's 64 columns.
// The k-loop is kept rolled (two k-steps per trip, B fragments ping-pong in registers): a fully unrolled loop lets
// the scheduler hoist every B-fragment load of the contraction at once and spill.
// SWAP: the operand-swapped form C^T = W x X^T -- the same fragments in the other operand slots; the accumulator quad of a lane
// is then (row i*16 + (lane & 15), columns j*16 + (lane >> 4)*4 .. +3) instead of (rows i*16 + (lane >> 4)*4 .. +3, column j*16 + (lane & 15)).
// SWZ: row bits XOR-ed into the 16-byte chunk index of the LDS tile.  With 7 (round 1) rows r and r + 8 of a 16-row fragment sit in
// the same 16-byte slot modulo 256 B, and the 16-lane groups of ds_read_b128 see a 2-way bank conflict on every A-fragment read
// (47 % of the LDS cycles of the round-1 kernels were conflict cycles); 15 spreads a fragment's 16 rows over all 16 slots.
// wmode (profiling builds only; 0 everywhere else and folded away): 1 = every k-step re-reads the wave's first fragments (L1 hits),
// 2 = no weight loads at all -- how much of a contraction is the weight stream?
template <int H, int MI, bool SWAP = false, int SWZ = 7>
__device__ __forceinline__ void fe_gemm_rolled_pf(const bf16_t* sX, const bf16_t* __restrict__ Wp, int wave, int lane,
                                                  f32x4_t (&acc)[MI][4], bf16x8_t (&b0)[4], const int wmode = 0) {
    constexpr int NKS = H / 32;
    const int NG = wmode == 1 ? 0 : H / 16;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16x8_t* wp = (const bf16x8_t*)Wp + ((size_t)wave * 4) * 64 + lane;
    bf16x8_t b1[4];
    if (wmode == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = b0[j];
    }
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 2) {
        if (wmode != 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b1[j] = wp[((size_t)(ks + 1) * NG + j) * 64];
        }
        bf16x8_t a[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = i * 16 + fr;
            a[i] = *(const bf16x8_t*)&sX[r * H + (((ks * 4 + fq) ^ (r & SWZ)) * 8)];
        }
        FE_PRIO(1);      // co-resident work-groups sit in VALU epilogues: the matrix-core stream goes first
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b0[j], acc[i][j], 0, 0, 0);
        FE_PRIO(0);
        // Unconditional (the last trip re-reads a fragment it does not use): a load under `if (ks + 2 < NKS)` gives the second half of
        // the trip two predecessors, and the compiler then waits there for the smaller in-flight count -- vmcnt(3) instead of
        // vmcnt(7), i.e. for the fragments just requested for the NEXT trip.  That wait was the contraction's main stall.
        if (wmode != 2) {
            const int kn = min(ks + 2, NKS - 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = wp[((size_t)kn * NG + j) * 64];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = i * 16 + fr;
            a[i] = *(const bf16x8_t*)&sX[r * H + ((((ks + 1) * 4 + fq) ^ (r & SWZ)) * 8)];
        }
        FE_PRIO(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b1[j], acc[i][j], 0, 0, 0);
        FE_PRIO(0);
    }
}

// The same contraction with ROTATING activation fragments (round 2, after the weight-gradient kernel's lesson that a wave's LDS latency
// and its MFMAs add up unless the reads are issued a phase ahead): a[i] feeds the four MFMAs of row granule i and is then free, so the
// read of the NEXT half-trip's a[i] is issued right behind them -- every fragment read has twelve MFMAs (192 cycles) to land under and
// no register is added.  sched_group_barrier pins the order (4 MFMA, 1 DS read) x MI; without it the scheduler hoists the reads to the
// top of the half-trip, which is the old form.  Same MFMA order as fe_gemm_rolled_pf: results are bit-identical.
// MEASURED in the store-wave backward (FE3_ROT=1): 1.3-3 % SLOWER per launch than the plain form -- at two to three waves per SIMD the
// co-resident waves already cover the fragment reads, and the per-granule lgkmcnt waits cost more than they hide.  Kept as a knob.
template <int H, int MI, bool SWAP = false, int SWZ = 7>
__device__ __forceinline__ void fe_gemm_rot_pf(const bf16_t* sX, const bf16_t* __restrict__ Wp, int wave, int lane,
                                               f32x4_t (&acc)[MI][4], bf16x8_t (&b0)[4]) {
    constexpr int NKS = H / 32;
    constexpr int NG = H / 16;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16x8_t* wp = (const bf16x8_t*)Wp + ((size_t)wave * 4) * 64 + lane;
    bf16x8_t b1[4], a[MI];
    auto frag = [&](int ks, int i) {
        const int r = i * 16 + fr;
        return *(const bf16x8_t*)&sX[r * H + (((ks * 4 + fq) ^ (r & SWZ)) * 8)];
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = frag(0, i);
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = wp[((size_t)(ks + 1) * NG + j) * 64];
        FE_PRIO(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b0[j], acc[i][j], 0, 0, 0);
            a[i] = frag(ks + 1, i);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);        // the four weight loads first
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
        }
        FE_PRIO(0);
        const int kn = min(ks + 2, NKS - 2);                      // unconditional: see fe_gemm_rolled_pf
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = wp[((size_t)kn * NG + j) * 64];
        FE_PRIO(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b1[j], acc[i][j], 0, 0, 0);
            a[i] = frag(kn, i);                                   // the last trip re-reads a fragment it does not use
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        FE_PRIO(0);
    }
}

// the first k-step's weight fragments of a contraction.  (Requesting them ahead of the barrier in front of the contraction, during the
// preceding epilogue, was tried in the store-wave backward: 5.45 -> 5.82 ms per launch -- sixteen more live registers in the epilogues.)
__device__ __forceinline__ void fe_gemm_prefetch(const bf16_t* __restrict__ Wp, int wave, int lane, bf16x8_t (&b0)[4]) {
    const bf16x8_t* wp = (const bf16x8_t*)Wp + ((size_t)wave * 4) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = wp[(size_t)j * 64];
}
template <int H, int MI, bool SWAP = false, int SWZ = 7>
__device__ __forceinline__ void fe_gemm_rolled(const bf16_t* sX, const bf16_t* __restrict__ Wp, int wave, int lane,
                                               f32x4_t (&acc)[MI][4], const int wmode = 0) {
    bf16x8_t b0[4];
    fe_gemm_prefetch(Wp, wave, lane, b0);
    fe_gemm_rolled_pf<H, MI, SWAP, SWZ>(sX, Wp, wave, lane, acc, b0, wmode);
}
template <int H, int MI, bool SWAP = false, int SWZ = 7>
__device__ __forceinline__ void fe_gemm_rot(const bf16_t* sX, const bf16_t* __restrict__ Wp, int wave, int lane, f32x4_t (&acc)[MI][4]) {
    bf16x8_t b0[4];
    fe_gemm_prefetch(Wp, wave, lane, b0);
    fe_gemm_rot_pf<H, MI, SWAP, SWZ>(sX, Wp, wave, lane, acc, b0);
}

// Split-bf16 ("bf16x3") form of the operand-swapped tile contraction: the activation tile exists as two swizzled bf16 LDS planes
// (hi = bf16(x), lo = bf16(x - hi)), the packed weight as two fragment arrays (hi | lo); a product term is three MFMAs
// lo*hi + hi*lo + hi*hi with fp32 accumulation -- fp32-grade results at 3/16 of the exact-fp32 matrix time (the
// reference contracts in fp32, egnn.py:68-144).  Same rolled two-k-steps-per-trip structure and unconditional next-trip loads as
// fe_gemm_rolled_pf; 96 MFMAs per trip and wave at MI = 4.
template <int H, int MI, int SWZ>
__device__ __forceinline__ void fe_gemm_x3(const bf16_t* sXh, const bf16_t* sXl, const bf16_t* __restrict__ Wph,
                                           const bf16_t* __restrict__ Wpl, int wave, int lane, f32x4_t (&acc)[MI][4]) {
    constexpr int NKS = H / 32;
    constexpr int NG = H / 16;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16x8_t* wh = (const bf16x8_t*)Wph + ((size_t)wave * 4) * 64 + lane;
    const bf16x8_t* wl = (const bf16x8_t*)Wpl + ((size_t)wave * 4) * 64 + lane;
    bf16x8_t h0[4], l0[4], h1[4], l1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h0[j] = wh[(size_t)j * 64]; l0[j] = wl[(size_t)j * 64]; }
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { h1[j] = wh[((size_t)(ks + 1) * NG + j) * 64]; l1[j] = wl[((size_t)(ks + 1) * NG + j) * 64]; }
        bf16x8_t ah[MI], al[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = i * 16 + fr;
            const int o = r * H + (((ks * 4 + fq) ^ (r & SWZ)) * 8);
            ah[i] = *(const bf16x8_t*)&sXh[o];
            al[i] = *(const bf16x8_t*)&sXl[o];
        }
        FE_PRIO(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0[j], ah[i], acc[i][j], 0, 0, 0);
        }
        FE_PRIO(0);
        {
            const int kn = min(ks + 2, NKS - 2);                  // unconditional: see fe_gemm_rolled_pf
#pragma unroll
            for (int j = 0; j < 4; ++j) { h0[j] = wh[((size_t)kn * NG + j) * 64]; l0[j] = wl[((size_t)kn * NG + j) * 64]; }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = i * 16 + fr;
            const int o = r * H + ((((ks + 1) * 4 + fq) ^ (r & SWZ)) * 8);
            ah[i] = *(const bf16x8_t*)&sXh[o];
            al[i] = *(const bf16x8_t*)&sXl[o];
        }
        FE_PRIO(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1[j], ah[i], acc[i][j], 0, 0, 0);
        }
        FE_PRIO(0);
    }
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi), for a pair of values: -> (packed hi pair, packed lo pair)
__device__ __forceinline__ void fe_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = pack2_bf16(x0, x1);
    lo = pack2_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

template <int H>
__device__ __forceinline__ void fe_tile_store(const bf16_t* sB, bf16_t* __restrict__ g, int e0, int ne, int tid) {
    constexpr int CH = H / 8;
    bf16_t* gt = g + (size_t)e0 * H;                      // uniform tile base (SGPRs) + 32-bit lane offsets below
    for (unsigned q = tid; q < (unsigned)ne * CH; q += H) {
        const unsigned rw = q / CH, ch = q % CH;
        *(uint4*)&gt[rw * H + ch * 8] = *(const uint4*)&sB[rw * H + ((ch ^ (rw & 7)) * 8)];
    }
}

// 16-byte store that does NOT keep its line in the XCD's L2 (sc1: write-through and drop).  The [E,H] operand tiles of the
// backward are read again only by later launches; stored plainly they allocate 4 MiB per XCD and tile phase -- all of L2 --
// and push the weight fragments and gathered rows every CU keeps re-reading out of it.
typedef unsigned int fe_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fe_store16_stream(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rsrc, (int)byte_off, 0, /*aux: sc1*/ 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fe_rsrc(void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
// fe_tile_store through streaming stores
template <int H>
__device__ __forceinline__ void fe_tile_store_stream(const bf16_t* sB, bf16_t* __restrict__ g, int e0, int ne, int tid) {
    constexpr int CH = H / 8;
    const __amdgpu_buffer_rsrc_t rs = fe_rsrc(g + (size_t)e0 * H, (unsigned)ne * H * 2);
    for (unsigned q = tid; q < (unsigned)ne * CH; q += H) {
        const unsigned rw = q / CH, ch = q % CH;
        fe_store16_stream(rs, (rw * H + ch * 8) * 2, *(const uint4*)&sB[rw * H + ((ch ^ (rw & 7)) * 8)]);
    }
}

template <int MI>
__device__ __forceinline__ void fe_zero(f32x4_t (&acc)[MI][4]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}


// ---------------------------------------------------------------------------------------------------------------------
// Deterministic segment sum of a [rows][H] LDS tile over row-sorted edges (one column per thread).
// A node's edges are contiguous, so inside a 64-edge tile every run of equal row ids is a complete node EXCEPT possibly the
// tile's first run (it may continue a run of the previous tile: head_cont) and its last run (it may continue into the next
// tile: tail_cont).  Complete runs are plain stores.  The two boundary runs go to bnd[tile][0 | 1][H] and a fix-up kernel
// (fe_boundary_fix_kernel) adds the pieces of every node that spans tiles IN TILE ORDER -- no float atomics, so the result
// does not depend on which work-group finishes first (round 1 used atomicAdd for the boundary runs: a node with 1,500 edges
// spans 24 tiles, and two identical forward passes differed by an ulp about once in ten runs, amplified to 1e-4 in h by
// bf16 rounding downstream).  Returns sum_rows rh[row] * value (the d w_r partial of the backward) when WITH_RH.
template <int H, bool WITH_RH, int BM_, int SWZ = 7>
__device__ __forceinline__ float fe_scan_rows(const bf16_t* sX, const int* sRow, const float* sRh, int ne, bool head_cont,
                                              bool tail_cont, float* out, unsigned ld, float* bnd_tile, int c,
                                              bf16_t* out16 = nullptr, unsigned ld16 = 0) {
    // complete runs go to out (fp32, if given) and / or out16 (bf16, if given)
    auto put = [&](int node, float v) {
        if (out) out[(unsigned)node * ld + c] = v;
        if (out16) out16[(unsigned)node * ld16 + c] = f32_to_bf16(v);
    };
    float run = 0.f, pwr = 0.f;
    int cur = sRow[0];
    bool first = true;
    for (int rw0 = 0; rw0 < ne; rw0 += 8) {
        int rr[8];
        float v[8], rhv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rw = min(rw0 + u, BM_ - 1);         // the tile height BM_ is a multiple of 8
            rr[u] = sRow[rw];
            rhv[u] = WITH_RH ? sRh[rw] : 0.f;
            v[u] = bf16_to_f32(sX[rw * H + (((c >> 3) ^ (rw & SWZ)) * 8) + (c & 7)]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rw0 + u < ne) {
                if (rr[u] != cur) {
                    if (first && head_cont) bnd_tile[c] = run; else put(cur, run);
                    run = 0.f; cur = rr[u]; first = false;
                }
                run += v[u];
                if (WITH_RH) pwr += rhv[u] * v[u];
            }
        }
    }
    if (ne > 0) {
        if (first && head_cont) bnd_tile[c] = run;            // the whole tile continues the previous tile's node
        else if (tail_cont) bnd_tile[H + c] = run;
        else put(cur, run);
    }
    return pwr;
}

// The same scan for 64-row tiles, latency paid once: all 64 values of the thread's column are read from LDS into registers back to
// back (the scan runs when the accumulators are dead), the run boundaries come from ONE ballot over the row ids (wave-uniform, so the
// per-row control flow is scalar), and the sums are taken from registers in row order -- same order, bit-identical results.
// Used by the forward kernel (fwd +0.2 ... +1.4 % in four same-box pairs); in the store-wave backward it measured -0.5 % (its 64 extra
// live registers at the 168 budget), which keeps fe_scan_rows.
template <int H, bool WITH_RH, int SWZ>
__device__ __forceinline__ float fe_scan_runs64(const bf16_t* sX, const int* sRow, const float* sRh, int ne, bool head_cont,
                                                bool tail_cont, float* out, unsigned ld, float* bnd_tile, int c,
                                                bf16_t* out16 = nullptr, unsigned ld16 = 0, const bf16_t* sXlo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int mine = sRow[lane], prev = sRow[max(lane - 1, 0)];
    const float rhl = WITH_RH ? sRh[lane] : 0.f;
    const unsigned long long m = __ballot(lane < ne && lane > 0 && mine != prev);    // bit r: row r starts a NEW run (r > 0)
    const int c8 = c >> 3, c7 = c & 7;
    float v[64];
#pragma unroll
    for (int rw = 0; rw < 64; ++rw) v[rw] = bf16_to_f32(sX[rw * H + ((c8 ^ (rw & SWZ)) * 8) + c7]);
    if (sXlo) {                                                     // split-bf16 tile: the value is hi + lo
#pragma unroll
        for (int rw = 0; rw < 64; ++rw) v[rw] += bf16_to_f32(sXlo[rw * H + ((c8 ^ (rw & SWZ)) * 8) + c7]);
    }
    float run = 0.f, pwr = 0.f;
    int rs = 0;                                                     // first row of the current run (wave-uniform)
    auto flush = [&](bool last) {
        if (rs == 0 && head_cont) bnd_tile[c] = run;                 // continues the previous tile's node (possibly through to the next)
        else if (last && tail_cont) bnd_tile[H + c] = run;           // continues into the next tile
        else {
            const unsigned node = (unsigned)__builtin_amdgcn_readlane(mine, rs);
            if (out) out[node * ld + c] = run;
            if (out16) out16[node * ld16 + c] = f32_to_bf16(run);
        }
    };
#pragma unroll
    for (int rw = 0; rw < 64; ++rw) {
        if (rw < ne) {
            if ((m >> rw) & 1ull) {
                flush(false);
                run = 0.f;
                rs = rw;
            }
            run += v[rw];
            if (WITH_RH) pwr += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rhl), rw)) * v[rw];
        }
    }
    if (ne > 0) flush(true);
    return pwr;
}

// One work-group (H threads) per tile: the tile where a node's tile-spanning run STARTS adds up its pieces in tile order.
template <int BM>
__global__ void fe_boundary_fix_kernel(const int* __restrict__ row, int E, int H, const float* __restrict__ bnd, float* out,
                                       unsigned ld, bf16_t* out16 = nullptr, unsigned ld16 = 0) {
    const int t = blockIdx.x, e0 = t * BM, ne = min(BM, E - e0);
    if (ne <= 0 || e0 + ne >= E) return;
    const int r = row[e0 + ne - 1];
    if (row[e0 + ne] != r) return;                                         // the last run ends with the tile
    if (row[e0] == r && e0 > 0 && row[e0 - 1] == r) return;                // a through tile: the run started earlier
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float sum = bnd[((size_t)t * 2 + 1) * H + c];
        for (int u = t + 1;; ++u) {
            const int f0 = u * BM, fn = min(BM, E - f0);
            sum += bnd[((size_t)u * 2) * H + c];
            const bool through = row[f0 + fn - 1] == r && f0 + fn < E && row[f0 + fn] == r;
            if (!through) break;
        }
        if (out) out[(unsigned)r * ld + c] = sum;
        if (out16) out16[(unsigned)r * ld16 + c] = f32_to_bf16(sum);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Element-wise helpers of the row-wise / operand-swapped kernels (fused_edge_bwd3.hip, fused_edge_bwd4.hip, the saving forward)
// sum over the 16 lanes of a DPP row (lanes 16q .. 16q+15); every lane of the row receives the total
__device__ __forceinline__ float fe3_row16_sum(float v) {
#define FE_DPP_ADD(ctrl_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), (ctrl_), 0xF, 0xF, true))
    FE_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    FE_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    FE_DPP_ADD(0x141);       // row_half_mirror
    FE_DPP_ADD(0x140);       // row_mirror
#undef FE_DPP_ADD
    return v;
}

// Pairs of fp32 values in adjacent registers: the arithmetic below compiles to the packed v_pk_{add,mul,fma}_f32 forms, two
// elements per VALU instruction; the two transcendentals per element (v_exp_f32, v_rcp_f32: quarter rate) stay scalar.
typedef float fe_f2 __attribute__((ext_vector_type(2)));
// m = silu(z) = z s(z), d = silu'(z) = s + m (1 - s)
__device__ __forceinline__ void fe3_silu_pair(const fe_f2 z, fe_f2& m, fe_f2& d) {
    const fe_f2 t = z * -1.44269504f;
    const fe_f2 o = fe_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    const fe_f2 s = fe_f2{__builtin_amdgcn_rcpf(o.x), __builtin_amdgcn_rcpf(o.y)};
    m = z * s;
    d = s + (m - m * s);
}
__device__ __forceinline__ uint32_t fe3_pack(const fe_f2 v) { return pack2_bf16(v.x, v.y); }
__device__ __forceinline__ fe_f2 fe3_unpack(const uint32_t u) { return fe_f2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

// 16-byte buffer accesses: (uniform base in a descriptor) + (32-bit lane offset) + (uniform offset in an SGPR / literal): no 64-bit
// per-lane pointers, which the compiler hoists out of the tile loop by the dozen and spills
__device__ __forceinline__ uint4 fe3_bload16(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned soff) {
    const fe_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fe3_bstore16(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned soff, const uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rs, (int)voff, (int)soff, 0);
}

// fused_edge_bwd2.hip: the row-wise / operand-swapped backward (variants 3 and 4 of fabind_gcl_edge_fused_bwd)
struct FabindEdgeBwdArgs;
// fused_edge_bwd3.hip: the same with a store wave (variants 5 and 6)
int fe_bwd3_launch(const FabindEdgeBwdArgs* a, int H, int bm, int n_groups, hipStream_t stream);
// fused_edge_fwd3.hip: the split-bf16 forward (fabind_gcl_edge_fused_x3: fp32 AB rows, hi | lo packed weights)
int fe_fwd3_launch(const float* AB, int ldab, int H, const int* row, const int* col, const float* rhohat, const float* w_r,
                   const void* W2ph, const void* W2pl, const float* b2, const void* Wcph, const void* Wcpl, const float* bc,
                   const float* w3, int E, float* agg, float* s_out, uint32_t thr16, float dscale, unsigned seed, float* bnd,
                   int xcd_aware, hipStream_t stream, void* Msave = nullptr, void* d2f = nullptr, void* z3f = nullptr);
// fused_edge_fwd2.hip: the row-wise / operand-swapped forward (variant 1 of fabind_gcl_edge_fused)
int fe_fwd2_launch(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat, const float* w_r,
                   const void* W2p, const float* b2, const void* Wcp, const float* bc, const float* w3, int E, float* agg,
                   float* s_out, uint32_t thr16, float dscale, unsigned seed, float* bnd, void* agg16, int xcd_aware, hipStream_t stream,
                   void* Msave = nullptr, void* d2f = nullptr, void* z3f = nullptr);
// fused_edge_bwd4.hip: the backward over what the saving forward left (FabindEdgeBwdArgs.d2f / z3f / Mm): two contractions per edge
int fe_bwd4_launch(const FabindEdgeBwdArgs* a, int H, int n_groups, hipStream_t stream);
