// Backward of the fused intra-graph edge pipeline (see fused_edge.hip for the algebra), fourth form = the row-wise / operand-swapped
// in-place tile of fused_edge_bwd2.hip plus a STORE WAVE.
//
// Why: the five [E,H] bf16 operand tiles a work-group writes per tile of edges (S1, M, dT, dP2 for the two weight-gradient
// contractions, dP1 for the sending-side sum: 7.9 GB per launch at the bench shape) cost the other forms 1.1-1.4 ms of a 6.3-6.5 ms
// launch -- the whole HBM write time, un-overlapped (tools/probes/edge_bwd2_sensitivity.py: 6.5 ms -> 5.2 ms with the stores
// skipped, 5.7 ms with them redirected into an L2-resident slab).  A wave's vector-memory operations complete in order, so every
// weight-fragment / gathered-row load issued after a 128 KiB store burst waits until the memory system has taken the burst; neither
// moving the burst behind the contraction, nor sc1 (no L2 allocation) stores, nor a start skew between work-groups changed that.
// tools/probes/store_rate.hip shows the way out: the same bursts issued by a wave of their own do not delay the loads of the
// other waves of the work-group at all (14.29 vs 14.27 us per tile, against +1.8 us per burst when the loading waves store).
//
// So the work-group is H/64 compute waves + ONE store wave (H + 64 threads).  Phases are separated by work-group barriers; while the
// compute waves run a contraction that only READS the LDS tile, the store wave copies that tile to its operand in global memory:
//
//     compute waves                                         store wave
//     P0  gather, S1 -> tile                                (idle)
//     --- barrier
//     contraction 1  (reads S1)                             S1 tile -> global, silu'(pre1) tile -> scratch slab
//     --- barrier
//     epilogue 1     (M -> tile, silu'(pre2) -> 2nd tile)   (idle)
//     --- barrier
//     contraction 2                                         M tile -> global
//     ...            (dT with contraction 3, dP2 with contraction 4)
//     P5  dP1 -> tile
//     --- barrier
//     P6  row scan   (reads dP1)                            dP1 tile -> global
//     --- barrier
//
// Nine waves put three on one SIMD, i.e. <= 168 VGPRs per wave: the tile is 64 edges (accumulators 64 + fragments 48 registers; a
// 96-edge instantiation spilled in the epilogues, each reload a vector-memory round trip behind the epilogue's own stores).  Two LDS
// tiles: the edge tile, rewritten in place by every stage, and a second one that holds silu'(pre1) from P0 until the store wave has
// copied it to the work-group's scratch slab (read back in P5) and then silu'(pre2) from the P1 epilogue to the P3 epilogue -- the
// compute waves issue no global stores except the per-node sums of the row scan and d rhohat.
// The gathered rows of P0 and the scratch rows of P5 are loaded a batch of iterations at a time (all loads of a batch in flight
// before the first is used); the row / col / rhohat / ds tables of the NEXT tile are fetched during P6.  (Issuing the next tile's
// gathers during P6 as well only moved their latency into the scan, whose own loads return behind them: tried, +5 %; issuing them at the
// start of P5 with the tables two tiles ahead keeps 64 registers live across the loop's back edge, which the allocator spills: 5.4 -> 7.7 ms;
// letting the STORE WAVE fetch the next tile's sender rows by LDS-DMA into the second LDS tile while contraction 4 runs took P0 from
// 20 k to 13 k cycles per tile and gave the 7 k back in contraction 4 and at the barriers: 5.40 -> 5.56 ms, same process.)
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"
#ifndef FE3_ROT
#define FE3_ROT 0     // 1 = contractions with rotating activation fragments (fe_gemm_rot): bit-identical, 5.37 / 5.46 ms per launch against
                      // 5.30 / 5.29 for the plain form on one box (DESIGN.md iteration-log row 40) -- not the default
#endif

// one wave copies the first ne rows of a swizzled [.][H] LDS tile to g[e0 .. e0+ne)[H]
template <int H, int SWZ>
__device__ __forceinline__ void fe3_wave_store(const bf16_t* sB, bf16_t* __restrict__ g, int e0, int ne, int lane) {
    constexpr int CH = H / 8;
    bf16_t* gt = g + (size_t)e0 * H;
    const unsigned total = (unsigned)ne * CH;
#pragma unroll 8
    for (unsigned q = lane; q < total; q += 64) {
        const unsigned rw = q / CH, ch = q % CH;
        *(uint4*)&gt[rw * H + ch * 8] = *(const uint4*)&sB[rw * H + ((ch ^ (rw & SWZ)) * 8)];
    }
}

// the same copy with NON-TEMPORAL stores (nt) for the five [E,H] operand tiles, which nothing in this launch reads again.  Round-3
// experiment (profiles/r03_edge_bwd_nt.txt): the question was whether the operand streams evict the work-group's silu'(pre1) scratch slab
// (64 KiB, rewritten and re-read every tile) from L2 and so cause its 3.2 GB of HBM round trip per launch.  They do not -- FETCH_SIZE /
// WRITE_SIZE are unchanged (5.7 GB / 9.7 GB per launch) -- but the launch is 1.2-2.5 % shorter in the step (5,327 / 5,356 / 5,306 us ->
// 5,196 / 5,229 / 5,241 us in three interleaved pairs), results bit-identical: the default since then (knob bit 32 of
// fabind_gcl_edge_fused_bwd3_set_exp restores the plain stores).
template <int H, int SWZ>
__device__ __forceinline__ void fe3_wave_store_nt(const bf16_t* sB, bf16_t* __restrict__ g, int e0, int ne, int lane) {
    constexpr int CH = H / 8;
    const __amdgpu_buffer_rsrc_t rs = fe_rsrc(g + (size_t)e0 * H, (unsigned)ne * H * 2);
    const unsigned total = (unsigned)ne * CH;
#pragma unroll 8
    for (unsigned q = lane; q < total; q += 64) {
        const unsigned rw = q / CH, ch = q % CH;
        const uint4 v = *(const uint4*)&sB[rw * H + ((ch ^ (rw & SWZ)) * 8)];
        __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rs, (int)((rw * H + ch * 8) * 2), 0, /*aux: nt*/ 2);
    }
}

template <int H, int BM, bool DROP, bool DBG>
__global__ __launch_bounds__(H + 64) void gcl_edge_fused_bwd3_kernel(const FabindEdgeBwdArgs p, const int xf) {
    constexpr int MI = BM / 16;                                   // 16-edge MFMA blocks per wave (the wave owns 64 features of ALL edges)
    constexpr int NW = H / 64;                                    // compute waves; wave NW is the store wave
    constexpr int CH = H / 8;                                     // 16-byte chunks per edge row
    constexpr int EPW = 64 / CH;                                  // edges per wave and iteration of the row-wise phases (1 at H = 512)
    constexpr int EPI = NW * EPW;                                 // = 8 edges per work-group iteration
    constexpr int NIT = BM / EPI;
    constexpr int PB = (NIT % 8 == 0) ? 8 : (NIT % 6 == 0) ? 6 : 4;   // iterations per load batch of the row-wise phases
    static_assert(NIT % PB == 0 && BM % 16 == 0, "tile height");
    constexpr int SWZ = (H >= 128) ? 15 : 7;                       // LDS tile swizzle: chunk ^ (row & SWZ) (fused_common.h fe_gemm_rolled)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [BM][H] swizzled: S1 -> M -> dT -> dP2 -> dS1 -> dP1
    bf16_t* sD = sX + BM * H;                                     // [BM][H] swizzled: silu'(pre1) from P0 until the store wave has copied it to
                                                                  // this work-group's scratch slab (during contraction 1), then silu'(pre2) (P1 -> P3)
    int* sTab = (int*)(sD + BM * H);                             // [2][4][BM]: {row, col, ds, rhohat} of this tile and the next
    float* sPart = (float*)(sTab + 8 * BM);                       // [4][H]: column sums of {d b2, d bc, d w3, d w_r} over this work-group's tiles
    const bf16_t* __restrict__ AB = (const bf16_t*)p.AB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (p.E + BM - 1) / BM;
    int t_first, t_end, t_step;
    if (p.xcd_aware && (gridDim.x & 7) == 0) {
        const int tpx = (n_tiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * tpx + (int)(blockIdx.x >> 3);
        t_end = min(n_tiles, (xcd + 1) * tpx);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t_first = blockIdx.x; t_end = n_tiles; t_step = gridDim.x;
    }

    if (wave == NW) {
        // ================= store wave: four operand tiles under the contractions, the fifth under the row scan =================
        __syncthreads();                                          // (tables of the first tile)
        for (int tile = t_first; tile < t_end; tile += t_step) {
            const int e0 = tile * BM;
            const int ne = min(BM, p.E - e0);
            const bool st = !(xf & 1), nt = (xf & 32) == 0;     // non-temporal operand stores unless knob bit 32 is set
            auto put = [&](bf16_t* dst) {
                if (nt) fe3_wave_store_nt<H, SWZ>(sX, dst, e0, ne, lane); else fe3_wave_store<H, SWZ>(sX, dst, e0, ne, lane);
            };
            __syncthreads();                                      // S1 ready
            if (st) put((bf16_t*)p.S1);
            fe3_wave_store<H, SWZ>(sD, (bf16_t*)p.d2scratch + (size_t)blockIdx.x * (BM * H), 0, BM, lane);   // silu'(pre1) -> scratch slab
            __syncthreads();                                      // contraction 1 done
            __syncthreads();                                      // M ready
            if (st) put((bf16_t*)p.Mm);
            __syncthreads();                                      // contraction 2 done
            __syncthreads();                                      // dT ready
            if (st) put((bf16_t*)p.dT);
            __syncthreads();                                      // contraction 3 done
            __syncthreads();                                      // dP2 ready
            if (st) put((bf16_t*)p.dP2);
            __syncthreads();                                      // contraction 4 done
            __syncthreads();                                      // dS1 ready
            __syncthreads();                                      // dP1 ready
            if (st) put((bf16_t*)p.dP1);
            __syncthreads();                                      // row scan done: the tile may be overwritten
        }
        __syncthreads();                                          // (partial sums)
        return;
    }

    // ================= compute waves =================
    const int fr = lane & 15, cq = lane >> 4;
    const int sub = lane / CH, ch = lane % CH;                    // row-wise phases: this lane's edge within the wave's group, its chunk
    const uint32_t thr16 = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);
    for (int k = 0; k < 4; ++k) sPart[k * H + tid] = 0.f;
    const unsigned uld = (unsigned)p.ldab;
    // scratch slabs of this work-group: silu'(pre2) in fragment order, silu'(pre1) in tile order
    const __amdgpu_buffer_rsrc_t rs_d1 = fe_rsrc((bf16_t*)p.d2scratch + (size_t)blockIdx.x * (BM * H), BM * H * 2);
    const unsigned rw_off = (unsigned)((wave * EPW + sub) * CH + ch) * 16;     // byte offset of this lane's chunk in a row-wise iteration
    const __amdgpu_buffer_rsrc_t rs_ab = fe_rsrc(const_cast<void*>(p.AB), 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rs_dagg = fe_rsrc(const_cast<float*>(p.dagg), 0xffffffffu);
    // LDS element offset of this lane's accumulator quad (i, j): edge i*16 + fr, features wave*64 + j*16 + cq*4 .. +3
    //   chunk = (wave*8 + j*2 + (cq>>1)) ^ (fr & SWZ) = cx ^ (j*2)
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
    const int wmode = DBG ? ((xf >> 2) & 3) : 0;                  // profiling build: weight-stream experiment of fe_gemm_rolled
#define FE_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    const int fcol = wave * 64 + cq * 4;                          // first feature of the quad of block j: fcol + j*16
    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = 0;
#define FE_TICK(k_)                                                   \
    if constexpr (DBG) {                                              \
        const long long now_ = __builtin_readcyclecounter();         \
        tph[k_] += now_ - tlast;                                      \
        tlast = now_;                                                 \
    }
    // tables {row, col, ds, rhohat} of a tile (rows past the end: row -1, the rest 0): fetched into registers by the first BM threads,
    // committed to half `hb` of sTab later
    int tb_r = -1, tb_c = 0;
    float tb_d = 0.f, tb_h = 0.f;
    auto fetch_tables = [&](int tile) {
        const int e0 = tile * BM;
        const int ne = min(BM, p.E - e0);
        if (tid < BM) {
            const bool ok = tid < ne;
            const unsigned ge = (unsigned)(e0 + (ok ? tid : 0));
            const int r = p.row[ge], c = p.col[ge];
            const float d = p.ds[ge], h = p.rhohat[ge];
            tb_r = ok ? r : -1; tb_c = ok ? c : 0; tb_d = ok ? d : 0.f; tb_h = ok ? h : 0.f;
        }
    };
    auto commit_tables = [&](int hb) {
        int* tb = sTab + hb * 4 * BM;
        if (tid < BM) {
            tb[tid] = tb_r;
            tb[BM + tid] = tb_c;
            ((float*)tb)[2 * BM + tid] = tb_d;
            ((float*)tb)[3 * BM + tid] = tb_h;
        }
    };
    // gathered AB rows of a tile, one batch of row-wise iterations per call
    constexpr int NB = NIT / PB;
    uint4 ga[NB][PB], gb[NB][PB];
    auto gather_batch = [&](const int* tab, int b) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int e = (b * PB + u) * EPI + wave * EPW + sub;
            const unsigned r = (unsigned)max(tab[e], 0), c = (unsigned)tab[BM + e];
            ga[b][u] = fe3_bload16(rs_ab, (r * uld + ch * 8) * 2, 0);
            gb[b][u] = fe3_bload16(rs_ab, (c * uld + H + ch * 8) * 2, 0);
        }
    };
    static_assert(BM <= H, "one thread per table row");
    if (t_first < t_end) { fetch_tables(t_first); commit_tables(0); }
    __syncthreads();
    int hb = 0;

    for (int tile = t_first; tile < t_end; tile += t_step, hb ^= 1) {
        const int e0 = tile * BM;
        const int ne = min(BM, p.E - e0);
        const int* sRow = sTab + hb * 4 * BM;
        const int* sCol = sRow + BM;
        const float* sDs = (const float*)(sRow + 2 * BM);
        const float* sRh = (const float*)(sRow + 3 * BM);
        int lz = 0;
        asm volatile("" : "+s"(lz));                              // loop-invariant operands are re-read per tile, not hoisted
        const float* w_r = p.w_r + lz;
        const float* b2 = p.b2 + lz;
        const float* bc = p.bc + lz;
        const float* w3 = p.w3 + lz;
        const bf16_t* W2p = (const bf16_t*)p.W2p + lz;
        const bf16_t* Wcp = (const bf16_t*)p.Wcp + lz;
        const bf16_t* W2Tp = (const bf16_t*)p.W2Tp + lz;
        const bf16_t* WcTp = (const bf16_t*)p.WcTp + lz;
        if constexpr (DBG) tlast = __builtin_readcyclecounter();

        // ---- P0 (row-wise): S1 = silu(A[row] + B[col] + rhohat w_r) -> tile; silu'(pre1) -> scratch
        {
            fe_f2 wv2[4];
            {
                const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
                wv2[0] = fe_f2{w0.x, w0.y}; wv2[1] = fe_f2{w0.z, w0.w}; wv2[2] = fe_f2{w1.x, w1.y}; wv2[3] = fe_f2{w1.z, w1.w};
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) gather_batch(sRow, b);   // every gathered row of the tile in flight before the first is used
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int e = (b * PB + u) * EPI + wave * EPW + sub;
                    const float rh = sRh[e];
                    const uint32_t ua[4] = {ga[b][u].x, ga[b][u].y, ga[b][u].z, ga[b][u].w}, ub[4] = {gb[b][u].x, gb[b][u].y, gb[b][u].z, gb[b][u].w};
                    uint32_t po[4], pd[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const fe_f2 z = fe3_unpack(ua[k]) + fe3_unpack(ub[k]) + wv2[k] * rh;
                        fe_f2 m, d;
                        fe3_silu_pair(z, m, d);
                        po[k] = fe3_pack(m);
                        pd[k] = fe3_pack(d);
                    }
                    const bool ok = e < ne;
                    *(uint4*)&sX[e * H + ((ch ^ (e & SWZ)) * 8)] = ok ? make_uint4(po[0], po[1], po[2], po[3]) : make_uint4(0u, 0u, 0u, 0u);
                    *(uint4*)&sD[e * H + ((ch ^ (e & SWZ)) * 8)] = make_uint4(pd[0], pd[1], pd[2], pd[3]);
                }
            }
        }
        __syncthreads();                                          // S1 ready
        FE_TICK(0)

        // ---- P1: pre2 = S1 W2^T + b2;  M -> tile in place;  silu'(pre2) -> scratch
        f32x4_t acc[MI][4];
        fe_zero(acc);
        if (DBG || FE3_ROT == 0) fe_gemm_rolled<H, MI, true, SWZ>(sX, W2p, wave, lane, acc, wmode);
        else fe_gemm_rot<H, MI, true, SWZ>(sX, W2p, wave, lane, acc);
        __syncthreads();                                          // every wave has finished reading S1 (and the store wave has copied it)
        FE_TICK(2)
        {
            float4 bq4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bq4[j] = *(const float4*)(b2 + fcol + j * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const fe_f2 b01 = fe_f2{bq4[j].x, bq4[j].y}, b23 = fe_f2{bq4[j].z, bq4[j].w};
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    fe_f2 m01, d01, m23, d23;
                    fe3_silu_pair(fe_f2{acc[i][j][0], acc[i][j][1]} + b01, m01, d01);
                    fe3_silu_pair(fe_f2{acc[i][j][2], acc[i][j][3]} + b23, m23, d23);
                    if constexpr (DROP) {
                        const uint32_t ee = (uint32_t)(e0 + i * 16 + fr), cc = (uint32_t)(fcol + j * 16);
                        const fe_f2 k01 = fe_f2{fe_keep(p.seed, ee, cc, H, thr16, dscale), fe_keep(p.seed, ee, cc + 1, H, thr16, dscale)};
                        const fe_f2 k23 = fe_f2{fe_keep(p.seed, ee, cc + 2, H, thr16, dscale), fe_keep(p.seed, ee, cc + 3, H, thr16, dscale)};
                        m01 *= k01; d01 *= k01; m23 *= k23; d23 *= k23;
                    }
                    *(uint2*)&sD[FE_QOFF(i, j)] = make_uint2(fe3_pack(d01), fe3_pack(d23));
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(fe3_pack(m01), fe3_pack(m23));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                          // M ready
        FE_TICK(3)

        // ---- P2: pre3 = M Wc^T + bc;  dT = ds * w3 * silu'(pre3) -> tile in place
        fe_zero(acc);
        if (DBG || FE3_ROT == 0) fe_gemm_rolled<H, MI, true, SWZ>(sX, Wcp, wave, lane, acc, wmode);
        else fe_gemm_rot<H, MI, true, SWZ>(sX, Wcp, wave, lane, acc);
        __syncthreads();
        FE_TICK(2)
        {
            float dsr[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) dsr[i] = sDs[i * 16 + fr];
            float4 bq4[4], wq4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { bq4[j] = *(const float4*)(bc + fcol + j * 16); wq4[j] = *(const float4*)(w3 + fcol + j * 16); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const fe_f2 b01 = fe_f2{bq4[j].x, bq4[j].y}, b23 = fe_f2{bq4[j].z, bq4[j].w};
                const fe_f2 w01 = fe_f2{wq4[j].x, wq4[j].y}, w23 = fe_f2{wq4[j].z, wq4[j].w};
                fe_f2 a3a = fe_f2{0.f, 0.f}, a3b = a3a, aca = a3a, acb = a3a;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    fe_f2 m01, d01, m23, d23;
                    fe3_silu_pair(fe_f2{acc[i][j][0], acc[i][j][1]} + b01, m01, d01);
                    fe3_silu_pair(fe_f2{acc[i][j][2], acc[i][j][3]} + b23, m23, d23);
                    const fe_f2 t01 = (w01 * dsr[i]) * d01, t23 = (w23 * dsr[i]) * d23;
                    a3a += m01 * dsr[i]; a3b += m23 * dsr[i];
                    aca += t01; acb += t23;
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(fe3_pack(t01), fe3_pack(t23));
                }
                float a3[4] = {a3a.x, a3a.y, a3b.x, a3b.y}, ac[4] = {aca.x, aca.y, acb.x, acb.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) { a3[r] = fe3_row16_sum(a3[r]); ac[r] = fe3_row16_sum(ac[r]); }
                if (fr < 4) {                                     // lane fr = r owns feature fcol + j*16 + r of sPart (one writer per slot)
                    const float va = fr == 0 ? a3[0] : fr == 1 ? a3[1] : fr == 2 ? a3[2] : a3[3];
                    const float vc = fr == 0 ? ac[0] : fr == 1 ? ac[1] : fr == 2 ? ac[2] : ac[3];
                    sPart[2 * H + fcol + j * 16 + fr] += va;
                    sPart[1 * H + fcol + j * 16 + fr] += vc;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                          // dT ready
        FE_TICK(4)

        // ---- P3: dM = dT Wc + dagg[row];  dP2 = dM * silu'(pre2) -> tile in place   (dagg and silu'(pre2) are fetched one 16-feature
        //      block ahead inside the epilogue: loading dagg into the accumulators first put its latency in front of the contraction)
        fe_zero(acc);
        FE_TICK(5)
        if (DBG || FE3_ROT == 0) fe_gemm_rolled<H, MI, true, SWZ>(sX, WcTp, wave, lane, acc, wmode);
        else fe_gemm_rot<H, MI, true, SWZ>(sX, WcTp, wave, lane, acc);
        __syncthreads();
        FE_TICK(2)
        {
            float4 dg[2][MI];                                     // dagg[row of edge i*16 + fr][fcol + j*16 ..]
            unsigned dgo[MI];
            float dgk[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rr = sRow[i * 16 + fr];
                dgk[i] = rr >= 0 ? 1.f : 0.f;
                dgo[i] = ((unsigned)max(rr, 0) * (unsigned)p.lddagg + fcol) * 4;
            }
            auto fetch_block = [&](int j, int slot) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const uint4 v = fe3_bload16(rs_dagg, dgo[i], (unsigned)(j * 64));
                    dg[slot][i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            };
            fetch_block(0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 3) fetch_block(j + 1, (j + 1) & 1);
                fe_f2 a2a = fe_f2{0.f, 0.f}, a2b = a2a;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float4 g4 = dg[j & 1][i];
                    const uint2 dq = *(const uint2*)&sD[FE_QOFF(i, j)];
                    const fe_f2 p01 = (fe_f2{acc[i][j][0], acc[i][j][1]} + fe_f2{g4.x, g4.y} * dgk[i]) * fe3_unpack(dq.x);
                    const fe_f2 p23 = (fe_f2{acc[i][j][2], acc[i][j][3]} + fe_f2{g4.z, g4.w} * dgk[i]) * fe3_unpack(dq.y);
                    a2a += p01; a2b += p23;
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(fe3_pack(p01), fe3_pack(p23));
                }
                float a2[4] = {a2a.x, a2a.y, a2b.x, a2b.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[r] = fe3_row16_sum(a2[r]);
                if (fr < 4) sPart[fcol + j * 16 + fr] += fr == 0 ? a2[0] : fr == 1 ? a2[1] : fr == 2 ? a2[2] : a2[3];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                          // dP2 ready
        FE_TICK(6)

        // ---- P4: dS1 = dP2 W2 -> tile in place
        fe_zero(acc);
        if (DBG || FE3_ROT == 0) fe_gemm_rolled<H, MI, true, SWZ>(sX, W2Tp, wave, lane, acc, wmode);
        else fe_gemm_rot<H, MI, true, SWZ>(sX, W2Tp, wave, lane, acc);
        __syncthreads();
        FE_TICK(2)
        uint4 gd[NIT];                                            // silu'(pre1) of this lane's row-wise chunks, back from the scratch slab for P5
#pragma unroll
        for (int it = 0; it < NIT; ++it) gd[it] = fe3_bload16(rs_d1, rw_off, (unsigned)(it * EPI * CH * 16));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(acc[i][j][0], acc[i][j][1]), pack2_bf16(acc[i][j][2], acc[i][j][3]));
        __syncthreads();                                          // dS1 ready
        FE_TICK(7)

        // ---- P5 (row-wise): dP1 = dS1 * silu'(pre1) -> tile in place;  d rhohat = dP1 . w_r   (the tables of the next tile are
        //      fetched underneath it)
        const bool has_next = tile + t_step < t_end;
        {
            fe_f2 wv2[4];
            {
                const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
                wv2[0] = fe_f2{w0.x, w0.y}; wv2[1] = fe_f2{w0.z, w0.w}; wv2[2] = fe_f2{w1.x, w1.y}; wv2[3] = fe_f2{w1.z, w1.w};
            }
#pragma unroll
            for (int it0 = 0; it0 < NIT; it0 += PB) {
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int e = (it0 + u) * EPI + wave * EPW + sub;
                    uint4* sp = (uint4*)&sX[e * H + ((ch ^ (e & SWZ)) * 8)];
                    const uint4 gs = *sp;
                    const uint4 gq = gd[it0 + u];
                    const uint32_t ug[4] = {gs.x, gs.y, gs.z, gs.w}, ud[4] = {gq.x, gq.y, gq.z, gq.w};
                    uint32_t pg[4];
                    fe_f2 dot2 = fe_f2{0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const fe_f2 g = fe3_unpack(ug[k]) * fe3_unpack(ud[k]);
                        dot2 += g * wv2[k];
                        pg[k] = fe3_pack(g);
                    }
                    float dot = dot2.x + dot2.y;
                    *sp = make_uint4(pg[0], pg[1], pg[2], pg[3]);
                    if constexpr (CH >= 16) {
                        dot = fe3_row16_sum(dot);
#pragma unroll
                        for (int o = 16; o < CH; o <<= 1) dot += __shfl_xor(dot, o, 64);
                    } else {
#pragma unroll
                        for (int o = 1; o < CH; o <<= 1) dot += __shfl_xor(dot, o, 64);
                    }
                    if (ch == 0 && e < ne) (p.drh + e0)[(unsigned)e] = dot;
                }
            }
        }
        __syncthreads();                                          // dP1 ready
        FE_TICK(8)

        // ---- P6 (one column per thread): receiving-side segment sum of dP1 (deterministic, fe_scan_rows) and the d w_r column sum;
        //      the tables of the next tile are fetched underneath it
        if (has_next) fetch_tables(tile + t_step);
        if (!(xf & 16)) {
            const bool head_cont = e0 > 0 && p.row[e0 - 1] == sRow[0];
            const bool tail_cont = e0 + ne < p.E && p.row[e0 + ne] == sRow[ne - 1];
            sPart[3 * H + tid] += fe_scan_rows<H, true, BM, SWZ>(sX, sRow, sRh, ne, head_cont, tail_cont, p.dAB16 ? nullptr : p.dABrow, (unsigned)p.lddab,
                                                         p.bnd + (size_t)tile * 2 * H, tid, (bf16_t*)p.dAB16, (unsigned)p.lddab16);
        }
        if (has_next) commit_tables(hb ^ 1);
        __syncthreads();                                          // the next tile overwrites the tile; its tables are in place
        FE_TICK(9)
    }
    if constexpr (DBG) {
        if (p.dbg && blockIdx.x == 0 && tid == 0)
            for (int k = 0; k < 12; ++k) ((long long*)p.dbg)[k] = tph[k];
    }
    // per-work-group partial column sums: part[blockIdx.x][{b2, bc, w3, w_r}][H]   (each sPart slot has ONE writer lane)
    __syncthreads();
    float* part = p.part + (size_t)blockIdx.x * 4 * H;
    for (int k = 0; k < 4; ++k) part[k * H + tid] = sPart[k * H + tid];
#undef FE_QOFF
#undef FE_TICK
}

// development knob: bit mask of work the kernel SKIPS (wrong results; sensitivity timing only): 1 operand stores, 16 no row scan
static int g_fe_bwd3_exp = 0;
extern "C" void fabind_gcl_edge_fused_bwd3_set_exp(int m) { g_fe_bwd3_exp = m; }      // (bit 32 gives correct results: plain instead of non-temporal operand stores)

// launched from fabind_gcl_edge_fused_bwd (fused_edge.hip), variant 5
int fe_bwd3_launch(const FabindEdgeBwdArgs* a, int H, int bm, int n_groups, hipStream_t stream) {
    FB_REQUIRE(a->d2scratch != nullptr, "fabind_gcl_edge_fused_bwd: the store-wave form needs d2scratch (n_groups x 64 x H x 2 bytes)");
    FB_REQUIRE(a->bnd != nullptr, "fabind_gcl_edge_fused_bwd: bnd (ceil(E / tile) x 2 x H floats of scratch) is required");
    FB_REQUIRE(a->lddagg % 4 == 0 && ((uintptr_t)a->dagg & 15) == 0, "fabind_gcl_edge_fused_bwd: dagg must be 16-byte aligned with lddagg % 4 == 0");
    const size_t lds = (size_t)2 * bm * H * 2 + (size_t)8 * bm * sizeof(int) + (size_t)4 * H * sizeof(float);
#define FE_LAUNCH3_(HH, BB, KK, DD)                                                                                   \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) {                                                                                               \
            const hipError_t e_ = hipFuncSetAttribute((const void*)gcl_edge_fused_bwd3_kernel<HH, BB, KK, DD>,         \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e_ != hipSuccess) { fabind_set_error(hipGetErrorString(e_)); return (int)e_; }                     \
            set_ = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((gcl_edge_fused_bwd3_kernel<HH, BB, KK, DD>), dim3(n_groups), dim3(HH + 64), lds, stream, *a, g_fe_bwd3_exp); \
    } while (0)
#define FE_LAUNCH3(HH, BB, DD) do { if (a->p_drop > 0.f) FE_LAUNCH3_(HH, BB, true, DD); else FE_LAUNCH3_(HH, BB, false, DD); } while (0)
    FB_REQUIRE(bm == 64, "fabind_gcl_edge_fused_bwd: the store-wave form has 64-edge tiles");
    if (a->dbg != nullptr && H == 512) FE_LAUNCH3(512, 64, true);
    else if (H == 512) FE_LAUNCH3(512, 64, false); else if (H == 256) FE_LAUNCH3(256, 64, false);
    else if (H == 128) FE_LAUNCH3(128, 64, false); else FE_LAUNCH3(64, 64, false);
    hipLaunchKernelGGL((fe_boundary_fix_kernel<64>), dim3((a->E + 63) / 64), dim3(H < 256 ? H : 256), 0, stream, a->row, a->E, H,
                       a->bnd, a->dAB16 ? nullptr : a->dABrow, (unsigned)a->lddab, (bf16_t*)a->dAB16, (unsigned)a->lddab16);
#undef FE_LAUNCH3
#undef FE_LAUNCH3_
    FB_CHECK_LAUNCH();
    return 0;
}
