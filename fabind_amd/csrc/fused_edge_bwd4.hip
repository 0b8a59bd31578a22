// Backward of the fused intra-graph edge pipeline (algebra and reference lines: fused_edge.hip) over what the SAVING forward left
// (fused_edge_fwd2.hip, SAVE): a structurally different adjoint -- no forward recompute.
//
// Why (round 4; the store-wave recompute form of fused_edge_bwd3.hip had reached 0.245 of the bf16 MFMA peak against a self-measured
// ceiling of 0.29 for that structure): of its 124-128 k cycles per 64-edge tile, the two RECOMPUTE contractions (pre2, pre3) and their
// SiLU epilogues are ~45 k, and the second LDS tile that parks silu'(pre1) / silu'(pre2) pins the kernel at one work-group per CU, so
// nothing overlaps a work-group's element-wise phases, gathers and barrier tails.  Here
//     * silu'(pre2) (x dropout keep) and pre3 come back from the forward as bf16 in ACCUMULATOR-FRAGMENT order (8-byte quad per lane,
//       512 contiguous bytes per load instruction: the lane <-> (edge, feature) map is the forward's), the messages M are already in
//       global memory as the weight gradient's operand -- two H x H contractions per edge instead of four;
//     * silu'(pre1) and S1 are rebuilt where they are used (the row-wise phase P5 re-gathers A[row] + B[col]: L2 hits, element-wise
//       work only), so there is ONE LDS tile, <= 128 VGPRs, and TWO work-groups per CU: one's contraction runs under the other's
//       element-wise / gather / store phases -- the overlap the forward kernel has (44 % matrix-core busy against this path's 30 %);
//     * no store wave: the operand tiles (dT, dP2 row-major for the weight gradients) are copied out of the LDS tile by all waves right
//       before the contraction that reads the same tile; S1 and dP1 leave straight from the registers of the row-wise phase.
//
//     P0  pre3 quads -> dT = ds w3 silu'(pre3) -> tile; column sums d bc, d w3            (element-wise, fragment order)
//     C3  dT tile -> global;  dM = dT Wc
//     E3  dP2 = (dM + dagg[row]) silu'(pre2) -> tile; column sum d b2                       (silu'(pre2) quads from the forward)
//     C4  dP2 tile -> global;  dS1 = dP2 W2
//     E4  dS1 -> tile
//     P5  (row-wise) gather, S1 = silu(pre1) -> global, dP1 = dS1 silu'(pre1) -> tile + global, d rhohat
//     P6  receiving-side segment sum of dP1 (deterministic scan + boundary fix-up), d w_r
//
// Per edge: 2 x 2 H^2 flops (the useful dgrad work), HBM 2 x 2H bytes read (d2f, z3f) + 4 x 2H written (dT, dP2, S1, dP1).  Numerics:
// silu'(pre2) is the same bf16 value the recompute form kept in LDS; pre3 is rounded to bf16 before silu / silu' (the recompute form
// evaluates them on the fp32 accumulator): a relative 2^-9 on pre3, below the bf16 rounding of dT itself.
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

#ifndef FE4_E3_DB
#define FE4_E3_DB 0        // 1 = dagg / silu'(pre2) of the next 16-feature block requested under the current block's arithmetic (48 more live registers)
#endif
#ifndef FE4_SCAN64
#define FE4_SCAN64 0       // 1 = the all-rows-in-registers scan of the forward kernel (fe_scan_runs64) instead of fe_scan_rows
#endif
#ifndef FE4_P5_BATCH
#define FE4_P5_BATCH 4     // gathered rows in flight per lane and batch of the row-wise phase (8 = the whole tile: 64 registers)
#endif

#ifndef FE4_NT_LOADS
#define FE4_NT_LOADS 1     // 1 = the forward's fragment-ordered tiles (read once, 2 x 64 KiB per tile) are loaded non-temporally: they stream past the L2
                           //     lines that hold the AB rows P5 gathers a second time (profiles/r04_pmc.json: 1.31 x the algorithmic bytes)
#endif
#if FE4_NT_LOADS
typedef unsigned fe_u32x2_nt_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 fe4_ldq(const bf16_t* p_) { const fe_u32x2_nt_t v = __builtin_nontemporal_load((const fe_u32x2_nt_t*)p_); return make_uint2(v.x, v.y); }
#else
__device__ __forceinline__ uint2 fe4_ldq(const bf16_t* p_) { return *(const uint2*)p_; }
#endif

template <int H>
__global__ __launch_bounds__(H, 4) void gcl_edge_fused_bwd4_kernel(const FabindEdgeBwdArgs p, const int xf) {
    constexpr int BM = FE_BM, MI = BM / 16;
    constexpr int NW = H / 64;
    constexpr int CH = H / 8;                                     // 16-byte chunks per edge row
    constexpr int EPW = 64 / CH;                                  // edges per wave and iteration of the row-wise phase (1 at H = 512)
    constexpr int EPI = NW * EPW;                                 // = 8
    constexpr int NIT = BM / EPI;                                 // = 8
    constexpr int SWZ = (H >= 128) ? 15 : 7;
    static_assert(BM <= H, "one thread per table row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [BM][H] swizzled: dT -> dP2 -> dS1 -> dP1
    int* sTab = (int*)(sX + BM * H);                             // [2][4][BM]: {row, col, ds, rhohat} of this tile and the next
    float* sPart = (float*)(sTab + 8 * BM);                       // [4][H]: column sums of {d b2, d bc, d w3, d w_r} over this work-group's tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int sub = lane / CH, ch = lane % CH;
    const int n_tiles = (p.E + BM - 1) / BM;
    int t_first, t_end, t_step;
    if (p.xcd_aware && (gridDim.x & 7) == 0) {                    // XCD x walks the x-th eighth of the tiles: whole complexes per L2
        const int tpx = (n_tiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * tpx + (int)(blockIdx.x >> 3);
        t_end = min(n_tiles, (xcd + 1) * tpx);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t_first = blockIdx.x; t_end = n_tiles; t_step = gridDim.x;
    }
    for (int k = 0; k < 4; ++k) sPart[k * H + tid] = 0.f;
    const unsigned uld = (unsigned)p.ldab;
    const __amdgpu_buffer_rsrc_t rs_ab = fe_rsrc(const_cast<void*>(p.AB), 0xffffffffu);
    const __amdgpu_buffer_rsrc_t rs_dagg = fe_rsrc(const_cast<float*>(p.dagg), 0xffffffffu);
    const bf16_t* __restrict__ d2f = (const bf16_t*)p.d2f;
    const bf16_t* __restrict__ z3f = (const bf16_t*)p.z3f;
    // LDS element offset of this lane's accumulator quad (i, j): edge i*16 + fr, features wave*64 + j*16 + cq*4 .. +3
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & SWZ);
    const int lbase = fr * H + (cq & 1) * 4;
    // (the per-lane bases are passed through an empty asm at the head of every phase: the compiler otherwise hoists the sixteen quad
    //  offsets, the copy-out offsets and the row-wise offsets out of the tile loop -- ~70 values -- and spills them to scratch)
#define FE_OPAQUE(v_) asm volatile("" : "+v"(v_))
#define FE_QOFF(i, j) (lb_ + (i) * 16 * H + ((cx_ ^ ((j) * 2)) * 8))
#define FE_FOFF(i, j) (fbase + (size_t)((i) * 4 + (j)) * 256)
    const int fcol = wave * 64 + cq * 4;                          // first feature of the quad of block j: fcol + j*16

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
        const float* w3 = p.w3 + lz;
        const bf16_t* W2Tp = (const bf16_t*)p.W2Tp + lz;
        const bf16_t* WcTp = (const bf16_t*)p.WcTp + lz;
        int ln_ = lane;
        FE_OPAQUE(ln_);
        const size_t fbase = (((size_t)tile * NW + wave) * (MI * 4) * 64 + ln_) * 4;
        // the [ne][H] LDS tile -> rows e0 .. e0+ne of a row-major operand (rows past ne fall outside the descriptor: dropped)
        auto copy_out = [&](void* dst) {
            const __amdgpu_buffer_rsrc_t rs = fe_rsrc((bf16_t*)dst + (size_t)e0 * H, (unsigned)ne * H * 2);
            int t_ = tid;
            FE_OPAQUE(t_);
#pragma unroll
            for (int it = 0; it < BM * CH / H; ++it) {
                const int q = t_ + it * H;
                const int rw = q / CH, c8 = q % CH;
                const uint4 v = *(const uint4*)&sX[rw * H + ((c8 ^ (rw & SWZ)) * 8)];
                __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{v.x, v.y, v.z, v.w}, rs, (int)((rw * H + c8 * 8) * 2), 0, /*aux: nt*/ 2);
            }
        };

        // ---- P0: dT = ds * w3 * silu'(pre3) -> tile;  column sums d bc = sum dT, d w3 = sum ds silu(pre3)
        {
            int cx_ = cx, lb_ = lbase;
            FE_OPAQUE(cx_); FE_OPAQUE(lb_);
            uint2 zq[4][MI];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) zq[j][i] = fe4_ldq(&z3f[FE_FOFF(i, j)]);
            float dsr[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) dsr[i] = sDs[i * 16 + fr];
            float4 wq4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wq4[j] = *(const float4*)(w3 + fcol + j * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const fe_f2 w01 = fe_f2{wq4[j].x, wq4[j].y}, w23 = fe_f2{wq4[j].z, wq4[j].w};
                fe_f2 a3a = fe_f2{0.f, 0.f}, a3b = a3a, aca = a3a, acb = a3a;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    fe_f2 m01, d01, m23, d23;
                    fe3_silu_pair(fe3_unpack(zq[j][i].x), m01, d01);
                    fe3_silu_pair(fe3_unpack(zq[j][i].y), m23, d23);
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

        // ---- C3: dM = dT Wc   (the dT tile leaves as the operand of d Wc = dT^T M)
        if (!(xf & 1)) copy_out(p.dT);
        f32x4_t acc[MI][4];
        fe_zero(acc);
        if (!(xf & 8)) fe_gemm_rolled<H, MI, true, SWZ>(sX, WcTp, wave, lane, acc);
        __syncthreads();                                          // every wave has finished reading dT

        // ---- E3: dP2 = (dM + dagg[row]) * silu'(pre2) -> tile in place   (dagg and silu'(pre2) fetched one 16-feature block ahead)
        {
            int cx_ = cx, lb_ = lbase;
            FE_OPAQUE(cx_); FE_OPAQUE(lb_);
            float4 dg[FE4_E3_DB ? 2 : 1][MI];                     // dagg[row of edge i*16 + fr][fcol + j*16 ..]
            uint2 dq[FE4_E3_DB ? 2 : 1][MI];                      // silu'(pre2) quads (i, j) as the forward stored them
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
                    dq[slot][i] = fe4_ldq(&d2f[FE_FOFF(i, j)]);
                }
            };
            if (FE4_E3_DB) fetch_block(0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (FE4_E3_DB) { if (j < 3) fetch_block(j + 1, (j + 1) & 1); } else fetch_block(j, 0);
                fe_f2 a2a = fe_f2{0.f, 0.f}, a2b = a2a;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float4 g4 = dg[FE4_E3_DB ? (j & 1) : 0][i];
                    const uint2 dd = dq[FE4_E3_DB ? (j & 1) : 0][i];
                    const fe_f2 p01 = (fe_f2{acc[i][j][0], acc[i][j][1]} + fe_f2{g4.x, g4.y} * dgk[i]) * fe3_unpack(dd.x);
                    const fe_f2 p23 = (fe_f2{acc[i][j][2], acc[i][j][3]} + fe_f2{g4.z, g4.w} * dgk[i]) * fe3_unpack(dd.y);
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

        // ---- C4: dS1 = dP2 W2   (the dP2 tile leaves as the operand of d W2 = dP2^T S1)
        if (!(xf & 1)) copy_out(p.dP2);
        fe_zero(acc);
        if (!(xf & 8)) fe_gemm_rolled<H, MI, true, SWZ>(sX, W2Tp, wave, lane, acc);
        __syncthreads();
        {
            int cx_ = cx, lb_ = lbase;
            FE_OPAQUE(cx_); FE_OPAQUE(lb_);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(acc[i][j][0], acc[i][j][1]), pack2_bf16(acc[i][j][2], acc[i][j][3]));
        }
        __syncthreads();                                          // dS1 ready

        // ---- P5 (row-wise): pre1 = A[row] + B[col] + rhohat w_r gathered again;  S1 = silu(pre1) -> global (operand of d W2);
        //      dP1 = dS1 * silu'(pre1) -> tile in place and global (the caller's sending-side sum);  d rhohat = dP1 . w_r
        const bool has_next = tile + t_step < t_end;
        {
            int ch_ = ch, es_ = wave * EPW + sub;                 // this lane's chunk and its edge within an iteration
            FE_OPAQUE(ch_); FE_OPAQUE(es_);
#define ch ch_
            fe_f2 wv2[4];
            {
                const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
                wv2[0] = fe_f2{w0.x, w0.y}; wv2[1] = fe_f2{w0.z, w0.w}; wv2[2] = fe_f2{w1.x, w1.y}; wv2[3] = fe_f2{w1.z, w1.w};
            }
            const __amdgpu_buffer_rsrc_t rs_s1 = fe_rsrc((bf16_t*)p.S1 + (size_t)e0 * H, (unsigned)ne * H * 2);
            const __amdgpu_buffer_rsrc_t rs_p1 = fe_rsrc((bf16_t*)p.dP1 + (size_t)e0 * H, (unsigned)ne * H * 2);
            constexpr int PB = (NIT % FE4_P5_BATCH == 0) ? FE4_P5_BATCH : NIT;      // gathered rows in flight per batch
#pragma unroll
            for (int u0 = 0; u0 < NIT; u0 += PB) {
            uint4 ga[PB], gb[PB];
#pragma unroll
            for (int v = 0; v < PB; ++v) {                        // the gathered rows of a batch in flight before the first is used
                const int e = (u0 + v) * EPI + es_;
                const unsigned r = (unsigned)max(sRow[e], 0), c = (unsigned)sCol[e];
                ga[v] = fe3_bload16(rs_ab, (r * uld + ch * 8) * 2, 0);
                gb[v] = fe3_bload16(rs_ab, (c * uld + H + ch * 8) * 2, 0);
            }
#pragma unroll
            for (int v = 0; v < PB; ++v) {
                const int u = v;
                const int e = (u0 + v) * EPI + es_;
                const float rh = sRh[e];
                uint4* sp = (uint4*)&sX[e * H + ((ch ^ (e & SWZ)) * 8)];
                const uint4 gs = *sp;
                const uint32_t ua[4] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w}, ub[4] = {gb[u].x, gb[u].y, gb[u].z, gb[u].w};
                const uint32_t ug[4] = {gs.x, gs.y, gs.z, gs.w};
                uint32_t po[4], pg[4];
                fe_f2 dot2 = fe_f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const fe_f2 z = fe3_unpack(ua[k]) + fe3_unpack(ub[k]) + wv2[k] * rh;
                    fe_f2 m, d;
                    fe3_silu_pair(z, m, d);
                    const fe_f2 g = fe3_unpack(ug[k]) * d;
                    dot2 += g * wv2[k];
                    po[k] = fe3_pack(m);
                    pg[k] = fe3_pack(g);
                }
                *sp = make_uint4(pg[0], pg[1], pg[2], pg[3]);
                const unsigned goff = (unsigned)(e * H + ch * 8) * 2;        // (rows >= ne: outside the descriptors, dropped)
                if (!(xf & 2)) {
                    __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{po[0], po[1], po[2], po[3]}, rs_s1, (int)goff, 0, /*aux: nt*/ 2);
                    __builtin_amdgcn_raw_buffer_store_b128(fe_u32x4_t{pg[0], pg[1], pg[2], pg[3]}, rs_p1, (int)goff, 0, /*aux: nt*/ 2);
                }
                float dot = dot2.x + dot2.y;
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
#undef ch
        }
        __syncthreads();                                          // dP1 ready

        // ---- P6 (one column per thread): receiving-side segment sum of dP1 (deterministic) and the d w_r column sum; the tables of the
        //      next tile are fetched underneath it
        if (has_next) fetch_tables(tile + t_step);
        if (!(xf & 4)) {
            const bool head_cont = e0 > 0 && p.row[e0 - 1] == sRow[0];
            const bool tail_cont = e0 + ne < p.E && p.row[e0 + ne] == sRow[ne - 1];
#if FE4_SCAN64
            sPart[3 * H + tid] += fe_scan_runs64<H, true, SWZ>(sX, sRow, sRh, ne, head_cont, tail_cont, p.dAB16 ? nullptr : p.dABrow, (unsigned)p.lddab,
                                                              p.bnd + (size_t)tile * 2 * H, tid, (bf16_t*)p.dAB16, (unsigned)p.lddab16);
#else
            sPart[3 * H + tid] += fe_scan_rows<H, true, BM, SWZ>(sX, sRow, sRh, ne, head_cont, tail_cont, p.dAB16 ? nullptr : p.dABrow, (unsigned)p.lddab,
                                                                p.bnd + (size_t)tile * 2 * H, tid, (bf16_t*)p.dAB16, (unsigned)p.lddab16);
#endif
        }
        if (has_next) commit_tables(hb ^ 1);
        __syncthreads();                                          // the next tile overwrites the tile; its tables are in place
    }
    // per-work-group partial column sums: part[blockIdx.x][{b2, bc, w3, w_r}][H]   (each sPart slot has ONE writer lane)
    __syncthreads();
    float* part = p.part + (size_t)blockIdx.x * 4 * H;
    for (int k = 0; k < 4; ++k) part[k * H + tid] = sPart[k * H + tid];
#undef FE_QOFF
#undef FE_FOFF
#undef FE_OPAQUE
}

// development knob: bit mask of work the kernel SKIPS (wrong results; sensitivity timing only): 1 dT / dP2 copy-outs, 2 S1 / dP1 stores,
// 4 row scan, 8 both contractions
static int g_fe_bwd4_exp = 0;
extern "C" void fabind_gcl_edge_fused_bwd4_set_exp(int m) { g_fe_bwd4_exp = m; }

// launched from fabind_gcl_edge_fused_bwd (fused_edge.hip) when the arguments carry the forward's d2f / z3f
int fe_bwd4_launch(const FabindEdgeBwdArgs* a, int H, int n_groups, hipStream_t stream) {
    FB_REQUIRE(a->bnd != nullptr, "fabind_gcl_edge_fused_bwd: bnd (ceil(E / 64) x 2 x H floats of scratch) is required");
    FB_REQUIRE(a->lddagg % 4 == 0 && ((uintptr_t)a->dagg & 15) == 0, "fabind_gcl_edge_fused_bwd: dagg must be 16-byte aligned with lddagg % 4 == 0");
    const size_t lds = (size_t)FE_BM * H * 2 + (size_t)8 * FE_BM * sizeof(int) + (size_t)4 * H * sizeof(float);
#define FE_LAUNCH4(HH)                                                                                                \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) {                                                                                               \
            const hipError_t e_ = hipFuncSetAttribute((const void*)gcl_edge_fused_bwd4_kernel<HH>,                    \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e_ != hipSuccess) { fabind_set_error(hipGetErrorString(e_)); return (int)e_; }                     \
            set_ = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((gcl_edge_fused_bwd4_kernel<HH>), dim3(n_groups), dim3(HH), lds, stream, *a, g_fe_bwd4_exp); \
    } while (0)
    if (H == 512) FE_LAUNCH4(512); else if (H == 256) FE_LAUNCH4(256); else if (H == 128) FE_LAUNCH4(128); else FE_LAUNCH4(64);
#undef FE_LAUNCH4
    hipLaunchKernelGGL((fe_boundary_fix_kernel<64>), dim3((a->E + 63) / 64), dim3(H < 256 ? H : 256), 0, stream, a->row, a->E, H,
                       a->bnd, a->dAB16 ? nullptr : a->dABrow, (unsigned)a->lddab, (bf16_t*)a->dAB16, (unsigned)a->lddab16);
    FB_CHECK_LAUNCH();
    return 0;
}
