// Backward of the fused intra-graph edge pipeline (see fused_edge.hip for the algebra), third form: ONE in-place [BM][H] LDS tile
// of BM = 128 edges per work-group (one work-group per CU at H = 512), with
//
//   * the OPERAND-SWAPPED matrix-core form  C^T = W x X^T  in all four contractions: a lane's accumulator quad is then four
//     consecutive FEATURES of one edge (not four consecutive edges of one feature), so every epilogue writes its results to the
//     LDS tile as one 8-byte store per quad, reads its biases / w3 as one 16-byte load per 16-column block and the message
//     gradient dagg[row] as 16-byte loads -- a quarter of the LDS-store / global-load instructions and half of the conversions of
//     the row-quad form (whose epilogues are 45 % of that kernel's cycles);
//   * a 128-edge tile: one weight-fragment load from L2 feeds EIGHT matrix-core instructions (four with 64 edges) -- half the L2
//     weight stream and half the barriers per edge;
//   * ROW-WISE gather phases: a wave handles one edge at a time, its 64 lanes the edge's 64 16-byte chunks, so the two gathered
//     AB rows, the S1 / dP1 tile rows and the scratch rows are full 1 KiB coalesced accesses (the per-thread-strip layout of
//     the other forms touches 64 different cache lines per load instruction);
//   * silu'(pre1) is kept (bf16, like silu'(pre2)) in a per-work-group scratch slab written in P0 and read back in P5 instead
//     of gathering both AB rows a second time and re-evaluating the sigmoid.
//
// Same inputs, outputs and per-element arithmetic as gcl_edge_fused_bwd_kernel except that silu'(pre1) passes through bf16.
#include "common.h"
#include "fabind_hip.h"
#include "fused_common.h"

// sum over the 16 lanes of a DPP row (lanes 16q .. 16q+15); every lane of the row receives the total
__device__ __forceinline__ float fe_row16_sum(float v) {
#define FE_DPP_ADD(ctrl_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), (ctrl_), 0xF, 0xF, true))
    FE_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    FE_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    FE_DPP_ADD(0x141);       // row_half_mirror
    FE_DPP_ADD(0x140);       // row_mirror
#undef FE_DPP_ADD
    return v;
}

template <int H, int BM, bool DBG>
__global__ __launch_bounds__(H, (BM == 128 ? 2 : 4)) void gcl_edge_fused_bwd2_kernel(const FabindEdgeBwdArgs p, const int xf) {
    constexpr int MI = BM / 16;                                   // 16-edge MFMA blocks per wave (the wave owns 64 features of ALL edges)
    constexpr int MQ = MI / 4;
    constexpr int NW = H / 64;
    constexpr int CH = H / 8;                                     // 16-byte chunks per edge row
    constexpr int EPW = 64 / CH;                                  // edges per wave and iteration of the row-wise phases (1 at H = 512)
    constexpr int EPI = NW * EPW;                                 // = 8 edges per work-group iteration
    constexpr int NIT = BM / EPI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [BM][H] swizzled: S1 -> M -> dT -> dP2 -> dS1 -> dP1
    int* sRow = (int*)(sX + BM * H);
    float* sDs = (float*)(sRow + BM);
    float* sRh = sDs + BM;
    float* sPart = sRh + BM;                                      // [4][H]: column sums of {d b2, d bc, d w3, d w_r} over this work-group's tiles
    const bf16_t* __restrict__ AB = (const bf16_t*)p.AB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int sub = lane / CH, ch = lane % CH;                    // row-wise phases: this lane's edge within the wave's group, its chunk
    const int n_tiles = (p.E + BM - 1) / BM;
    const uint32_t thr16 = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);
    for (int k = 0; k < 4; ++k) sPart[k * H + tid] = 0.f;
    const unsigned uld = (unsigned)p.ldab;
    // scratch slabs of this work-group: silu'(pre2) in fragment order, silu'(pre1) in tile order
    uint4* d2s = (uint4*)p.d2scratch + (size_t)blockIdx.x * (BM * H / 8);
    uint4* d1s = (uint4*)p.d2scratch + ((size_t)gridDim.x + blockIdx.x) * (BM * H / 8);
    // LDS element offset of this lane's accumulator quad (i, j): edge i*16 + fr, features wave*64 + j*16 + cq*4 .. +3
    //   chunk = (wave*8 + j*2 + (cq>>1)) ^ (fr & 7) = cx ^ (j*2)
    const int cx = (wave * 8 + (cq >> 1)) ^ (fr & 7);
    const int lbase = fr * H + (cq & 1) * 4;
#define FE_QOFF(i, j) (lbase + (i) * 16 * H + ((cx ^ ((j) * 2)) * 8))
    const int fcol = wave * 64 + cq * 4;                          // first feature of the quad of block j: fcol + j*16

    int t_first, t_end, t_step;
    if (p.xcd_aware && (gridDim.x & 7) == 0) {
        const int tpx = (n_tiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * tpx + (int)(blockIdx.x >> 3);
        t_end = min(n_tiles, (xcd + 1) * tpx);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t_first = blockIdx.x; t_end = n_tiles; t_step = gridDim.x;
    }
    // Start skew: the persistent work-groups do identical work per tile, so without it all 256 CUs reach every tile-store phase
    // together and the stores of a phase (128 KiB per CU, 32 MiB over the chip: all of L2) drain at HBM rate while every CU waits
    // (a wave's loads return behind its earlier stores).  Work-group b starts ((b / 8) % 16) / 16 of a tile late instead.
    if (const int skew = xf >> 8) {
        const long long until = __builtin_readcyclecounter() + (long long)((blockIdx.x >> 3) & 15) * skew;
        while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(32);
    }
    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = 0;
#define FE_TICK(k_)                                                   \
    if constexpr (DBG) {                                              \
        const long long now_ = __builtin_readcyclecounter();         \
        tph[k_] += now_ - tlast;                                      \
        tlast = now_;                                                 \
    }

    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int e0 = tile * BM;
        const int ne = min(BM, p.E - e0);
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
        const int* grow = p.row + e0;
        const int* gcol = p.col + e0;
        const float* grh = p.rhohat + e0;
        const float* gds = p.ds + e0;
        if constexpr (DBG) tlast = __builtin_readcyclecounter();
        const int es = (xf & 128) ? (int)blockIdx.x * BM : e0;    // experiment: every tile of this work-group is stored over the same rows
        for (int t = tid; t < BM; t += H) {
            const bool ok = t < ne;
            sRow[t] = ok ? grow[(unsigned)t] : -1;
            sDs[t] = ok ? gds[(unsigned)t] : 0.f;
            sRh[t] = ok ? grh[(unsigned)t] : 0.f;
        }
        // ---- P0 (row-wise): S1 = silu(A[row] + B[col] + rhohat w_r) -> sX and the S1 operand; silu'(pre1) -> scratch
        {
            float wv[8];
            {
                const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
                wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
            }
            bf16_t* gS1 = (bf16_t*)p.S1 + (size_t)es * H;
            const __amdgpu_buffer_rsrc_t rS1 = fe_rsrc(gS1, (unsigned)ne * H * 2);
#pragma unroll 4
            for (int it = 0; it < NIT; ++it) {
                const int e = it * EPI + wave * EPW + sub;
                uint4* sp = (uint4*)&sX[e * H + ((ch ^ (e & 7)) * 8)];
                if (e < ne) {
                    const unsigned r = (xf & 4) ? 0u : (unsigned)grow[(unsigned)e], c = (xf & 4) ? 1u : (unsigned)gcol[(unsigned)e];
                    const float rh = grh[(unsigned)e];
                    float fa[8], fb[8], o[8], d[8];
                    unpack8(*(const uint4*)(AB + (r * uld + ch * 8)), fa);
                    unpack8(*(const uint4*)(AB + (c * uld + H + ch * 8)), fb);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float z = fa[k] + fb[k] + rh * wv[k], sg = fe_sigmoid(z);
                        o[k] = z * sg;
                        d[k] = sg * (1.0f + z * (1.0f - sg));
                    }
                    const uint4 u = pack8(o);
                    *sp = u;
                    if (!(xf & 1)) { if (xf & 32) *(uint4*)&gS1[(unsigned)(e * H + ch * 8)] = u; else fe_store16_stream(rS1, (unsigned)(e * H + ch * 8) * 2, u); }
                    if (!(xf & 2)) d1s[e * CH + ch] = pack8(d);
                } else {
                    *sp = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        __syncthreads();
        FE_TICK(0)

        // ---- P1: pre2 = S1 W2^T + b2;  M -> sX in place;  silu'(pre2) -> scratch
        f32x4_t acc[MI][4];
        fe_zero(acc);
        fe_gemm_rolled<H, MI, true>(sX, W2p, wave, lane, acc);
        __syncthreads();                                          // every wave has finished reading S1
        FE_TICK(2)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bq = *(const float4*)(b2 + fcol + j * 16);
                const float bj[4] = {bq.x, bq.y, bq.z, bq.w};
                uint32_t da[MI], db[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float m[4], dd[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][j][r] + bj[r], sg = fe_sigmoid(z);
                        float kp = 1.0f;
                        if (thr16) kp = fe_keep(p.seed, (uint32_t)(e0 + i * 16 + fr), (uint32_t)(fcol + j * 16 + r), H, thr16, dscale);
                        dd[r] = kp * (sg * (1.0f + z * (1.0f - sg)));
                        m[r] = kp * (z * sg);
                    }
                    da[i] = pack2_bf16(dd[0], dd[1]);
                    db[i] = pack2_bf16(dd[2], dd[3]);
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(m[0], m[1]), pack2_bf16(m[2], m[3]));
                }
                if (!(xf & 2))
#pragma unroll
                for (int q = 0; q < MQ; ++q) {
                    d2s[((j * 2) * MQ + q) * H + tid] = make_uint4(da[q * 4], da[q * 4 + 1], da[q * 4 + 2], da[q * 4 + 3]);
                    d2s[((j * 2 + 1) * MQ + q) * H + tid] = make_uint4(db[q * 4], db[q * 4 + 1], db[q * 4 + 2], db[q * 4 + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        FE_TICK(3)
        if (!(xf & 1) && !(xf & 64)) { if (xf & 32) fe_tile_store<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.Mm), es, ne, tid); else fe_tile_store_stream<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.Mm), es, ne, tid); }
        FE_TICK(1)

        // ---- P2: pre3 = M Wc^T + bc;  dT = ds * w3 * silu'(pre3) -> sX in place
        fe_zero(acc);
        fe_gemm_rolled<H, MI, true>(sX, Wcp, wave, lane, acc);
        float4 bq4[4], wq4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bq4[j] = *(const float4*)(bc + fcol + j * 16); wq4[j] = *(const float4*)(w3 + fcol + j * 16); }
        __builtin_amdgcn_sched_barrier(0);
        if (!(xf & 1) && (xf & 64)) fe_tile_store_stream<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.Mm), es, ne, tid);
        __syncthreads();
        FE_TICK(2)
        {
            float dsr[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) dsr[i] = sDs[i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bq = bq4[j], wq = wq4[j];
                const float bj[4] = {bq.x, bq.y, bq.z, bq.w}, wj[4] = {wq.x, wq.y, wq.z, wq.w};
                float a3[4] = {0.f, 0.f, 0.f, 0.f}, ac[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float dt[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][j][r] + bj[r], sg = fe_sigmoid(z);
                        dt[r] = dsr[i] * wj[r] * (sg * (1.0f + z * (1.0f - sg)));
                        a3[r] += dsr[i] * (z * sg);
                        ac[r] += dt[r];
                    }
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(dt[0], dt[1]), pack2_bf16(dt[2], dt[3]));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { a3[r] = fe_row16_sum(a3[r]); ac[r] = fe_row16_sum(ac[r]); }
                if (fr < 4) {                                     // lane fr = r owns feature fcol + j*16 + r of sPart (one writer per slot)
                    const float va = fr == 0 ? a3[0] : fr == 1 ? a3[1] : fr == 2 ? a3[2] : a3[3];
                    const float vc = fr == 0 ? ac[0] : fr == 1 ? ac[1] : fr == 2 ? ac[2] : ac[3];
                    sPart[2 * H + fcol + j * 16 + fr] += va;
                    sPart[1 * H + fcol + j * 16 + fr] += vc;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        FE_TICK(4)
        if (!(xf & 1)) { if (xf & 32) fe_tile_store<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.dT), es, ne, tid); else fe_tile_store_stream<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.dT), es, ne, tid); }
        FE_TICK(1)

        // ---- P3: dM = dT Wc + dagg[row];  dP2 = dM * silu'(pre2) -> sX in place
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rr = (xf & 8) ? 0 : sRow[i * 16 + fr];
            const float keep = rr >= 0 ? 1.f : 0.f;
            const float* dg = p.dagg + ((unsigned)max(rr, 0) * (unsigned)p.lddagg + fcol);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *(const float4*)(dg + j * 16);
                acc[i][j] = f32x4_t{v.x * keep, v.y * keep, v.z * keep, v.w * keep};
            }
        }
        FE_TICK(5)
        fe_gemm_rolled<H, MI, true>(sX, WcTp, wave, lane, acc);
        __syncthreads();
        FE_TICK(2)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t da[MI], db[MI];
#pragma unroll
                for (int q = 0; q < MQ; ++q) {
                    const uint4 qa = d2s[((j * 2) * MQ + q) * H + tid], qb = d2s[((j * 2 + 1) * MQ + q) * H + tid];
                    da[q * 4] = qa.x; da[q * 4 + 1] = qa.y; da[q * 4 + 2] = qa.z; da[q * 4 + 3] = qa.w;
                    db[q * 4] = qb.x; db[q * 4 + 1] = qb.y; db[q * 4 + 2] = qb.z; db[q * 4 + 3] = qb.w;
                }
                float a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float dp[4];
                    dp[0] = acc[i][j][0] * __uint_as_float(da[i] << 16);
                    dp[1] = acc[i][j][1] * __uint_as_float(da[i] & 0xffff0000u);
                    dp[2] = acc[i][j][2] * __uint_as_float(db[i] << 16);
                    dp[3] = acc[i][j][3] * __uint_as_float(db[i] & 0xffff0000u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a2[r] += dp[r];
                    *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(dp[0], dp[1]), pack2_bf16(dp[2], dp[3]));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[r] = fe_row16_sum(a2[r]);
                if (fr < 4) sPart[fcol + j * 16 + fr] += fr == 0 ? a2[0] : fr == 1 ? a2[1] : fr == 2 ? a2[2] : a2[3];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        FE_TICK(6)
        if (!(xf & 1) && !(xf & 64)) { if (xf & 32) fe_tile_store<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.dP2), es, ne, tid); else fe_tile_store_stream<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.dP2), es, ne, tid); }
        FE_TICK(1)

        // ---- P4: dS1 = dP2 W2 -> sX in place
        fe_zero(acc);
        fe_gemm_rolled<H, MI, true>(sX, W2Tp, wave, lane, acc);
        if (!(xf & 1) && (xf & 64)) fe_tile_store_stream<H>(sX, (bf16_t*)((xf & 128) ? p.S1 : p.dP2), es, ne, tid);
        __syncthreads();
        FE_TICK(2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                *(uint2*)&sX[FE_QOFF(i, j)] = make_uint2(pack2_bf16(acc[i][j][0], acc[i][j][1]), pack2_bf16(acc[i][j][2], acc[i][j][3]));
        __syncthreads();
        FE_TICK(7)

        // ---- P5 (row-wise): dP1 = dS1 * silu'(pre1) -> sX in place and the dP1 operand;  d rhohat = dP1 . w_r
        {
            float wv[8];
            {
                const float4 w0 = *(const float4*)(w_r + ch * 8), w1 = *(const float4*)(w_r + ch * 8 + 4);
                wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
            }
            bf16_t* gP1 = (bf16_t*)((xf & 128) ? p.S1 : p.dP1) + (size_t)es * H;
            const __amdgpu_buffer_rsrc_t rP1 = fe_rsrc(gP1, (unsigned)ne * H * 2);
#pragma unroll 4
            for (int it = 0; it < NIT; ++it) {
                const int e = it * EPI + wave * EPW + sub;
                float dot = 0.f;
                if (e < ne) {
                    uint4* sp = (uint4*)&sX[e * H + ((ch ^ (e & 7)) * 8)];
                    float g[8], d[8];
                    unpack8(*sp, g);
                    unpack8(d1s[e * CH + ch], d);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        g[k] *= d[k];
                        dot += g[k] * wv[k];
                    }
                    const uint4 u = pack8(g);
                    *sp = u;
                    if (!(xf & 1)) { if (xf & 32) *(uint4*)&gP1[(unsigned)(e * H + ch * 8)] = u; else fe_store16_stream(rP1, (unsigned)(e * H + ch * 8) * 2, u); }
                }
#pragma unroll
                for (int o = 1; o < CH && o < 64; o <<= 1) dot += __shfl_xor(dot, o, 64);
                if (ch == 0 && e < ne) (p.drh + e0)[(unsigned)e] = dot;
            }
        }
        __syncthreads();
        FE_TICK(8)

        // ---- P6 (one column per thread): receiving-side segment sum of dP1 (deterministic, fe_scan_rows) and the d w_r column sum
        {
            const bool head_cont = e0 > 0 && p.row[e0 - 1] == sRow[0];
            const bool tail_cont = e0 + ne < p.E && p.row[e0 + ne] == sRow[ne - 1];
            if (!(xf & 16)) sPart[3 * H + tid] += fe_scan_rows<H, true, BM>(sX, sRow, sRh, ne, head_cont, tail_cont, p.dABrow, (unsigned)p.lddab,
                                                         p.bnd + (size_t)tile * 2 * H, tid);
        }
        __syncthreads();                                          // the next tile overwrites sX / sRow
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

// development knob: bit mask of work the kernel SKIPS (wrong results; sensitivity timing only): 1 operand stores, 2 scratch
// stores, 4 gathers collapse onto two rows, 8 dagg loads collapse onto one row, 16 no row scan
static int g_fe_bwd2_exp = 0;
extern "C" void fabind_gcl_edge_fused_bwd2_set_exp(int m) { g_fe_bwd2_exp = m; }

// launched from fabind_gcl_edge_fused_bwd (fused_edge.hip), variant 3 (BM = 128) / 4 (BM = 64, two work-groups per CU)
int fe_bwd2_launch(const FabindEdgeBwdArgs* a, int H, int bm, int n_groups, hipStream_t stream) {
    FB_REQUIRE(a->d2scratch != nullptr, "fabind_gcl_edge_fused_bwd: the row-wise variants need d2scratch (2 x n_groups x tile x H x 2 bytes)");
    FB_REQUIRE(a->bnd != nullptr, "fabind_gcl_edge_fused_bwd: bnd (ceil(E / tile) x 2 x H floats of scratch) is required");
    FB_REQUIRE(a->lddagg % 4 == 0 && ((uintptr_t)a->dagg & 15) == 0, "fabind_gcl_edge_fused_bwd: dagg must be 16-byte aligned with lddagg % 4 == 0");
    const size_t lds = (size_t)bm * H * 2 + bm * (sizeof(int) + 2 * sizeof(float)) + (size_t)4 * H * sizeof(float);
#define FE_LAUNCH3(HH, BB, DD)                                                                                     \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) {                                                                                               \
            const hipError_t e_ = hipFuncSetAttribute((const void*)gcl_edge_fused_bwd2_kernel<HH, BB, DD>,         \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e_ != hipSuccess) { fabind_set_error(hipGetErrorString(e_)); return (int)e_; }                     \
            set_ = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((gcl_edge_fused_bwd2_kernel<HH, BB, DD>), dim3(n_groups), dim3(HH), lds, stream, *a, g_fe_bwd2_exp); \
    } while (0)
    if (bm == 128) {
        if (a->dbg != nullptr && H == 512) FE_LAUNCH3(512, 128, true);
        else if (H == 512) FE_LAUNCH3(512, 128, false); else if (H == 256) FE_LAUNCH3(256, 128, false);
        else if (H == 128) FE_LAUNCH3(128, 128, false); else FE_LAUNCH3(64, 128, false);
        hipLaunchKernelGGL((fe_boundary_fix_kernel<128>), dim3((a->E + 127) / 128), dim3(H < 256 ? H : 256), 0, stream, a->row, a->E, H,
                           a->bnd, a->dABrow, (unsigned)a->lddab);
    } else {
        if (a->dbg != nullptr && H == 512) FE_LAUNCH3(512, 64, true);
        else if (H == 512) FE_LAUNCH3(512, 64, false); else if (H == 256) FE_LAUNCH3(256, 64, false);
        else if (H == 128) FE_LAUNCH3(128, 64, false); else FE_LAUNCH3(64, 64, false);
        hipLaunchKernelGGL((fe_boundary_fix_kernel<64>), dim3((a->E + 63) / 64), dim3(H < 256 ? H : 256), 0, stream, a->row, a->E, H,
                           a->bnd, a->dABrow, (unsigned)a->lddab);
    }
#undef FE_LAUNCH3
    FB_CHECK_LAUNCH();
    return 0;
}
