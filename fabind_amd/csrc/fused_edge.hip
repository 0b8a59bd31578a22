// Fused intra-graph edge pipeline of MC_E_GCL (reference models/egnn.py:68-128), bf16 path -- forward kernel (its
// adjoint, which recomputes these tensors tile by tile, follows below):
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
#include "fused_common.h"

static int g_fe_xcd_aware = 1;       // development knob: XCD-aware tile order (1) vs linear / grid-stride order (0)
extern "C" void fabind_gcl_edge_fused_set_xcd_aware(int on) { g_fe_xcd_aware = on ? 1 : 0; }
static int g_fe_fwd_variant = 1;     // 1 = row-wise gather / operand-swapped epilogues (fused_edge_fwd2.hip): the only form built since round 5
                                     // (round 1's kernel, variant 0, was a knob nobody reached: retired, git history up to 8fbe227)
extern "C" int fabind_gcl_edge_fused_set_variant(int v) {
    FB_REQUIRE(v == 1, "fabind_gcl_edge_fused_set_variant: 1 (row-wise / operand-swapped) is the only forward form built (variant 0 was retired in round 5)");
    g_fe_fwd_variant = v;
    return 0;
}
extern "C" int fabind_gcl_edge_fused_variant(void) { return g_fe_fwd_variant; }

static int fe_fwd_entry(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                        const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                        const float* w3, int E, float* agg, float* s_out, float p_drop, unsigned seed,
                        float* bnd, void* agg16, void* Msave, void* d2f, void* z3f, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE((Msave != nullptr) == (d2f != nullptr) && (Msave != nullptr) == (z3f != nullptr),
               "fabind_gcl_edge_fused_train: M, d2f and z3f come together");
    FB_REQUIRE(Msave == nullptr || g_fe_fwd_variant == 1, "fabind_gcl_edge_fused_train: the saving forward is the row-wise / operand-swapped kernel (variant 1)");
    FB_REQUIRE(bnd != nullptr, "fabind_gcl_edge_fused: bnd (ceil(E/64) x 2 x H floats of scratch) is required");
    FB_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "fabind_gcl_edge_fused: p_drop in [0, 1)");
    const uint32_t thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_gcl_edge_fused: H must be 64, 128, 256 or 512");
    FB_REQUIRE(ldab % 8 == 0, "fabind_gcl_edge_fused: ldab % 8");
    {
        const int rc = fe_fwd2_launch(AB, ldab, H, row, col, rhohat, w_r, W2p, b2, Wcp, bc, w3, E, agg, s_out, thr16, dscale, seed, bnd,
                                      agg16, g_fe_xcd_aware, stream, Msave, d2f, z3f);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((fe_boundary_fix_kernel<FE_BM>), dim3((E + FE_BM - 1) / FE_BM), dim3(H < 256 ? H : 256), 0, stream, row, E, H,
                       bnd, agg, (unsigned)H, (bf16_t*)agg16, (unsigned)H);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_gcl_edge_fused(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                                     const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                                     const float* w3, int E, float* agg, float* s_out, float p_drop, unsigned seed,
                                     float* bnd, void* agg16, hipStream_t stream) {
    return fe_fwd_entry(AB, ldab, H, row, col, rhohat, w_r, W2p, b2, Wcp, bc, w3, E, agg, s_out, p_drop, seed, bnd, agg16, nullptr, nullptr,
                        nullptr, stream);
}
extern "C" int fabind_gcl_edge_fused_train(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                                           const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                                           const float* w3, int E, float* agg, float* s_out, float p_drop, unsigned seed,
                                           float* bnd, void* agg16, void* M, void* d2f, void* z3f, hipStream_t stream) {
    FB_REQUIRE(M != nullptr && d2f != nullptr && z3f != nullptr, "fabind_gcl_edge_fused_train: M, d2f, z3f are required");
    return fe_fwd_entry(AB, ldab, H, row, col, rhohat, w_r, W2p, b2, Wcp, bc, w3, E, agg, s_out, p_drop, seed, bnd, agg16, M, d2f, z3f, stream);
}

static int fe_x3_entry(const float* AB, int ldab, int n_rows, int H, const int* row, const int* col, const float* rhohat,
                       const float* w_r, const void* W2ph, const void* W2pl, const float* b2, const void* Wcph,
                       const void* Wcpl, const float* bc, const float* w3, int E, float* agg, float* s_out,
                       float p_drop, unsigned seed, float* bnd, void* Msave, void* d2f, void* z3f, hipStream_t stream) {
    if (E <= 0) return 0;
    FB_REQUIRE(bnd != nullptr, "fabind_gcl_edge_fused_x3: bnd (ceil(E/64) x 2 x H floats of scratch) is required");
    FB_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "fabind_gcl_edge_fused_x3: p_drop in [0, 1)");
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_gcl_edge_fused_x3: H must be 64, 128, 256 or 512");
    FB_REQUIRE(ldab % 4 == 0 && ((uintptr_t)AB & 15) == 0, "fabind_gcl_edge_fused_x3: AB must be 16-byte aligned with ldab % 4 == 0");
    // the gathers address AB rows with 32-bit BYTE offsets into one buffer resource: fp32 rows are twice the bf16 kernel's
    FB_REQUIRE(n_rows >= 0 && (size_t)n_rows * (size_t)ldab * 4 < ((size_t)1 << 32), "fabind_gcl_edge_fused_x3: AB larger than 4 GiB (n_rows * ldab * 4 bytes): split the batch");
    const uint32_t thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);
    const int rc = fe_fwd3_launch(AB, ldab, H, row, col, rhohat, w_r, W2ph, W2pl, b2, Wcph, Wcpl, bc, w3, E, agg, s_out, thr16, dscale,
                                  seed, bnd, g_fe_xcd_aware, stream, Msave, d2f, z3f);
    if (rc) return rc;
    hipLaunchKernelGGL((fe_boundary_fix_kernel<FE_BM>), dim3((E + FE_BM - 1) / FE_BM), dim3(H < 256 ? H : 256), 0, stream, row, E, H,
                       bnd, agg, (unsigned)H, (bf16_t*)nullptr, 0u);
    FB_CHECK_LAUNCH();
    return 0;
}
extern "C" int fabind_gcl_edge_fused_x3(const float* AB, int ldab, int n_rows, int H, const int* row, const int* col, const float* rhohat,
                                        const float* w_r, const void* W2ph, const void* W2pl, const float* b2, const void* Wcph,
                                        const void* Wcpl, const float* bc, const float* w3, int E, float* agg, float* s_out,
                                        float p_drop, unsigned seed, float* bnd, hipStream_t stream) {
    return fe_x3_entry(AB, ldab, n_rows, H, row, col, rhohat, w_r, W2ph, W2pl, b2, Wcph, Wcpl, bc, w3, E, agg, s_out, p_drop, seed, bnd,
                       nullptr, nullptr, nullptr, stream);
}
extern "C" int fabind_gcl_edge_fused_x3_train(const float* AB, int ldab, int n_rows, int H, const int* row, const int* col, const float* rhohat,
                                              const float* w_r, const void* W2ph, const void* W2pl, const float* b2, const void* Wcph,
                                              const void* Wcpl, const float* bc, const float* w3, int E, float* agg, float* s_out,
                                              float p_drop, unsigned seed, float* bnd, void* M, void* d2f, void* z3f, hipStream_t stream) {
    FB_REQUIRE(M != nullptr && d2f != nullptr && z3f != nullptr, "fabind_gcl_edge_fused_x3_train: M, d2f, z3f are required");
    return fe_x3_entry(AB, ldab, n_rows, H, row, col, rhohat, w_r, W2ph, W2pl, b2, Wcph, Wcpl, bc, w3, E, agg, s_out, p_drop, seed, bnd,
                       M, d2f, z3f, stream);
}

// =====================================================================================================
// Backward of the same pipeline (training, bf16): nothing per-edge was saved by the forward kernel.
// Per 64-edge tile the work-group recomputes S1, pre2, M, pre3 and chains the four H x H contractions
//
//   S1 -> pre2 = S1 W2^T + b2 -> M = silu(pre2)            (recompute, keeps silu'(pre2) in registers)
//   pre3 = M Wc^T + bc;  dT  = ds * w3 * silu'(pre3)        (ds = gradient of the per-edge scalar s)
//   dM   = dT Wc + dagg[row];   dP2 = dM * silu'(pre2)      (dagg = gradient of the segment sum)
//   dS1  = dP2 W2;              dP1 = dS1 * silu'(pre1)
//
// with S1 / dT / dS1 / dP1 passing through one LDS tile and M / dP2 through a second one.  It writes the
// bf16 operands of the two weight-gradient contractions (S1, M, dT, dP2: consumed by fabind_gemm_tn) and
// dP1 (sending-side segment sum) once each -- 5 [E,H] writes and no [E,H] read, instead of the ~20 round
// trips of the unfused backward -- plus d rhohat, the receiving-side segment sum of dP1 (run-length scan,
// boundary runs finish with atomics) and per-work-group partial column sums for d b2, d bc, d w3, d w_r.
// Persistent work-groups (one per CU: 2 x 64 KiB LDS tiles at H = 512) walk the tiles grid-stride.
// DBG: per-phase cycle counters of work-group 0 (tools/edge_bwd_phases.py).  A compile-time switch: as a run-time test of p.dbg the
// thirteen 64-bit counters stayed live across the whole tile loop of a kernel that sits at its register limit.
template <int H, int BM, bool DBG>
__global__ __launch_bounds__(H, (BM == 32 ? 4 : 2)) void gcl_edge_fused_bwd_kernel(const FabindEdgeBwdArgs p) {
    constexpr int TPE = H / BM;                                   // threads per edge in the gather layout
    constexpr int CPT = BM / 8;                                   // 16-byte chunks per thread there
    constexpr int MI = BM / 16;                                   // 16-row MFMA fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* sX = (bf16_t*)smem;                                   // [64][H] swizzled: S1 -> dT -> dS1 -> dP1
    bf16_t* sY = sX + BM * H;                                  // [64][H] swizzled: M -> dP2
    int* sRow = (int*)(sY + BM * H);
    float* sDs = (float*)(sRow + BM);
    float* sRh = sDs + BM;
    const bf16_t* __restrict__ AB = (const bf16_t*)p.AB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, cq = lane >> 4;
    const int el = tid / TPE, cth = tid % TPE;
    const int n_tiles = (p.E + BM - 1) / BM;
    const uint32_t thr16 = (uint32_t)(p.p_drop * 65536.0f + 0.5f);
    const float dscale = 1.0f / (1.0f - (float)thr16 / 65536.0f);
    // LDS element offset of C-fragment element (i, j, r) of this lane in a swizzled [64][H] tile, as one of four
    // lane-dependent bases plus a compile-time constant (so the 64 stores of an epilogue need 4 address registers):
    //   row = i*16 + cq*4 + r, col = wave*64 + j*16 + fr, chunk = (col>>3) ^ (row&7); bit0 flips with r&1, bit2 with j>>1
    int cbase[2][2];
#pragma unroll
    for (int f0 = 0; f0 < 2; ++f0)
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2)
            cbase[f0][f2] = cq * 4 * H + (wave * 8 + (((cq & 1) ^ f2) << 2) + ((fr >> 3) ^ f0)) * 8 + (fr & 7);
#define FE_COFF(i, j, r) (cbase[(r) & 1][(j) >> 1] + ((i) * 16 + (r)) * H + ((((j) & 1) ^ ((r) >> 1)) * 16))
    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // optional per-phase cycle counts (p.dbg != NULL)
    long long tlast = 0;
#define FE_TICK(k_)                                                   \
    if constexpr (DBG) {                                              \
        const long long now_ = __builtin_readcyclecounter();         \
        tph[k_] += now_ - tlast;                                      \
        tlast = now_;                                                 \
    }
    float pb2[4] = {0.f, 0.f, 0.f, 0.f}, pbc[4] = {0.f, 0.f, 0.f, 0.f}, pw3[4] = {0.f, 0.f, 0.f, 0.f}, pwr = 0.f;

    // Gathered AB rows are prefetched into registers one phase ahead of their use: the rows of the NEXT tile while
    // this tile finishes (dP1 store + P6), and this tile's rows again (for silu'(pre1)) while contraction 4 runs.
    uint4 pfa[CPT], pfb[CPT];
    const unsigned uld = (unsigned)p.ldab;
    auto fe_prefetch = [&](unsigned r, unsigned c) {
        const unsigned aoff = r * uld + cth * (CPT * 8), boff = c * uld + H + cth * (CPT * 8);
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            pfa[q] = *(const uint4*)(AB + (aoff + q * 8));
            pfb[q] = *(const uint4*)(AB + (boff + q * 8));
        }
    };
    // XCD-aware persistent walk (work-group b runs on XCD b % 8): XCD x owns the x-th eighth of the tile range, i.e. whole
    // complexes, and its work-groups walk that range with stride gridDim.x / 8 -- the gathered AB rows of a complex stay in ONE
    // XCD's L2 (the grid-stride walk spread every complex over all eight: 1.46x HBM over-fetch, profiles/r01_pmc.json).
    int t_first, t_end, t_step;
    if (p.xcd_aware && (gridDim.x & 7) == 0) {
        const int tpx = (n_tiles + 7) / 8, xcd = blockIdx.x & 7;
        t_first = xcd * tpx + (int)(blockIdx.x >> 3);
        t_end = min(n_tiles, (xcd + 1) * tpx);
        t_step = (int)(gridDim.x >> 3);
    } else {
        t_first = blockIdx.x; t_end = n_tiles; t_step = gridDim.x;
    }
    unsigned ur = 0, uc = 0;                                      // this lane's edge of the current tile
    if (t_first < t_end) {
        const int e0 = t_first * BM;
        if (el < min(BM, p.E - e0)) {
            ur = (unsigned)(p.row + e0)[(unsigned)el];
            uc = (unsigned)(p.col + e0)[(unsigned)el];
        }
    }
    fe_prefetch(ur, uc);                                          // unconditional (row 0 for idle lanes): a guarded
                                                                  // load would keep the old registers live instead

    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int e0 = tile * BM;
        const int ne = min(BM, p.E - e0);
        // Loop-invariant operands (weights, biases, w_r) must be re-read from L2 per tile, not hoisted into ~140
        // registers that stay live across the whole tile loop: launder their base offsets once per trip.
        int lz = 0;
        asm volatile("" : "+s"(lz));
        const float* w_r = p.w_r + lz;
        const float* b2 = p.b2 + lz;
        const float* bc = p.bc + lz;
        const float* w3 = p.w3 + lz;
        const bf16_t* W2p = (const bf16_t*)p.W2p + lz;
        const bf16_t* Wcp = (const bf16_t*)p.Wcp + lz;
        const bf16_t* W2Tp = (const bf16_t*)p.W2Tp + lz;
        const bf16_t* WcTp = (const bf16_t*)p.WcTp + lz;
        if constexpr (DBG) tlast = __builtin_readcyclecounter();
        // ---- P0: gather + first Linear + SiLU -> sX
        // every global access below is (uniform base in SGPRs) + (32-bit lane offset): 64-bit per-lane pointers
        // are loop-invariant per access site, get hoisted out of the tile loop and spill by the hundred
        const int* grow = p.row + e0;
        const int* gcol = p.col + e0;
        const float* grh = p.rhohat + e0;
        const float* gds = p.ds + e0;
        if (tid < BM) {
            const bool ok = tid < ne;
            sRow[tid] = ok ? grow[(unsigned)tid] : -1;
            sDs[tid] = ok ? gds[(unsigned)tid] : 0.f;
            sRh[tid] = ok ? grh[(unsigned)tid] : 0.f;
        }
        // indices of this lane's edge in the next tile (their rows are fetched after P5)
        unsigned nr = 0, nc = 0;
        const int e0n = e0 + t_step * BM;
        const bool has_next = tile + t_step < t_end && el < min(BM, p.E - e0n);
        if (has_next) {
            nr = (unsigned)(p.row + e0n)[(unsigned)el];
            nc = (unsigned)(p.col + e0n)[(unsigned)el];
        }
        if (el < ne) {
            const float rh = grh[(unsigned)el];
            const float* wp = w_r + (unsigned)(cth * (CPT * 8));
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                float fa[8], fb[8], o[8];
                unpack8(pfa[q], fa);
                unpack8(pfb[q], fb);
                const float4 w0 = *(const float4*)(wp + q * 8), w1 = *(const float4*)(wp + q * 8 + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = fe_silu(fa[k] + fb[k] + rh * wv[k]);
                *(uint4*)&sX[el * H + (((cth * CPT + q) ^ (el & 7)) * 8)] = pack8(o);
            }
        } else {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int q = 0; q < CPT; ++q) *(uint4*)&sX[el * H + (((cth * CPT + q) ^ (el & 7)) * 8)] = z;
        }
        __syncthreads();
        FE_TICK(0)
        fe_tile_store<H>(sX, (bf16_t*)p.S1, e0, ne, tid);
        FE_TICK(1)

        // Epilogues run as ROLLED loops over the wave's four 16-column blocks: slot 0 of acc / d2 / the partial sums
        // is processed, then every array rotates left by one block (a full turn after four trips).  A fully
        // unrolled epilogue lets the scheduler interleave all 64 exp/rcp chains of a lane and spill by the hundred.
#define FE_ROT4(a_) { auto t_ = a_[0]; a_[0] = a_[1]; a_[1] = a_[2]; a_[2] = a_[3]; a_[3] = t_; }
#define FE_ROT_ACC() _Pragma("unroll") for (int i = 0; i < MI; ++i) FE_ROT4(acc[i])
#define FE_ROT_D2() _Pragma("unroll") for (int i = 0; i < MI; ++i) { FE_ROT4(d2a[i]); FE_ROT4(d2b[i]); }
        // LDS offset of element (i, block j, r): cb[r&1] + (i*16 + r)*H + o2[r>>1], cb / o2 recomputed per block
#define FE_JADDR(j_)                                                                          \
        const int cb[2] = {((j_) >> 1) ? cbase[0][1] : cbase[0][0], ((j_) >> 1) ? cbase[1][1] : cbase[1][0]}; \
        const int o2[2] = {((j_) & 1) * 16, 16 - ((j_) & 1) * 16};
#define FE_LOFF(i, r) (cb[(r) & 1] + ((i) * 16 + (r)) * H + o2[(r) >> 1])

        // ---- P1: pre2 = S1 W2^T + b2;  M -> sY;  silu'(pre2) stays in registers (packed bf16)
        f32x4_t acc[MI][4];
        uint32_t d2a[MI][4], d2b[MI][4];
        fe_zero(acc);
        fe_gemm_rolled<H, MI>(sX, W2p, wave, lane, acc);
        FE_TICK(2)
        {
            float bj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bj[j] = b2[wave * 64 + j * 16 + fr];
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                FE_JADDR(j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float dd[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][0][r] + bj[0], sg = fe_sigmoid(z);
                        float kp = 1.0f;
                        if (thr16)
                            kp = fe_keep(p.seed, (uint32_t)(e0 + i * 16 + cq * 4 + r), (uint32_t)(wave * 64 + j * 16 + fr), H,
                                         thr16, dscale);
                        dd[r] = kp * (sg * (1.0f + z * (1.0f - sg)));
                        sY[FE_LOFF(i, r)] = f32_to_bf16(kp * (z * sg));
                    }
                    d2a[i][0] = pack2_bf16(dd[0], dd[1]);
                    d2b[i][0] = pack2_bf16(dd[2], dd[3]);
                }
                FE_ROT_ACC() FE_ROT_D2() FE_ROT4(bj)
            }
        }
        __syncthreads();
        FE_TICK(3)
        fe_tile_store<H>(sY, (bf16_t*)p.Mm, e0, ne, tid);
        FE_TICK(1)

        // ---- P2: pre3 = M Wc^T + bc;  dT = ds * w3 * silu'(pre3) -> sX
        fe_zero(acc);
        fe_gemm_rolled<H, MI>(sY, Wcp, wave, lane, acc);
        FE_TICK(2)
        {
            float bj[4], wj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { bj[j] = bc[wave * 64 + j * 16 + fr]; wj[j] = w3[wave * 64 + j * 16 + fr]; }
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                FE_JADDR(j)
                float a3 = 0.f, ac = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][0][r] + bj[0], sg = fe_sigmoid(z), dsr = sDs[i * 16 + cq * 4 + r];
                        const float dt = dsr * wj[0] * (sg * (1.0f + z * (1.0f - sg)));
                        a3 += dsr * (z * sg);
                        ac += dt;
                        sX[FE_LOFF(i, r)] = f32_to_bf16(dt);
                    }
                pw3[0] += a3;
                pbc[0] += ac;
                FE_ROT_ACC() FE_ROT4(bj) FE_ROT4(wj) FE_ROT4(pw3) FE_ROT4(pbc)
            }
        }
        __syncthreads();
        FE_TICK(4)
        fe_tile_store<H>(sX, (bf16_t*)p.dT, e0, ne, tid);
        FE_TICK(1)

        // ---- P3: dM = dT Wc + dagg[row];  dP2 = dM * silu'(pre2) -> sY
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = sRow[i * 16 + cq * 4 + r];
                const unsigned doff = (unsigned)max(rr, 0) * (unsigned)p.lddagg + wave * 64 + fr;   // + j*16: immediates
                const float keep = rr >= 0 ? 1.f : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = p.dagg[doff + j * 16] * keep;
            }
        FE_TICK(5)
        fe_gemm_rolled<H, MI>(sX, WcTp, wave, lane, acc);
        FE_TICK(2)
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            FE_JADDR(j)
            float a2 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t pk = (r >> 1) ? d2b[i][0] : d2a[i][0];
                    const float dd = __uint_as_float((r & 1) ? (pk & 0xffff0000u) : (pk << 16));
                    const float dp = acc[i][0][r] * dd;
                    a2 += dp;
                    sY[FE_LOFF(i, r)] = f32_to_bf16(dp);
                }
            pb2[0] += a2;
            FE_ROT_ACC() FE_ROT_D2() FE_ROT4(pb2)
        }
        __syncthreads();
        FE_TICK(6)
        fe_tile_store<H>(sY, (bf16_t*)p.dP2, e0, ne, tid);
        FE_TICK(1)

        // ---- P4: dS1 = dP2 W2 -> sX   (this tile's AB rows are fetched again underneath it, for P5)
        fe_prefetch(ur, uc);
        fe_zero(acc);
        fe_gemm_rolled<H, MI>(sY, W2Tp, wave, lane, acc);
        FE_TICK(2)
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            FE_JADDR(j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) sX[FE_LOFF(i, r)] = f32_to_bf16(acc[i][0][r]);
            FE_ROT_ACC()
        }
        __syncthreads();
        FE_TICK(7)

        // ---- P5 (gather layout): dP1 = dS1 * silu'(pre1) in place;  d rhohat = dP1 . w_r
        if (el < ne) {
            const float rh = sRh[el];
            const float* wp = w_r + (unsigned)(cth * (CPT * 8));
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                float fa[8], fb[8], g[8];
                unpack8(pfa[q], fa);
                unpack8(pfb[q], fb);
                uint4* sp = (uint4*)&sX[el * H + (((cth * CPT + q) ^ (el & 7)) * 8)];
                unpack8(*sp, g);
                const float4 w0 = *(const float4*)(wp + q * 8), w1 = *(const float4*)(wp + q * 8 + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float z = fa[k] + fb[k] + rh * wv[k], sg = fe_sigmoid(z);
                    g[k] *= sg * (1.0f + z * (1.0f - sg));
                    dot += g[k] * wv[k];
                }
                *sp = pack8(g);
            }
#pragma unroll
            for (int o = 1; o < TPE; o <<= 1) dot += __shfl_xor(dot, o, 64);
            if (cth == 0) (p.drh + e0)[(unsigned)el] = dot;
        }
        ur = nr; uc = nc;
        fe_prefetch(ur, uc);
        __syncthreads();
        FE_TICK(8)
        fe_tile_store<H>(sX, (bf16_t*)p.dP1, e0, ne, tid);
        FE_TICK(1)

        // ---- P6 (one column per thread): receiving-side segment sum of dP1 and the d w_r column sum (deterministic:
        //      complete runs are plain stores, runs that span tiles go through p.bnd and fe_boundary_fix_kernel)
        {
            const bool head_cont = e0 > 0 && p.row[e0 - 1] == sRow[0];
            const bool tail_cont = e0 + ne < p.E && p.row[e0 + ne] == sRow[ne - 1];
            pwr += fe_scan_rows<H, true, BM>(sX, sRow, sRh, ne, head_cont, tail_cont, p.dABrow, (unsigned)p.lddab,
                                             p.bnd + (size_t)tile * 2 * H, tid);
        }
        __syncthreads();                                          // the next tile overwrites sX / sRow
        FE_TICK(9)
    }
    if constexpr (DBG) {
        if (p.dbg && blockIdx.x == 0 && tid == 0)
            for (int k = 0; k < 12; ++k) ((long long*)p.dbg)[k] = tph[k];
    }

    // per-work-group partial column sums: part[blockIdx.x][{b2, bc, w3, w_r}][H]
    float* part = p.part + (size_t)blockIdx.x * 4 * H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = pb2[j], b = pbc[j], c = pw3[j];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
        if (cq == 0) {
            const int colj = wave * 64 + j * 16 + fr;
            part[colj] = a; part[H + colj] = b; part[2 * H + colj] = c;
        }
    }
    part[3 * H + tid] = pwr;
}

#undef FE_COFF
#undef FE_ROT4
#undef FE_ROT_ACC
#undef FE_ROT_D2
#undef FE_JADDR
#undef FE_LOFF
#undef FE_TICK

static int g_fe_bwd_bm = 64;
extern "C" int fabind_gcl_edge_fused_bwd_set_tile(int bm) {
    FB_REQUIRE(bm == 64, "fabind_gcl_edge_fused_bwd_set_tile: 64 edges per tile (the 32-edge build of the two-tile kernel was retired in round 5)");
    g_fe_bwd_bm = bm;
    return 0;
}
extern "C" int fabind_gcl_edge_fused_bwd_tile(void) { return g_fe_bwd_bm; }
static bool g_fe_bwd_variant_set = false;
static int g_fe_bwd_variant = 5;     // 5 = row-wise / operand-swapped in-place tile + store wave (default, fused_edge_bwd3.hip);
                                     // 0 = two LDS tiles, one work-group per CU (round 1; the default for H <= 128).  Variants 1-4 (in-place
                                     // single-tile forms, fused_edge_bwd2.hip) were knob-only and were retired in round 5 (history: 8fbe227)
extern "C" int fabind_gcl_edge_fused_bwd_set_variant(int v) {
    FB_REQUIRE(v == 0 || v == 5,
               "fabind_gcl_edge_fused_bwd_set_variant: 0 (two LDS tiles of 64 edges: the default for H <= 128) or 5 (row-wise, operand-swapped in-place tile + store wave: the recompute default above); variants 1-4 were knobs nobody reached and were retired in round 5");
    g_fe_bwd_variant = v;
    g_fe_bwd_variant_set = true;
    return 0;
}
extern "C" int fabind_gcl_edge_fused_bwd_variant(void) { return g_fe_bwd_variant; }
// The variant a launch at hidden size H takes.  Unless a variant was set explicitly, H <= 128 runs the two-tile kernel (variant 0):
// at H = 128 a 64-edge tile of the store-wave form is two compute waves and thirteen barriers around four 64 x 128 x 128 contractions --
// the pocket model of the full IaBNet (1 layer + out layer, H = 128, 1.5 M edges of the whole proteins): 2,799 us per call with the
// store-wave form, 2,138 with the two-tile form (tools/probes/edge_bwd_h128.py); at H = 512 the order is 7,441 / 8,407 us.
extern "C" int fabind_gcl_edge_fused_bwd_variant_for(int H) {
    return (!g_fe_bwd_variant_set && g_fe_bwd_variant == 5 && H <= 128) ? 0 : g_fe_bwd_variant;
}

extern "C" int fabind_gcl_edge_fused_bwd(const FabindEdgeBwdArgs* a_in, int H, int n_groups, hipStream_t stream) {
    FabindEdgeBwdArgs a_copy = *a_in;
    a_copy.xcd_aware = g_fe_xcd_aware;           // set by the library (callers leave it 0)
    const FabindEdgeBwdArgs* a = &a_copy;
    if (a->E <= 0 || n_groups <= 0) return 0;
    FB_REQUIRE(H == 512 || H == 256 || H == 128 || H == 64, "fabind_gcl_edge_fused_bwd: H must be 64, 128, 256 or 512");
    FB_REQUIRE(a->ldab % 8 == 0, "fabind_gcl_edge_fused_bwd: ldab % 8");
    FB_REQUIRE(a->p_drop >= 0.f && a->p_drop < 1.f, "fabind_gcl_edge_fused_bwd: p_drop in [0, 1)");
    const int g_fe_bwd_variant = fabind_gcl_edge_fused_bwd_variant_for(H);      // (shadows the file-level setting inside this launch)
    if (a->d2f != nullptr || a->z3f != nullptr) {
        FB_REQUIRE(a->d2f != nullptr && a->z3f != nullptr, "fabind_gcl_edge_fused_bwd: d2f and z3f (saved by fabind_gcl_edge_fused_train) come together");
        return fe_bwd4_launch(a, H, n_groups, stream);
    }
    if (g_fe_bwd_variant == 5) return fe_bwd3_launch(a, H, 64, n_groups, stream);
    FB_REQUIRE(a->dAB16 == nullptr, "fabind_gcl_edge_fused_bwd: dAB16 (bf16 receiving-side sums) exists in variant 5 only");
    FB_REQUIRE(a->bnd != nullptr, "fabind_gcl_edge_fused_bwd: bnd (ceil(E / tile) x 2 x H floats of scratch) is required");
    const int BMr = g_fe_bwd_bm;
    const size_t lds = (size_t)2 * BMr * H * 2 + BMr * (sizeof(int) + 2 * sizeof(float));
#define FE_LAUNCH2(HH, BB)                                                                                         \
    do {                                                                                                           \
        static bool set_ = false;                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)gcl_edge_fused_bwd_kernel<HH, BB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((gcl_edge_fused_bwd_kernel<HH, BB, false>), dim3(n_groups), dim3(HH), lds, stream, *a);  \
    } while (0)
#define FE_LAUNCH(HH) FE_LAUNCH2(HH, 64)
    if (a->dbg != nullptr && H == 512) {          // the instrumented build exists for the bench shape only
        static bool setd_ = false;
        if (!setd_) { (void)hipFuncSetAttribute((const void*)gcl_edge_fused_bwd_kernel<512, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
        hipLaunchKernelGGL((gcl_edge_fused_bwd_kernel<512, 64, true>), dim3(n_groups), dim3(512), lds, stream, *a);
    } else
    if (H == 512) FE_LAUNCH(512); else if (H == 256) FE_LAUNCH(256); else if (H == 128) FE_LAUNCH(128); else FE_LAUNCH(64);
    hipLaunchKernelGGL((fe_boundary_fix_kernel<64>), dim3((a->E + 63) / 64), dim3(H < 256 ? H : 256), 0, stream, a->row, a->E, H,
                           a->bnd, a->dABrow, (unsigned)a->lddab);
#undef FE_LAUNCH
#undef FE_LAUNCH2
#undef FE_COFF
#undef FE_ROT4
#undef FE_ROT_ACC
#undef FE_ROT_D2
#undef FE_JADDR
#undef FE_LOFF
#undef FE_TICK
    FB_CHECK_LAUNCH();
    return 0;
}
