// Edge construction and per-edge geometry for the FABind complex graph (gfx950).
//
// Replaces ComplexGraph.construct_edges/_radial_edges (reference models/att_model.py:37-128), which
// builds a dense [N, max_n] fp32 mask and calls nonzero, by direct emission of row-sorted CSR:
// one thread per (aggregating) node walks the candidate nodes of its own complex, whose coordinates
// sit in LDS.  Within a row the emitted order is the reference's (bonds | radial ascending | star |
// global-global), so per-row reductions see the same operand order as the reference's row-major
// nonzero enumeration.  Index type is int32 throughout.
#include "common.h"
#include "fabind_hip.h"

#define EB_THREADS 256
#define EB_LDS_NODES 4096  // complexes up to this many nodes stage their coordinates in LDS (48 KiB)

// distance predicate with the reference's op order: sub -> square-sum -> sqrt -> <=  (att_model.py:124-126).
// Contraction is switched OFF for this function: HIP's __fmul_rn / __fadd_rn are plain `*` / `+` (contractible under
// the default -ffp-contract=fast), and the count pass, the fill pass and the two directions of one edge each inline this
// predicate at a different site -- if the compiler fused a different multiply-add at each, a distance exactly at the
// cut-off would be counted but not filled (an uninitialised CSR slot) or emitted in one direction only.
__device__ __forceinline__ bool within(float ax, float ay, float az, float bx, float by, float bz, float cut) {
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float d2 = (xx + yy) + zz;
    return __fsqrt_rn(d2) <= cut;
}

// The walk of one node over the other nodes of its complex, on coordinates in LDS or (complexes of more than EB_LDS_NODES nodes) in
// global memory.  Two instantiations called from two branches: with ONE pointer selected between the two address spaces the walk
// compiled to flat_load_dwordx3 (30 of them in the count kernel) -- flat accesses to LDS take the vector-memory path, and the count /
// fill passes ran at 536 / 721 us for 1.5e8 candidate pairs; with the address space known they are ds_read_b96 broadcasts.
template <bool FILL>
__device__ __forceinline__ void edges_walk(const float* px, const int b, const int off, const int n, const int C,
                                           const int* bond_row, const int* bond_col, const int* bond_off, const float cut_intra,
                                           const float cut_inter, int* deg_ctx, int* deg_int, const int* rowptr_ctx,
                                           const int* rowptr_int, int* col_ctx, int* row_ctx, int* col_int, int* row_int) {
    const int u = blockIdx.x * EB_THREADS + threadIdx.x;  // local node index
    if (u >= n) return;
    const int gu = off + u;
    const float ux = px[u * 3], uy = px[u * 3 + 1], uz = px[u * 3 + 2];
    int nc = 0, ni = 0;
    int* cc = nullptr; int* rc = nullptr; int* ci = nullptr; int* ri = nullptr;
    if (FILL) {
        int pc = rowptr_ctx[gu], pi = rowptr_int[gu];
        cc = col_ctx + pc; rc = row_ctx + pc; ci = col_int + pi; ri = row_int + pi;
    }
#define EMIT_CTX(v) do { if (FILL) { cc[nc] = (v); rc[nc] = gu; } ++nc; } while (0)
#define EMIT_INT(v) do { if (FILL) { ci[ni] = (v); ri[ni] = gu; } ++ni; } while (0)
    if (u == 0) {                       // glb_c: star to every ligand atom, then glb_p
        for (int v = 1; v < C; ++v) EMIT_CTX(off + v);
        EMIT_CTX(off + C);
    } else if (u < C) {                 // ligand atom: bonds (list order), star to glb_c; inter to protein
        for (int e = bond_off[b]; e < bond_off[b + 1]; ++e)
            if (bond_row[e] == gu) EMIT_CTX(FILL ? bond_col[e] : 0);
        EMIT_CTX(off);
#pragma unroll 8
        for (int v = C + 1; v < n; ++v)
            if (within(ux, uy, uz, px[v * 3], px[v * 3 + 1], px[v * 3 + 2], cut_inter)) EMIT_INT(off + v);
    } else if (u == C) {                // glb_p: glb_c sorts first (smaller index), then every residue
        // reference order inside the row: global_normal run (residues ascending) then global_global (glb_c)
        for (int v = C + 1; v < n; ++v) EMIT_CTX(off + v);
        EMIT_CTX(off);
    } else {                            // protein residue: radial neighbours ascending, then star to glb_p
        for (int v = 1; v < C; ++v)
            if (within(ux, uy, uz, px[v * 3], px[v * 3 + 1], px[v * 3 + 2], cut_inter)) EMIT_INT(off + v);
        // (one thread per node walks every other node of its complex: the walk is a chain of broadcast LDS reads -- unrolled so that
        //  eight candidates' coordinates are in flight per lane; with < 2 waves per SIMD nothing else hides their latency)
#pragma unroll 8
        for (int v = C + 1; v < n; ++v)
            if (v != u && within(ux, uy, uz, px[v * 3], px[v * 3 + 1], px[v * 3 + 2], cut_intra)) EMIT_CTX(off + v);
        EMIT_CTX(off + C);
    }
#undef EMIT_CTX
#undef EMIT_INT
    if (!FILL) { deg_ctx[gu] = nc; deg_int[gu] = ni; }
}

template <bool FILL>
__global__ __launch_bounds__(EB_THREADS) void edges_kernel(const float* __restrict__ x, const int* node_off,
                                                           const int* c_cnt, const int* bond_row, const int* bond_col,
                                                           const int* bond_off, float cut_intra, float cut_inter,
                                                           int* deg_ctx, int* deg_int, const int* rowptr_ctx,
                                                           const int* rowptr_int, int* col_ctx, int* row_ctx,
                                                           int* col_int, int* row_int) {
    __shared__ float sx[EB_LDS_NODES * 3];
    const int b = blockIdx.y;
    const int off = node_off[b], n = node_off[b + 1] - off, C = c_cnt[b];
    if ((int)(blockIdx.x * EB_THREADS) >= n) return;
    const bool use_lds = n <= EB_LDS_NODES;
    if (use_lds) {
        for (int i = threadIdx.x; i < n * 3; i += EB_THREADS) sx[i] = x[(size_t)off * 3 + i];
        __syncthreads();
    }
    if (use_lds) edges_walk<FILL>(sx, b, off, n, C, bond_row, bond_col, bond_off, cut_intra, cut_inter, deg_ctx, deg_int, rowptr_ctx,
                                  rowptr_int, col_ctx, row_ctx, col_int, row_int);
    else edges_walk<FILL>(x + (size_t)off * 3, b, off, n, C, bond_row, bond_col, bond_off, cut_intra, cut_inter, deg_ctx, deg_int,
                          rowptr_ctx, rowptr_int, col_ctx, row_ctx, col_int, row_int);
}

extern "C" int fabind_edges_count(const float* x, const int* node_off, const int* c_cnt, int B, int max_n,
                                  const int* bond_row, const int* bond_off, float cut_intra, float cut_inter,
                                  int* deg_ctx, int* deg_int, hipStream_t stream) {
    if (B <= 0) return 0;
    dim3 grid((max_n + EB_THREADS - 1) / EB_THREADS, B);
    hipLaunchKernelGGL((edges_kernel<false>), grid, dim3(EB_THREADS), 0, stream, x, node_off, c_cnt, bond_row,
                       (const int*)nullptr, bond_off, cut_intra, cut_inter, deg_ctx, deg_int, (const int*)nullptr,
                       (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_edges_fill(const float* x, const int* node_off, const int* c_cnt, int B, int max_n,
                                 const int* bond_row, const int* bond_col, const int* bond_off, float cut_intra,
                                 float cut_inter, const int* rowptr_ctx, const int* rowptr_int, int* col_ctx,
                                 int* row_ctx, int* col_int, int* row_int, hipStream_t stream) {
    if (B <= 0) return 0;
    dim3 grid((max_n + EB_THREADS - 1) / EB_THREADS, B);
    hipLaunchKernelGGL((edges_kernel<true>), grid, dim3(EB_THREADS), 0, stream, x, node_off, c_cnt, bond_row, bond_col,
                       bond_off, cut_intra, cut_inter, (int*)nullptr, (int*)nullptr, rowptr_ctx, rowptr_int, col_ctx,
                       row_ctx, col_int, row_int);
    FB_CHECK_LAUNCH();
    return 0;
}

// ---- single work-group exclusive scan (n ~ 1e5..1e6 node degrees; a few microseconds) ----
__global__ __launch_bounds__(1024) void scan_kernel(const int* in, int* out, int n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { carry_s = 0; out[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        int v = i < n ? in[i] : 0;
        int s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        int wo = 0;
        for (int k = 0; k < w; ++k) wo += wsum[k];
        int carry = carry_s;
        if (i < n) out[i + 1] = carry + wo + s;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wo + s;
        __syncthreads();
    }
}

extern "C" int fabind_exclusive_scan(const int* in, int* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, in, out, n);
    FB_CHECK_LAUNCH();
    return 0;
}

// out[k] = index of the first element of the non-decreasing int32 array `vals` that is >= keys[k] (lower bound; n if none): the
// per-complex [start, end) offsets of a complex-contiguous edge list from node_off (engine.Layout.ranges: replaces ATen's bucketize ->
// index_add histogram -> cumsum -> cast chain, five launches of index glue per list).
__global__ void lower_bound_kernel(const int* __restrict__ vals, int n, const int* __restrict__ keys, int n_keys, int* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    const int key = keys[k];
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (vals[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[k] = lo;
}
extern "C" int fabind_lower_bound(const int* vals, int n, const int* keys, int n_keys, int* out, hipStream_t stream) {
    if (n_keys <= 0) return 0;
    FB_REQUIRE(keys && out && (n == 0 || vals), "fabind_lower_bound: null pointer");
    hipLaunchKernelGGL(lower_bound_kernel, dim3((n_keys + 255) / 256), dim3(256), 0, stream, vals, n, keys, n_keys, out);
    FB_CHECK_LAUNCH();
    return 0;
}

// ---- inter-edge pair bookkeeping ----
__global__ void red_off_kernel(const int* node_off, const int* c_cnt, int B, const int* rowptr_int, int* red_off) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        red_off[0] = 0;
        for (int b = 0; b < B; ++b) {
            acc += rowptr_int[node_off[b] + c_cnt[b]] - rowptr_int[node_off[b]];
            red_off[b + 1] = acc;
        }
    }
}

__global__ void inter_meta_kernel(const int* node_off, const int* c_cnt, int B, const int* rowptr_int,
                                  const int* col_int, const int* row_int, int E, const int* red_off, int* red_idx,
                                  int* red_c, int* red_p, int* mirror) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int r = row_int[e], c = col_int[e];
    // complex of this edge: binary search node_off
    int lo = 0, hi = B;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (node_off[mid] <= r) lo = mid; else hi = mid; }
    const int b = lo, off = node_off[b];
    const int base = rowptr_int[off];
    if (r - off < c_cnt[b]) {  // ligand row: this edge defines the pair
        int k = red_off[b] + (e - base);
        red_idx[e] = k; red_c[k] = r; red_p[k] = c;
    } else {                   // protein row: locate the mirrored edge (row=c ligand, col=r) by binary search
        int s = rowptr_int[c], t = rowptr_int[c + 1];
        while (t - s > 1) { int mid = (s + t) >> 1; if (col_int[mid] <= r) s = mid; else t = mid; }
        red_idx[e] = red_off[b] + (s - base);
        mirror[e] = s;
        mirror[s] = e;
    }
}

extern "C" int fabind_inter_meta(const int* node_off, const int* c_cnt, int B, const int* rowptr_int,
                                 const int* col_int, const int* row_int, int E_int, int* red_off, int* red_idx,
                                 int* red_c, int* red_p, int* mirror, hipStream_t stream) {
    hipLaunchKernelGGL(red_off_kernel, dim3(1), dim3(64), 0, stream, node_off, c_cnt, B, rowptr_int, red_off);
    if (E_int > 0)
        hipLaunchKernelGGL(inter_meta_kernel, dim3((E_int + 255) / 256), dim3(256), 0, stream, node_off, c_cnt, B,
                           rowptr_int, col_int, row_int, E_int, red_off, red_idx, red_c, red_p, mirror);
    FB_CHECK_LAUNCH();
    return 0;
}

// ---- coord2radial, norm_type = per_sample (egnn.py:767-787) ----
__global__ __launch_bounds__(1024) void edge_geom_kernel(const float* __restrict__ x, const int* row, const int* col,
                                                         const int* rowptr, const int* node_off, float* d, float* rho,
                                                         float* rhohat, float* norm) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int e0 = rowptr[node_off[b]], e1 = rowptr[node_off[b + 1]];
    float acc = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += 1024) {
        int r = row[e], c = col[e];
        float dx = x[r * 3] - x[c * 3], dy = x[r * 3 + 1] - x[c * 3 + 1], dz = x[r * 3 + 2] - x[c * 3 + 2];
        float q = dx * dx + dy * dy + dz * dz;
        d[(size_t)e * 3] = dx; d[(size_t)e * 3 + 1] = dy; d[(size_t)e * 3 + 2] = dz;
        rho[e] = q;
        acc += q * q;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += red[k];
    float nrm = sqrtf(tot);
    if (threadIdx.x == 0) norm[b] = nrm;
    for (int e = e0 + threadIdx.x; e < e1; e += 1024) rhohat[e] = rho[e] / nrm;
}

extern "C" int fabind_edge_geom(const float* x, const int* row, const int* col, const int* rowptr, const int* node_off,
                                int B, float* d, float* rho, float* rhohat, float* norm, hipStream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(edge_geom_kernel, dim3(B), dim3(1024), 0, stream, x, row, col, rowptr, node_off, d, rho, rhohat,
                       norm);
    FB_CHECK_LAUNCH();
    return 0;
}
