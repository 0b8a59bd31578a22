// heads.hip -- round 6: the two small fused head kernels of SURVEY 2.3 that rounds 1-5 ran as chains of element-wise torch ops.
//   K17  pocket centre   (reference FABind/fabind/models/model.py:146-158): sigmoid -> clamp -> log -> (Gumbel-)softmax over the two
//        classes -> masked weighted mean of the residue coordinates; one launch forward, one launch for the adjoint.
//   K20  six-term loss   (reference FABind/fabind/main_fabind.py:398-417): SmoothL1(coords), MSE(y_pred, dis_map), MSE(y_by, dis_map),
//        MSE(y_by, y_pred), BCE-with-logits * numel / sum(mask), 0.05 * Huber_delta(centre) -- ONE launch for all six terms and their
//        weighted sum, ONE launch for the five gradient seeds (d coords, d y_pred, d y_by, d logits, d centre).
// Latency-bound vector work (3.84 M pairs at the headline shape, ~100 k at the production shape): plain grid-stride loops, fp32,
// precise expf / logf / division (these feed the 1e-5-relative loss gate), fixed-order reductions (no float atomics: bit-repeatable).
#include "common.h"
#include "fabind_hip.h"

namespace {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_TERMS = 8;        // six terms, the mask count, spare

__device__ __forceinline__ float agent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float cls_target(const void* cls, int cls_dt, long i) {
    switch (cls_dt) {                                  // 0 fp32, 1 int64, 2 int32, 3 uint8 / bool
        case 0: return ((const float*)cls)[i];
        case 1: return (float)((const long long*)cls)[i];
        case 2: return (float)((const int*)cls)[i];
        default: return (float)((const uint8_t*)cls)[i];
    }
}

// Block sums of the six (un-normalised) terms + the mask count -> part[block][8]; the LAST block to arrive (ticket) adds the partials in
// block order and writes out[0..5] = the six weighted terms (order: pocket_cls, pocket_center, contact, contact_by_pred, distill, coord),
// out[6] = their sum, out[7] = sum(mask).  The partials travel through agent-scope loads / stores (the eight XCDs' L2s are not coherent
// for plain accesses inside one kernel).
__global__ __launch_bounds__(LOSS_THREADS) void loss_fwd_kernel(
    const float* __restrict__ coords, const float* __restrict__ coords_true, long n_coord, const float* __restrict__ y_pred,
    const float* __restrict__ y_by, const float* __restrict__ dis_map, long n_pair, const float* __restrict__ logits,
    const void* __restrict__ cls, int cls_dt, const uint8_t* __restrict__ mask, long n_cls, const float* __restrict__ center,
    const float* __restrict__ center_true, long n_center, float w_coord, float w_pair, float w_distill, float w_cls, float w_center,
    float delta, float* part, unsigned* ticket, float* out, float* loss_out, float* terms_out) {
    const long tid = (long)blockIdx.x * LOSS_THREADS + threadIdx.x, nthr = (long)gridDim.x * LOSS_THREADS;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // cls, center, contact, contact_by, distill, coord, mask count
    for (long i = tid; i < n_pair; i += nthr) {
        const float yp = y_pred[i], yb = y_by[i], dm = dis_map[i];
        const float a = yp - dm, b = yb - dm, c = yb - yp;
        acc[2] += a * a; acc[3] += b * b; acc[4] += c * c;
    }
    for (long i = tid; i < n_coord; i += nthr) {               // SmoothL1, beta = 1
        const float d = fabsf(coords[i] - coords_true[i]);
        acc[5] += d < 1.f ? 0.5f * d * d : d - 0.5f;
    }
    for (long i = tid; i < n_cls; i += nthr) {                 // BCE with logits: max(x, 0) - x y + log(1 + exp(-|x|)); padding counts (log 2)
        const float x = logits[i], y = cls_target(cls, cls_dt, i);
        acc[0] += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        acc[6] += mask[i] ? 1.f : 0.f;
    }
    for (long i = tid; i < n_center; i += nthr) {              // Huber
        const float d = fabsf(center[i] - center_true[i]);
        acc[1] += d <= delta ? 0.5f * d * d : delta * (d - 0.5f * delta);
    }
    __shared__ float sh[LOSS_THREADS / 64][7];
    __shared__ unsigned sh_ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float v = 0.f;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) v += sh[w][threadIdx.x];
        agent_store(part + (size_t)blockIdx.x * LOSS_TERMS + threadIdx.x, v);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) sh_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    if (sh_ticket != gridDim.x - 1) return;
    __threadfence();
    if (wave == 0) {
        float tot[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            float v = 0.f;
            for (unsigned b = lane; b < gridDim.x; b += 64) v += agent_load(part + (size_t)b * LOSS_TERMS + k);
            tot[k] = wave_sum(v);
        }
        if (lane == 0) {
            const float t_cls = n_cls > 0 ? w_cls * tot[0] / tot[6] : 0.f;          // mean over numel x numel / sum(mask)
            const float t_cen = n_center > 0 ? w_center * tot[1] / (float)n_center : 0.f;
            const float t_con = n_pair > 0 ? w_pair * tot[2] / (float)n_pair : 0.f;
            const float t_cby = n_pair > 0 ? w_pair * tot[3] / (float)n_pair : 0.f;
            const float t_dis = n_pair > 0 ? w_distill * tot[4] / (float)n_pair : 0.f;
            const float t_crd = n_coord > 0 ? w_coord * tot[5] / (float)n_coord : 0.f;
            out[0] = t_cls; out[1] = t_cen; out[2] = t_con; out[3] = t_cby; out[4] = t_dis; out[5] = t_crd;
            out[6] = ((((t_cls + t_cen) + t_con) + t_cby) + t_dis) + t_crd;          // the order of python's sum(terms.values())
            out[7] = tot[6];
            if (loss_out) loss_out[0] = out[6];                                      // the same values as separate tensors (autograd outputs)
            if (terms_out) { terms_out[0] = t_cls; terms_out[1] = t_cen; terms_out[2] = t_con; terms_out[3] = t_cby; terms_out[4] = t_dis; terms_out[5] = t_crd; }
            *ticket = 0u;                                                            // re-armed for the next launch on this stream
        }
    }
}

// Gradient seeds of the six terms: g[k] = g_loss + g_terms[k] is the upstream gradient of term k (either pointer may be NULL = 0).
__global__ __launch_bounds__(LOSS_THREADS) void loss_bwd_kernel(
    const float* __restrict__ coords, const float* __restrict__ coords_true, long n_coord, const float* __restrict__ y_pred,
    const float* __restrict__ y_by, const float* __restrict__ dis_map, long n_pair, const float* __restrict__ logits,
    const void* __restrict__ cls, int cls_dt, long n_cls, const float* __restrict__ center, const float* __restrict__ center_true,
    long n_center, float w_coord, float w_pair, float w_distill, float w_cls, float w_center, float delta,
    const float* __restrict__ fwd_out, const float* __restrict__ g_loss, const float* __restrict__ g_terms, float* d_coords,
    float* d_y_pred, float* d_y_by, float* d_logits, float* d_center) {
    const long tid = (long)blockIdx.x * LOSS_THREADS + threadIdx.x, nthr = (long)gridDim.x * LOSS_THREADS;
    const float gl = g_loss ? g_loss[0] : 0.f;
    float g[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = gl + (g_terms ? g_terms[k] : 0.f);
    const float s_pair = n_pair > 0 ? 2.f * w_pair / (float)n_pair : 0.f, s_dis = n_pair > 0 ? 2.f * w_distill / (float)n_pair : 0.f;
    for (long i = tid; i < n_pair; i += nthr) {
        const float yp = y_pred[i], yb = y_by[i], dm = dis_map[i];
        if (d_y_pred) d_y_pred[i] = g[2] * s_pair * (yp - dm) - g[4] * s_dis * (yb - yp);
        if (d_y_by) d_y_by[i] = g[3] * s_pair * (yb - dm) + g[4] * s_dis * (yb - yp);
    }
    if (d_coords) {
        const float s = n_coord > 0 ? g[5] * w_coord / (float)n_coord : 0.f;
        for (long i = tid; i < n_coord; i += nthr) {
            const float d = coords[i] - coords_true[i];
            d_coords[i] = s * (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f));
        }
    }
    if (d_logits) {
        const float s = n_cls > 0 ? g[0] * w_cls / fwd_out[7] : 0.f;
        for (long i = tid; i < n_cls; i += nthr) {
            const float x = logits[i];
            d_logits[i] = s * (1.f / (1.f + expf(-x)) - cls_target(cls, cls_dt, i));
        }
    }
    if (d_center) {
        const float s = n_center > 0 ? g[1] * w_center / (float)n_center : 0.f;
        for (long i = tid; i < n_center; i += nthr) {
            const float d = center[i] - center_true[i];
            d_center[i] = s * (fabsf(d) <= delta ? d : (d > 0.f ? delta : -delta));
        }
    }
}

// ---- K17: pocket centre ---------------------------------------------------------------------------------------------------------
// y1 of one residue: softmax over (log p0 + n0, log p1 + n1) / tau, class 1; p = clamp([1 - s, s], 1e-6, 1 - 1e-6), s = sigmoid(logit)
__device__ __forceinline__ float pc_soft(float logit, const float* noise, long idx, float inv_tau, float* dsoft_dlogit) {
    const float s = 1.f / (1.f + expf(-logit));
    const float q0 = 1.f - s, q1 = s;
    const float p0 = fminf(fmaxf(q0, 1e-6f), 1.f - 1e-6f), p1 = fminf(fmaxf(q1, 1e-6f), 1.f - 1e-6f);
    float a0 = logf(p0), a1 = logf(p1);
    if (noise) { a0 += noise[idx * 2]; a1 += noise[idx * 2 + 1]; }
    a0 *= inv_tau; a1 *= inv_tau;
    const float m = fmaxf(a0, a1), e0 = expf(a0 - m), e1 = expf(a1 - m);
    const float y1 = e1 / (e0 + e1);
    if (dsoft_dlogit) {
        // d y1 / d a1 = y1 (1 - y1) inv_tau = - d y1 / d a0; d a1 / d s = [q1 inside the clamp] / p1; d a0 / d s = -[q0 inside] / p0; d s / d logit = s (1 - s)
        const float in1 = (q1 >= 1e-6f && q1 <= 1.f - 1e-6f) ? 1.f / p1 : 0.f, in0 = (q0 >= 1e-6f && q0 <= 1.f - 1e-6f) ? 1.f / p0 : 0.f;
        *dsoft_dlogit = y1 * (1.f - y1) * inv_tau * (in1 + in0) * s * (1.f - s);
    }
    return y1;
}

// one work-group per complex: centre[b] = sum_l w_l xyz[b,l] / sum_l w_l, w_l = y1_l (hard: the straight-through one-hot) * mask_l
__global__ __launch_bounds__(256) void pocket_center_fwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                                const float* __restrict__ xyz, const float* __restrict__ noise,
                                                                int L, float inv_tau, int hard, float* center, float* wsum) {
    const int b = blockIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = threadIdx.x; l < L; l += 256) {
        const long i = (long)b * L + l;
        if (!mask[i]) continue;
        float y1 = pc_soft(logits[i], noise, i, inv_tau, nullptr);
        if (hard) y1 = ((y1 > 0.5f ? 1.f : 0.f) - y1) + y1;        // y_hard - y_soft.detach() + y_soft; argmax tie -> class 0 (torch.max: first index)
        a[0] += y1; a[1] += y1 * xyz[i * 3]; a[2] += y1 * xyz[i * 3 + 1]; a[3] += y1 * xyz[i * 3 + 2];
    }
    __shared__ float sh[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = wave_sum(a[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
        center[b * 3] = t[1] / t[0]; center[b * 3 + 1] = t[2] / t[0]; center[b * 3 + 2] = t[3] / t[0];
        wsum[b] = t[0];
    }
}

// d logits[b,l] = mask * (d centre . (xyz_l - centre)) / wsum * d y1_soft / d logit      (hard: the straight-through estimator's gradient)
__global__ __launch_bounds__(256) void pocket_center_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                                const float* __restrict__ xyz, const float* __restrict__ noise,
                                                                int L, float inv_tau, const float* __restrict__ center,
                                                                const float* __restrict__ wsum, const float* __restrict__ dcenter,
                                                                long n, float* dlogits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (!mask[i]) { dlogits[i] = 0.f; return; }
    const int b = (int)(i / L);
    float ds;
    (void)pc_soft(logits[i], noise, i, inv_tau, &ds);
    const float dw = (dcenter[b * 3] * (xyz[i * 3] - center[b * 3]) + dcenter[b * 3 + 1] * (xyz[i * 3 + 1] - center[b * 3 + 1]) +
                      dcenter[b * 3 + 2] * (xyz[i * 3 + 2] - center[b * 3 + 2])) / wsum[b];
    dlogits[i] = dw * ds;
}


// ---- K19 glue: the distance-map head's pair operations on per-complex BLOCK descriptors ------------------------------------------------
// The valid (pocket residue, ligand atom) pairs of complex b are the dense block [P_b x C_b], protein-major, at pairs pair_off[b] + i * C_b + j
// (reference model.py:352-361: z[z_mask] of the padded [B, P, C] tensors).  desc[b] = {pair_off, P_b, C_b, p_row0, c_row0, tile0}: rows of the
// pocket / ligand operands and the first of the complex's ceil(P_b / TP) protein tiles.  Rounds 1-5 described the same pairs by two index lists
// and, for the adjoints, built a CSR per call (two stable sorts of 3.84 M keys at the headline shape) and walked it one wave per row.
constexpr int PB_TP = 128;             // proteins per tile
constexpr int PB_CH = 20;              // ligand atoms per register chunk of the Hadamard adjoint (20 tc pairs + 20 accumulator pairs + 20 loads per thread)

struct PairBlock { long pair_off; int P, C, p_row0, c_row0, tile0, pad; };

__device__ __forceinline__ int pb_find(const PairBlock* __restrict__ desc, int B, int tile) {   // the complex that owns `tile` (tile0 is increasing)
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// y[pair] = clamp(scale * |xp[p_row0 + i] - xc[c_row0 + j]|, lo, hi)                      (torch.cdist -> mask -> unnormalize -> clamp)
__global__ __launch_bounds__(256) void pair_dist_fwd_kernel(const PairBlock* __restrict__ desc, int B, const float* __restrict__ xp,
                                                            const float* __restrict__ xc, float scale, float lo, float hi, float* y) {
    const int b = pb_find(desc, B, blockIdx.x);
    const PairBlock d = desc[b];
    const int i0 = (blockIdx.x - d.tile0) * PB_TP, i1 = min(d.P, i0 + PB_TP);
    if (i0 >= d.P) return;
    const long n = (long)(i1 - i0) * d.C;
    for (long k = threadIdx.x; k < n; k += 256) {
        const int i = i0 + (int)(k / d.C), j = (int)(k % d.C);
        const float* a = xp + (size_t)(d.p_row0 + i) * 3;
        const float* c = xc + (size_t)(d.c_row0 + j) * 3;
        const float dx = a[0] - c[0], dy = a[1] - c[1], dz = a[2] - c[2];
        const float v = scale * sqrtf(dx * dx + dy * dy + dz * dz);
        y[d.pair_off + (long)i0 * d.C + k] = fminf(fmaxf(v, lo), hi);
    }
}

// d xc[c_row0 + j] = sum_i dy[pair(i, j)] * scale * (xc_j - xp_i) / |xc_j - xp_i|  where the clamp is inactive (and the distance non-zero).
// One work-group per (complex, one of 8 protein strides): thread (il, j) walks its proteins for ONE atom -- consecutive threads read consecutive
// pairs --, the il lanes meet through LDS in lane order -> part[b][s][j][3]; pair_dist_bwd_reduce adds the 8 strides in order.
__global__ __launch_bounds__(256) void pair_dist_bwd_kernel(const PairBlock* __restrict__ desc, const float* __restrict__ xp,
                                                            const float* __restrict__ xc, const float* __restrict__ dy, float scale,
                                                            float lo, float hi, int max_C, float* part) {
    const int b = blockIdx.x, s = blockIdx.y;
    const PairBlock d = desc[b];
    __shared__ float sh[256][3];
    for (int j0 = 0; j0 < d.C; j0 += 256) {
        const int cw = min(256, d.C - j0);                 // atoms of this chunk; lanes = 256 / cw proteins in flight per atom
        const int nl = 256 / cw, il = threadIdx.x / cw, j = j0 + threadIdx.x % cw;
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (il < nl) {
            const float* c = xc + (size_t)(d.c_row0 + j) * 3;
            const float cx = c[0], cy = c[1], cz = c[2];
            for (int i = s + 8 * il; i < d.P; i += 8 * nl) {
                const float* a = xp + (size_t)(d.p_row0 + i) * 3;
                const float dx = cx - a[0], dyv = cy - a[1], dz = cz - a[2];
                const float r = sqrtf(dx * dx + dyv * dyv + dz * dz), v = scale * r;
                const float g = (r > 0.f && v >= lo && v <= hi) ? dy[d.pair_off + (long)i * d.C + j] * scale / r : 0.f;
                ax += g * dx; ay += g * dyv; az += g * dz;
            }
        }
        sh[threadIdx.x][0] = ax; sh[threadIdx.x][1] = ay; sh[threadIdx.x][2] = az;
        __syncthreads();
        if (threadIdx.x < cw) {
            float tx = 0.f, ty = 0.f, tz = 0.f;
            for (int l = 0; l < nl; ++l) { tx += sh[l * cw + threadIdx.x][0]; ty += sh[l * cw + threadIdx.x][1]; tz += sh[l * cw + threadIdx.x][2]; }
            float* o = part + (((size_t)b * 8 + s) * max_C + j0 + threadIdx.x) * 3;
            o[0] = tx; o[1] = ty; o[2] = tz;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pair_dist_bwd_reduce_kernel(const PairBlock* __restrict__ desc, int B, int max_C,
                                                                   const float* __restrict__ part, float* dxc) {
    const int b = blockIdx.x;
    const PairBlock d = desc[b];
    for (int k = threadIdx.x; k < d.C * 3; k += 256) {
        float v = 0.f;
        for (int s = 0; s < 8; ++s) v += part[(((size_t)b * 8 + s) * max_C) * 3 + k];
        dxc[(size_t)d.c_row0 * 3 + k] = v;
    }
}

// Adjoint of out[pair(i, j), :] = tp[p_row0 + i, :] * tc[c_row0 + j, :] (einsum 'bik,bjk->bijk' on the valid pairs, model.py:355):
//   d tp[i] = sum_j dout[pair] * tc[j]  (complete inside the tile: written here),   d tc[j] = sum_i dout[pair] * tp[i]  (the tile's share -> part).
// A thread owns TWO columns and, for the current chunk of <= 20 atoms, the 20 x 2 accumulators of d tc in registers; W / 2 threads span a row
// (one 1 KiB row of dout per pair at W = 512, read once, non-temporal), 256 / (W / 2) proteins run side by side and meet through LDS in lane order.
template <int W>
__global__ __launch_bounds__(256) void block_hadamard_bwd_kernel(const PairBlock* __restrict__ desc, int B, const bf16_t* __restrict__ dout,
                                                                 int ldo, const float* __restrict__ tp, int ldtp,
                                                                 const float* __restrict__ tc, int ldtc, float* dtp, int lddp,
                                                                 int nchunk_max, float* part) {
    constexpr int CT = W / 2, NL = 256 / CT;              // column threads per row, protein lanes
    const int b = pb_find(desc, B, blockIdx.x);
    const PairBlock d = desc[b];
    const int i0 = (blockIdx.x - d.tile0) * PB_TP, i1 = min(d.P, i0 + PB_TP);
    if (i0 >= d.P) return;
    // (NL == 1, W = 512: every thread of the work-group walks the SAME protein -- row bases are wave-uniform: scalar registers + one lane offset)
    const int ct = (NL == 1) ? (int)threadIdx.x : (int)threadIdx.x % CT, il = (NL == 1) ? 0 : (int)threadIdx.x / CT, c2 = ct * 2;
    extern __shared__ float sm[];                          // [NL][PB_CH][W] for the lane reduction (NL > 1 only)
    for (int j0 = 0, ch = 0; j0 < d.C; j0 += PB_CH, ++ch) {
        const int cw = min(PB_CH, d.C - j0);
        // this thread's two columns of the chunk's tc rows and of its d tc accumulators: registers (rows past the chunk: zeros)
        float2 c[PB_CH], acc[PB_CH];
#pragma unroll
        for (int j = 0; j < PB_CH; ++j) {
            c[j] = j < cw ? *(const float2*)(tc + (size_t)(d.c_row0 + j0 + j) * ldtc + c2) : make_float2(0.f, 0.f);
            acc[j] = make_float2(0.f, 0.f);
        }
        for (int i = i0 + il; i < i1; i += NL) {
            const float2 p = *(const float2*)(tp + (size_t)(d.p_row0 + i) * ldtp + c2);
            const bf16_t* row = dout + (size_t)(d.pair_off + (long)i * d.C + j0) * ldo;      // + c2 per lane below
            // ALL loads of the protein's chunk rows first (read once: non-temporal), then the arithmetic: 20 x 256 B per wave in flight
            uint32_t u[PB_CH];
            if (cw == PB_CH) {
#pragma unroll
                for (int j = 0; j < PB_CH; ++j) u[j] = __builtin_nontemporal_load((const uint32_t*)(row + (size_t)j * ldo) + ct);
            } else {
#pragma unroll
                for (int j = 0; j < PB_CH; ++j) u[j] = j < cw ? __builtin_nontemporal_load((const uint32_t*)(row + (size_t)j * ldo) + ct) : 0u;
            }
            float ax = 0.f, ay = 0.f;
#pragma unroll
            for (int j = 0; j < PB_CH; ++j) {
                const float dx = __uint_as_float(u[j] << 16), dy = __uint_as_float(u[j] & 0xffff0000u);
                ax += dx * c[j].x; ay += dy * c[j].y;
                acc[j].x += dx * p.x; acc[j].y += dy * p.y;
            }
            float* o = dtp + (size_t)(d.p_row0 + i) * lddp + c2;
            if (ch == 0) *(float2*)o = make_float2(ax, ay);                       // (this work-group owns the row)
            else { const float2 q = *(const float2*)o; *(float2*)o = make_float2(q.x + ax, q.y + ay); }
        }
        float* po = part + ((size_t)blockIdx.x * nchunk_max + ch) * PB_CH * W;
        if (NL == 1) {
#pragma unroll
            for (int j = 0; j < PB_CH; ++j)
                if (j < cw) *(float2*)(po + j * W + c2) = acc[j];
        } else {
            __syncthreads();                               // (the previous chunk's readers are done with the buffer)
#pragma unroll
            for (int j = 0; j < PB_CH; ++j)
                if (j < cw) *(float2*)(sm + ((size_t)il * PB_CH + j) * W + c2) = acc[j];
            __syncthreads();
            for (int k = threadIdx.x; k < cw * W; k += 256) {
                float v = 0.f;
                for (int l = 0; l < NL; ++l) v += sm[(size_t)l * PB_CH * W + k];
                po[k] = v;
            }
        }
    }
}
// Forward of the same product on the block descriptors: out[pair(i, j), :] = bf16(tp[p_row0 + i, :] * tc[c_row0 + j, :]).  A thread owns 8
// columns (one 16-byte bf16 store per pair), W / 8 threads span a row, 256 / (W / 8) atoms j run side by side; the protein row stays in
// registers across its C_b pairs, the complex's ligand rows come from L1 / L2 (40 x 2 KiB at the headline shape).
template <int W>
__global__ __launch_bounds__(256) void block_hadamard_fwd_kernel(const PairBlock* __restrict__ desc, int B, const float* __restrict__ tp,
                                                                 int ldtp, const float* __restrict__ tc, int ldtc, bf16_t* out, int ldo) {
    constexpr int CT = W / 8, RP = 256 / CT;               // column threads per row, rows (atoms) in parallel
    const int b = pb_find(desc, B, blockIdx.x);
    const PairBlock d = desc[b];
    const int i0 = (blockIdx.x - d.tile0) * PB_TP, i1 = min(d.P, i0 + PB_TP);
    if (i0 >= d.P) return;
    const int ct = threadIdx.x % CT, rg = threadIdx.x / CT, c8 = ct * 8;
    for (int i = i0; i < i1; ++i) {
        const float4 p0 = *(const float4*)(tp + (size_t)(d.p_row0 + i) * ldtp + c8), p1 = *(const float4*)(tp + (size_t)(d.p_row0 + i) * ldtp + c8 + 4);
        bf16_t* orow = out + (size_t)(d.pair_off + (long)i * d.C) * ldo + c8;
        for (int j = rg; j < d.C; j += RP) {
            const float* c = tc + (size_t)(d.c_row0 + j) * ldtc + c8;
            const float4 q0 = *(const float4*)c, q1 = *(const float4*)(c + 4);
            typedef __attribute__((ext_vector_type(4))) unsigned int nc_u4;
            const nc_u4 v = {pack2_bf16(p0.x * q0.x, p0.y * q0.y), pack2_bf16(p0.z * q0.z, p0.w * q0.w),
                             pack2_bf16(p1.x * q1.x, p1.y * q1.y), pack2_bf16(p1.z * q1.z, p1.w * q1.w)};
            __builtin_nontemporal_store(v, (nc_u4*)(orow + (size_t)j * ldo));
        }
    }
}

// d tc[c_row0 + j, :] = sum over the complex's tiles (in tile order) of part[tile][chunk of j][j % 20][:]
__global__ __launch_bounds__(256) void block_hadamard_bwd_reduce_kernel(const PairBlock* __restrict__ desc, int B, const int* __restrict__ row_b,
                                                                        int W, int nchunk_max, const float* __restrict__ part, float* dtc,
                                                                        int lddc) {
    const int r = blockIdx.x, b = row_b[r];                // r: ligand row (global index into the ligand operand), b its complex
    const PairBlock d = desc[b];
    const int j = r - d.c_row0, ch = j / PB_CH, jj = j % PB_CH;
    const int nt = (d.P + PB_TP - 1) / PB_TP;
    for (int c = threadIdx.x; c < W; c += 256) {
        float v = 0.f;
        for (int t = 0; t < nt; ++t) v += part[(((size_t)(d.tile0 + t) * nchunk_max + ch) * PB_CH + jj) * W + c];
        dtc[(size_t)r * lddc + c] = v;
    }
}

}  // namespace

extern "C" int fabind_pair_block_tile(void) { return PB_TP; }
extern "C" int fabind_pair_block_chunk(void) { return PB_CH; }

extern "C" int fabind_pair_dist_fwd(const void* desc, int B, int n_tiles, const float* xp, const float* xc, float scale, float lo, float hi,
                                    float* y, hipStream_t stream) {
    if (B <= 0 || n_tiles <= 0) return 0;
    hipLaunchKernelGGL(pair_dist_fwd_kernel, dim3(n_tiles), dim3(256), 0, stream, (const PairBlock*)desc, B, xp, xc, scale, lo, hi, y);
    FB_CHECK_LAUNCH();
    return 0;
}
extern "C" int fabind_pair_dist_bwd(const void* desc, int B, int max_C, const float* xp, const float* xc, const float* dy, float scale,
                                    float lo, float hi, float* part, float* dxc, hipStream_t stream) {
    if (B <= 0) return 0;
    FB_REQUIRE(part && dxc && max_C > 0, "fabind_pair_dist_bwd: scratch [B][8][max_C][3] and the output are required");
    hipLaunchKernelGGL(pair_dist_bwd_kernel, dim3(B, 8), dim3(256), 0, stream, (const PairBlock*)desc, xp, xc, dy, scale, lo, hi, max_C, part);
    FB_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_dist_bwd_reduce_kernel, dim3(B), dim3(256), 0, stream, (const PairBlock*)desc, B, max_C, part, dxc);
    FB_CHECK_LAUNCH();
    return 0;
}
extern "C" int fabind_block_hadamard_fwd(const void* desc, int B, int n_tiles, const float* tp, int ldtp, const float* tc, int ldtc, int W,
                                         void* out, int ldo, hipStream_t stream) {
    FB_REQUIRE(W == 64 || W == 128 || W == 256 || W == 512, "fabind_block_hadamard_fwd: W in {64, 128, 256, 512}");
    FB_REQUIRE(ldtp % 4 == 0 && ldtc % 4 == 0 && ldo % 8 == 0 && ((((uintptr_t)tp | (uintptr_t)tc | (uintptr_t)out) & 15) == 0),
               "fabind_block_hadamard_fwd: 16-byte aligned rows");
    if (B <= 0 || n_tiles <= 0) return 0;
    const PairBlock* dd = (const PairBlock*)desc;
    bf16_t* o = (bf16_t*)out;
    if (W == 512) hipLaunchKernelGGL((block_hadamard_fwd_kernel<512>), dim3(n_tiles), dim3(256), 0, stream, dd, B, tp, ldtp, tc, ldtc, o, ldo);
    else if (W == 256) hipLaunchKernelGGL((block_hadamard_fwd_kernel<256>), dim3(n_tiles), dim3(256), 0, stream, dd, B, tp, ldtp, tc, ldtc, o, ldo);
    else if (W == 128) hipLaunchKernelGGL((block_hadamard_fwd_kernel<128>), dim3(n_tiles), dim3(256), 0, stream, dd, B, tp, ldtp, tc, ldtc, o, ldo);
    else hipLaunchKernelGGL((block_hadamard_fwd_kernel<64>), dim3(n_tiles), dim3(256), 0, stream, dd, B, tp, ldtp, tc, ldtc, o, ldo);
    FB_CHECK_LAUNCH();
    return 0;
}
extern "C" int fabind_block_hadamard_bwd(const void* desc, int B, int n_tiles, const void* dout, int ldo, const float* tp, int ldtp,
                                         const float* tc, int ldtc, int W, const int* row_b, int n_crows, int nchunk_max, float* part,
                                         float* dtp, int lddp, float* dtc, int lddc, hipStream_t stream) {
    FB_REQUIRE(W == 64 || W == 128 || W == 256 || W == 512, "fabind_block_hadamard_bwd: W in {64, 128, 256, 512}");
    FB_REQUIRE(ldo % 2 == 0 && ldtp % 2 == 0 && lddp % 2 == 0 && ((((uintptr_t)dout) & 3) == 0) && ((((uintptr_t)tp | (uintptr_t)dtp | (uintptr_t)part) & 7) == 0),
               "fabind_block_hadamard_bwd: alignment (bf16 pairs / float2)");
    if (B <= 0 || n_tiles <= 0) return 0;
#define BH_LAUNCH(WW)                                                                                                              \
    do {                                                                                                                           \
        constexpr int NL_ = 256 / (WW / 2);                                                                                        \
        const size_t lds = NL_ > 1 ? (size_t)PB_CH * WW * 4 * NL_ : 0;                                                            \
        static bool set_ = false;                                                                                                  \
        if (!set_) { (void)hipFuncSetAttribute((const void*)block_hadamard_bwd_kernel<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_ = true; } \
        hipLaunchKernelGGL((block_hadamard_bwd_kernel<WW>), dim3(n_tiles), dim3(256), lds, stream, (const PairBlock*)desc, B,      \
                           (const bf16_t*)dout, ldo, tp, ldtp, tc, ldtc, dtp, lddp, nchunk_max, part);                             \
    } while (0)
    if (W == 512) BH_LAUNCH(512); else if (W == 256) BH_LAUNCH(256); else if (W == 128) BH_LAUNCH(128); else BH_LAUNCH(64);
#undef BH_LAUNCH
    FB_CHECK_LAUNCH();
    if (n_crows > 0) {
        hipLaunchKernelGGL(block_hadamard_bwd_reduce_kernel, dim3(n_crows), dim3(256), 0, stream, (const PairBlock*)desc, B, row_b, W,
                           nchunk_max, part, dtc, lddc);
        FB_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int fabind_loss_blocks(long n_pair, long n_coord, long n_cls) {
    const long n = n_pair > n_coord ? (n_pair > n_cls ? n_pair : n_cls) : (n_coord > n_cls ? n_coord : n_cls);
    long b = (n + LOSS_THREADS * 8 - 1) / (LOSS_THREADS * 8);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int fabind_loss_fwd(const float* coords, const float* coords_true, long n_coord, const float* y_pred, const float* y_by,
                               const float* dis_map, long n_pair, const float* logits, const void* cls, int cls_dt,
                               const uint8_t* mask, long n_cls, const float* center, const float* center_true, long n_center,
                               float w_coord, float w_pair, float w_distill, float w_cls, float w_center, float delta, float* part,
                               unsigned* ticket, float* out, float* loss_out, float* terms_out, hipStream_t stream) {
    FB_REQUIRE(part && ticket && out, "fabind_loss_fwd: null scratch / output");
    FB_REQUIRE(cls_dt >= 0 && cls_dt <= 3, "fabind_loss_fwd: cls_dt is 0 (fp32), 1 (int64), 2 (int32) or 3 (uint8 / bool)");
    FB_REQUIRE(n_coord >= 0 && n_pair >= 0 && n_cls >= 0 && n_center >= 0, "fabind_loss_fwd: negative size");
    const int nblk = fabind_loss_blocks(n_pair, n_coord, n_cls);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(nblk), dim3(LOSS_THREADS), 0, stream, coords, coords_true, n_coord, y_pred, y_by, dis_map,
                       n_pair, logits, cls, cls_dt, mask, n_cls, center, center_true, n_center, w_coord, w_pair, w_distill, w_cls,
                       w_center, delta, part, ticket, out, loss_out, terms_out);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_loss_bwd(const float* coords, const float* coords_true, long n_coord, const float* y_pred, const float* y_by,
                               const float* dis_map, long n_pair, const float* logits, const void* cls, int cls_dt, long n_cls,
                               const float* center, const float* center_true, long n_center, float w_coord, float w_pair,
                               float w_distill, float w_cls, float w_center, float delta, const float* fwd_out, const float* g_loss,
                               const float* g_terms, float* d_coords, float* d_y_pred, float* d_y_by, float* d_logits,
                               float* d_center, hipStream_t stream) {
    FB_REQUIRE(fwd_out && (g_loss || g_terms), "fabind_loss_bwd: needs the forward's output vector and an upstream gradient");
    FB_REQUIRE(cls_dt >= 0 && cls_dt <= 3, "fabind_loss_bwd: cls_dt");
    const int nblk = fabind_loss_blocks(n_pair, n_coord, n_cls);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nblk), dim3(LOSS_THREADS), 0, stream, coords, coords_true, n_coord, y_pred, y_by, dis_map,
                       n_pair, logits, cls, cls_dt, n_cls, center, center_true, n_center, w_coord, w_pair, w_distill, w_cls, w_center,
                       delta, fwd_out, g_loss, g_terms, d_coords, d_y_pred, d_y_by, d_logits, d_center);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_pocket_center_fwd(const float* logits, const uint8_t* mask, const float* xyz, const float* noise, int B, int L,
                                        float tau, int hard, float* center, float* wsum, hipStream_t stream) {
    FB_REQUIRE(tau > 0.f, "fabind_pocket_center_fwd: tau > 0");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pocket_center_fwd_kernel, dim3(B), dim3(256), 0, stream, logits, mask, xyz, noise, L, 1.f / tau, hard, center, wsum);
    FB_CHECK_LAUNCH();
    return 0;
}

extern "C" int fabind_pocket_center_bwd(const float* logits, const uint8_t* mask, const float* xyz, const float* noise, int B, int L,
                                        float tau, const float* center, const float* wsum, const float* dcenter, float* dlogits,
                                        hipStream_t stream) {
    FB_REQUIRE(tau > 0.f, "fabind_pocket_center_bwd: tau > 0");
    const long n = (long)B * L;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pocket_center_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, logits, mask, xyz, noise, L,
                       1.f / tau, center, wsum, dcenter, n, dlogits);
    FB_CHECK_LAUNCH();
    return 0;
}
