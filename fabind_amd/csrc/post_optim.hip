// Ligand post-optimisation (reference FABind/fabind/utils/post_optim_utils.py:9-64; run serially per complex on the CPU
// by fabind_inference.py:285-328): `steps` Adam iterations (lr 0.1, betas 0.9/0.999, eps 1e-8) on the predicted ligand
// coordinates x against
//
//   loss = sum_{(i,j) in LAS} | |x_i - x_j| - |r_i - r_j| |  +  2 sum_{i,j} relu(1.22 - |x_i - x_j|)
//
// (r = RDKit conformer; LAS = dense adjacency of the LAS edges; without LAS edges the first term runs over all ordered
// pairs and the excluded-volume term is dropped).  One work-group per ligand runs ALL iterations in one launch: the
// coordinates live in LDS, each thread owns up to two atoms (positions, both Adam moments and the reference row in
// registers) and evaluates its own gradient row -- |x_k - x_j| against every j for the excluded-volume term, against
// its LAS neighbours (CSR of the symmetrised, de-duplicated edge set) for the distance term.  A batch of ligands is one
// launch; the only HBM traffic is the initial read and the final write.  Gradient conventions follow torch: d|.|/dx = 0
// at 0, relu'(0) = 0, cdist' = 0 at distance 0.
#include "common.h"
#include "fabind_hip.h"

#define PO_THREADS 256
#define PO_APT 2                                  // atoms per thread -> ligands of up to 512 atoms

__device__ __forceinline__ float po_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < PO_THREADS / 64; ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(PO_THREADS) void post_optim_kernel(const float* __restrict__ x0, const float* __restrict__ ref,
                                                                const int* __restrict__ atom_off,
                                                                const int* __restrict__ nbr_ptr, const int* __restrict__ nbr_idx,
                                                                int all_pairs, int steps, float lr, float* __restrict__ x_out,
                                                                float* __restrict__ loss_out, float* __restrict__ rmsd_out) {
    extern __shared__ float sm[];
    const int lig = blockIdx.x, tid = threadIdx.x;
    const int a0 = atom_off[lig], n = atom_off[lig + 1] - a0;
    float* sx = sm;                               // [n][3] current coordinates
    float* sr = sm + 3 * n;                       // [n][3] reference conformer
    float* red = sr + 3 * n;                      // [4]
    for (int i = tid; i < 3 * n; i += PO_THREADS) { sx[i] = x0[(size_t)a0 * 3 + i]; sr[i] = ref[(size_t)a0 * 3 + i]; }
    __syncthreads();
    float x[PO_APT][3], r[PO_APT][3], m[PO_APT][3], v[PO_APT][3];
    int e0[PO_APT], e1[PO_APT];
#pragma unroll
    for (int a = 0; a < PO_APT; ++a) {
        const int k = tid + a * PO_THREADS;
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[a][c] = k < n ? sx[3 * k + c] : 0.f; r[a][c] = k < n ? sr[3 * k + c] : 0.f; m[a][c] = 0.f; v[a][c] = 0.f; }
        e0[a] = (k < n && !all_pairs) ? nbr_ptr[a0 + k] : 0;
        e1[a] = (k < n && !all_pairs) ? nbr_ptr[a0 + k + 1] : 0;
    }
    float b1t = 1.f, b2t = 1.f;                   // beta^t
    for (int t = 1; t <= steps; ++t) {
        float g[PO_APT][3], lsum = 0.f;
#pragma unroll
        for (int a = 0; a < PO_APT; ++a) {
            const int k = tid + a * PO_THREADS;
            g[a][0] = g[a][1] = g[a][2] = 0.f;
            if (k >= n) continue;
            if (all_pairs) {                      // |d - dref| over every ordered pair: (k,j) and (j,k) both pull on x_k
                for (int j = 0; j < n; ++j) {
                    const float dx = x[a][0] - sx[3 * j], dy = x[a][1] - sx[3 * j + 1], dz = x[a][2] - sx[3 * j + 2];
                    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                    const float qx = r[a][0] - sr[3 * j], qy = r[a][1] - sr[3 * j + 1], qz = r[a][2] - sr[3 * j + 2];
                    const float dev = d - sqrtf(qx * qx + qy * qy + qz * qz);
                    lsum += fabsf(dev);
                    if (d > 0.f) {
                        const float s = 2.f * (float)((dev > 0.f) - (dev < 0.f)) / d;
                        g[a][0] += s * dx; g[a][1] += s * dy; g[a][2] += s * dz;
                    }
                }
            } else {
                for (int j = 0; j < n; ++j) {     // excluded volume: 2 relu(1.22 - d) for (k,j) and for (j,k)
                    const float dx = x[a][0] - sx[3 * j], dy = x[a][1] - sx[3 * j + 1], dz = x[a][2] - sx[3 * j + 2];
                    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                    if (d < 1.22f) {
                        lsum += 2.f * (1.22f - d);          // includes j == k: the constant 2.44 per atom of the reference
                        if (d > 0.f) { const float s = -4.f / d; g[a][0] += s * dx; g[a][1] += s * dy; g[a][2] += s * dz; }
                    }
                }
                for (int e = e0[a]; e < e1[a]; ++e) {       // LAS distance term; every directed edge sits in both end lists
                    const int j = nbr_idx[e] - a0;
                    const float dx = x[a][0] - sx[3 * j], dy = x[a][1] - sx[3 * j + 1], dz = x[a][2] - sx[3 * j + 2];
                    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                    const float qx = r[a][0] - sr[3 * j], qy = r[a][1] - sr[3 * j + 1], qz = r[a][2] - sr[3 * j + 2];
                    const float dev = d - sqrtf(qx * qx + qy * qy + qz * qz);
                    lsum += 0.5f * fabsf(dev);
                    if (d > 0.f) {
                        const float s = (float)((dev > 0.f) - (dev < 0.f)) / d;
                        g[a][0] += s * dx; g[a][1] += s * dy; g[a][2] += s * dz;
                    }
                }
            }
        }
        if (t == steps) {                         // the reference returns the loss of the last epoch (before its step)
            const float tot = po_block_sum(lsum, red);
            if (tid == 0) loss_out[lig] = tot;
        }
        __syncthreads();                          // every thread has read the old coordinates
        b1t *= 0.9f; b2t *= 0.999f;
        const float step = lr / (1.f - b1t), bc2 = sqrtf(1.f - b2t);
#pragma unroll
        for (int a = 0; a < PO_APT; ++a) {
            const int k = tid + a * PO_THREADS;
            if (k >= n) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                m[a][c] = 0.9f * m[a][c] + 0.1f * g[a][c];
                v[a][c] = 0.999f * v[a][c] + 0.001f * g[a][c] * g[a][c];
                x[a][c] -= step * m[a][c] / (sqrtf(v[a][c]) / bc2 + 1e-8f);
                sx[3 * k + c] = x[a][c];
            }
        }
        __syncthreads();
    }
    float sq = 0.f;
#pragma unroll
    for (int a = 0; a < PO_APT; ++a) {
        const int k = tid + a * PO_THREADS;
        if (k >= n) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x_out[(size_t)(a0 + k) * 3 + c] = x[a][c];
            const float dq = r[a][c] - x[a][c];
            sq += dq * dq;
        }
    }
    const float tot = po_block_sum(sq, red);
    if (tid == 0) rmsd_out[lig] = sqrtf(tot / (float)max(n, 1));
}

extern "C" int fabind_post_optimize(const float* x0, const float* ref, const int* atom_off, const int* nbr_ptr,
                                    const int* nbr_idx, int n_ligands, int max_atoms, int all_pairs, int steps, float lr,
                                    float* x_out, float* loss_out, float* rmsd_out, hipStream_t stream) {
    if (n_ligands <= 0) return 0;
    FB_REQUIRE(max_atoms <= PO_THREADS * PO_APT, "fabind_post_optimize: at most 512 atoms per ligand");
    FB_REQUIRE(all_pairs || (nbr_ptr != nullptr && nbr_idx != nullptr), "fabind_post_optimize: LAS neighbour lists missing");
    FB_REQUIRE(steps >= 1, "fabind_post_optimize: steps >= 1");
    const size_t lds = (size_t)(6 * max_atoms + 8) * sizeof(float);
    hipLaunchKernelGGL(post_optim_kernel, dim3(n_ligands), dim3(PO_THREADS), lds, stream, x0, ref, atom_off, nbr_ptr, nbr_idx,
                       all_pairs, steps, lr, x_out, loss_out, rmsd_out);
    FB_CHECK_LAUNCH();
    return 0;
}
