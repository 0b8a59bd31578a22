// Attention-type kernels of the FABind layer for gfx950:
//   * protein<->ligand cross attention with gated pair bias (RowAttentionBlock, reference
//     models/cross_att.py:118-134 + models/model_utils.py:21-38,96-133), ragged (no padding);
//   * inter-graph edge attention + coordinate update (MC_Att_L, models/egnn.py:186-252), one wave per
//     aggregating node with an online segment softmax;
//   * pair-embedding helpers, LAS geometry step (models/egnn.py:433-449), row select.
#include "common.h"
#include "fabind_hip.h"

// ------------------------------------------------------------------------------------------------
// cross attention forward: 64 queries x 4 heads per work-group, keys streamed through LDS in tiles
// ------------------------------------------------------------------------------------------------
#define CA_KT 32
__global__ __launch_bounds__(256) void cross_attn_fwd_kernel(const float* __restrict__ q, int ldq,
                                                             const float* __restrict__ k, const float* __restrict__ v,
                                                             int ldkv, const float* __restrict__ gpre, int ldg,
                                                             const float* __restrict__ bias, int bias_ld, int lin_col,
                                                             int gate_col, const int* desc, float scale, float* out,
                                                             int ldo, float* lse, int ksplit, float* part) {
    __shared__ __attribute__((aligned(16))) float sK[CA_KT * 128];
    __shared__ __attribute__((aligned(16))) float sV[CA_KT * 128];
    const int* ds = desc + blockIdx.y * 8;
    const int q_off = ds[0], nq = ds[1], k_off = ds[2], nk = ds[3];
    const long pair_off = (long)(unsigned)ds[4] | ((long)ds[5] << 32);
    const int sq = ds[6], sk = ds[7];
    if ((int)(blockIdx.x * 64) >= nq) return;
    const int tid = threadIdx.x, ql = tid >> 2, h = tid & 3;
    const int qi = blockIdx.x * 64 + ql;
    const bool valid = qi < nq;
    float qr[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { qr[d] = 0.f; o[d] = 0.f; }
    if (valid) {
        const float* qp = q + (size_t)(q_off + qi) * ldq + h * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            float4 t = *(const float4*)(qp + d);
            qr[d] = t.x * scale; qr[d + 1] = t.y * scale; qr[d + 2] = t.z * scale; qr[d + 3] = t.w * scale;
        }
    }
    float m = -INFINITY, l = 0.f;
    // key range of this work-group (ksplit > 1: flash-decoding style split over the keys, merged by a second kernel)
    int kb = 0, ke = nk;
    if (ksplit > 1) {
        const int per = ((nk + ksplit - 1) / ksplit + CA_KT - 1) / CA_KT * CA_KT;
        kb = blockIdx.z * per; ke = min(nk, kb + per);
    }
    for (int j0 = kb; j0 < ke; j0 += CA_KT) {
        __syncthreads();
        for (int i = tid; i < CA_KT * 32; i += 256) {  // 32 float4 per key row
            int jr = i >> 5, c4 = (i & 31) * 4;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (j0 + jr < ke) {
                kk = *(const float4*)(k + (size_t)(k_off + j0 + jr) * ldkv + c4);
                vv = *(const float4*)(v + (size_t)(k_off + j0 + jr) * ldkv + c4);
            }
            *(float4*)(&sK[jr * 128 + c4]) = kk;
            *(float4*)(&sV[jr * 128 + c4]) = vv;
        }
        __syncthreads();
        if (!valid) continue;
        const int jn = min(CA_KT, ke - j0);
        for (int j = 0; j < jn; ++j) {
            const float* kp = &sK[j * 128 + h * 32];
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; d += 4) {
                float4 t = *(const float4*)(kp + d);
                s += qr[d] * t.x + qr[d + 1] * t.y + qr[d + 2] * t.z + qr[d + 3] * t.w;
            }
            const size_t bi = (size_t)(pair_off + (long)qi * sq + (long)(j0 + j) * sk) * bias_ld;
            s += bias[bi + lin_col + h] * sigmoid_f(bias[bi + gate_col + h]);
            const float mn = fmaxf(m, s);
            const float corr = __expf(m - mn), pj = __expf(s - mn);
            l = l * corr + pj;
            const float* vp = &sV[j * 128 + h * 32];
#pragma unroll
            for (int d = 0; d < 32; d += 4) {
                float4 t = *(const float4*)(vp + d);
                o[d] = o[d] * corr + pj * t.x; o[d + 1] = o[d + 1] * corr + pj * t.y;
                o[d + 2] = o[d + 2] * corr + pj * t.z; o[d + 3] = o[d + 3] * corr + pj * t.w;
            }
            m = mn;
        }
    }
    if (!valid) return;
    if (ksplit > 1) {      // un-normalised partial: [row][split][head][34] = o[32], m, l
        float* pp = part + (((size_t)(q_off + qi) * ksplit + blockIdx.z) * 4 + h) * 34;
#pragma unroll
        for (int d = 0; d < 32; ++d) pp[d] = o[d];
        pp[32] = m; pp[33] = l;
        return;
    }
    const float inv = 1.f / l;
    const float* gp = gpre + (size_t)(q_off + qi) * ldg + h * 32;
    float* op = out + (size_t)(q_off + qi) * ldo + h * 32;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
        float4 g = *(const float4*)(gp + d);
        *(float4*)(op + d) = make_float4(o[d] * inv * sigmoid_f(g.x), o[d + 1] * inv * sigmoid_f(g.y),
                                         o[d + 2] * inv * sigmoid_f(g.z), o[d + 3] * inv * sigmoid_f(g.w));
    }
    if (lse) lse[(size_t)(q_off + qi) * 4 + h] = m + __logf(l);
}

__global__ __launch_bounds__(256) void cross_attn_combine_kernel(const float* __restrict__ part, int ksplit,
                                                                 const float* __restrict__ gpre, int ldg, int n_rows,
                                                                 float* out, int ldo, float* lse) {
    const int t = blockIdx.x * 256 + threadIdx.x;          // (row, head)
    if (t >= n_rows * 4) return;
    const int rowi = t >> 2, h = t & 3;
    float M = -INFINITY;
    for (int s_ = 0; s_ < ksplit; ++s_) M = fmaxf(M, part[(((size_t)rowi * ksplit + s_) * 4 + h) * 34 + 32]);
    float L = 0.f, o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    for (int s_ = 0; s_ < ksplit; ++s_) {
        const float* pp = part + (((size_t)rowi * ksplit + s_) * 4 + h) * 34;
        const float w = (pp[33] > 0.f) ? __expf(pp[32] - M) : 0.f;
        L += pp[33] * w;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] += pp[d] * w;
    }
    const float inv = 1.f / L;
#pragma unroll
    for (int d = 0; d < 32; ++d)
        out[(size_t)rowi * ldo + h * 32 + d] = o[d] * inv * sigmoid_f(gpre[(size_t)rowi * ldg + h * 32 + d]);
    if (lse) lse[(size_t)rowi * 4 + h] = M + __logf(L);
}

extern "C" int fabind_cross_attn_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv,
                                     const float* gpre, int ldg, const float* bias, int bias_ld, int lin_col,
                                     int gate_col, const int* desc, int B, int max_nq, float scale, float* out, int ldo,
                                     float* lse, int ksplit, float* part, int n_rows, hipStream_t stream) {
    FB_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0, "fabind_cross_attn_fwd: strides % 4");
    if (B <= 0 || max_nq <= 0) return 0;
    if (ksplit < 1) ksplit = 1;
    dim3 grid((max_nq + 63) / 64, B, ksplit);
    hipLaunchKernelGGL(cross_attn_fwd_kernel, grid, dim3(256), 0, stream, q, ldq, k, v, ldkv, gpre, ldg, bias, bias_ld,
                       lin_col, gate_col, desc, scale, out, ldo, lse, ksplit, part);
    if (ksplit > 1)
        hipLaunchKernelGGL(cross_attn_combine_kernel, dim3((n_rows * 4 + 255) / 256), dim3(256), 0, stream, part, ksplit,
                           gpre, ldg, n_rows, out, ldo, lse);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// pair helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_bmat_kernel(const float* __restrict__ b0, int ldb,
                                                        const float* __restrict__ wcomp, int NO, int H,
                                                        const int* c_node, int n_rows, void* bmat, int bmat_dt) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int j = r / NO, o = r % NO;
    const float* bp = b0 + (size_t)c_node[j] * ldb;
    const float* wp = wcomp + (size_t)o * H;
    for (int c = lane * 4; c < H; c += 256) {
        float4 a = *(const float4*)(bp + c), w = *(const float4*)(wp + c);
        st4_any(bmat, bmat_dt, (size_t)r * H + c, make_float4(a.x * w.x, a.y * w.y, a.z * w.z, a.w * w.w));
    }
}

extern "C" int fabind_pair_bmat(const float* b0, int ldb, const float* wcomp, int NO, int H, const int* c_node, int n_c,
                                void* bmat, int bmat_dt, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && ldb % 4 == 0, "fabind_pair_bmat: H/ldb % 4");
    int n_rows = n_c * NO;
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(pair_bmat_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, b0, ldb, wcomp, NO, H, c_node,
                       n_rows, bmat, bmat_dt);
    FB_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void pair_hadamard_kernel(const float* __restrict__ a0, const float* __restrict__ b0,
                                                            int ld0, int H, const float* __restrict__ a1,
                                                            const float* __restrict__ b1, int ld1, int H2,
                                                            const int* red_p, const int* red_c, int n_red, void* hd,
                                                            int hd_dt, int ldh, int lp) {
    // lp lanes per pair (16 / 32 / 64): narrow products (the 64-wide all-pairs product of FABind+, 3.9 M pairs in the pocket
    // model) put 4 / 2 pairs on a wave -- one pair per wave kept 16 lanes busy and wrote 504 MB in 828 us
    const int lane = threadIdx.x & 63, sl = lane % lp;
    const int e = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lp) + lane / lp;
    if (e >= n_red) return;
    const int pn = red_p[e], cn = red_c[e];
    for (int c = sl * 4; c < H; c += lp * 4) {
        float4 a = *(const float4*)(a0 + (size_t)pn * ld0 + c), b = *(const float4*)(b0 + (size_t)cn * ld0 + c);
        st4_any(hd, hd_dt, (size_t)e * ldh + c, make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w));
    }
    for (int c = sl * 4; c < H2; c += lp * 4) {
        float4 a = *(const float4*)(a1 + (size_t)pn * ld1 + c), b = *(const float4*)(b1 + (size_t)cn * ld1 + c);
        st4_any(hd, hd_dt, (size_t)e * ldh + H + c, make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w));
    }
}

extern "C" int fabind_pair_hadamard(const float* a0, const float* b0, int ld0, int H, const float* a1, const float* b1,
                                    int ld1, int H2, const int* red_p, const int* red_c, int n_red, void* hd, int hd_dt,
                                    int ldh, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && H2 % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 && ldh % 4 == 0, "fabind_pair_hadamard: % 4");
    if (n_red <= 0) return 0;
    const int widest = H > H2 ? H : H2;
    const int lp = widest <= 64 ? 16 : widest <= 128 ? 32 : 64;
    const int per_block = 4 * (64 / lp);
    hipLaunchKernelGGL(pair_hadamard_kernel, dim3((n_red + per_block - 1) / per_block), dim3(256), 0, stream, a0, b0, ld0, H, a1, b1, ld1,
                       H2, red_p, red_c, n_red, hd, hd_dt, ldh, lp);
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// inter-graph edge attention, forward.  One wave per aggregating node.
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void inter_attn_fwd_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* __restrict__ h,
    int ldh, const float* __restrict__ x, const float* __restrict__ d, const float* __restrict__ rhohat,
    const int* rowptr, const int* col, const int* red_idx, const float* bias_red, int bias_np,
    const float* __restrict__ w_rk, const float* __restrict__ w_rv, const float* __restrict__ wcr,
    const float* __restrict__ w3, float clampv, int n_rows, float* h_out, float* x_out, float* alpha, float* cvs, bf16_t* h16,
    const float* __restrict__ s_ext) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    if (e0 == e1) {                        // no inter edges (92 % of the rows of the bench batch): h and x pass through
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            if (c < H) {
                const float4 o = *(const float4*)(h + (size_t)r * ldh + c);
                *(float4*)(h_out + (size_t)r * ldh + c) = o;
                if (h16) *(uint2*)(h16 + (size_t)r * H + c) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));
            }
        }
        if (lane < 3) x_out[(size_t)r * 3 + lane] = x[(size_t)r * 3 + lane];
        return;
    }
    float4 qv[NS], acc[NS], wk[NS], wc[NS], w3v[NS];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float qw = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int c = s * 256 + lane * 4;
        qv[s] = z4; acc[s] = z4; wk[s] = z4; wc[s] = z4; w3v[s] = z4;
        if (c < H) {
            qv[s] = *(const float4*)(qkv + (size_t)r * ldqkv + c);
            wk[s] = *(const float4*)(w_rk + c);
            wc[s] = *(const float4*)(wcr + c);
            w3v[s] = *(const float4*)(w3 + c);
            qw += qv[s].x * wk[s].x + qv[s].y * wk[s].y + qv[s].z * wk[s].z + qv[s].w * wk[s].w;
        }
    }
    qw = wave_sum(qw);
    float m = -INFINITY, l = 0.f, sar = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    // The edge loop is latency-bound: three gathered 2 KB rows per edge, then two wave reductions and two exponentials that depend
    // on them.  (a) The rows of edge j + 1 are requested before edge j is reduced (software prefetch, one edge deep).  (b) Everything
    // else an edge needs -- its sending node, rhohat, the summed attention bias (a load through red_idx[]: two dependent global
    // loads), the three components of d, the FABind+ scalar -- is loaded for up to 64 edges AT ONCE, one edge per lane, ahead of the
    // loop and handed out with v_readlane; the per-edge results (logit, coordinate scalar) collect in the lane of their edge and leave
    // as one coalesced store per batch.  Inside the loop those were ~2 us of exposed latency per edge in a serial chain of ~27 edges
    // (7,380 of the 98,688 rows of the bench batch hold all inter edges, 5,496 of them 96 %: few waves, long chains -- the launch
    // took 366 us at 98,688 nodes and 156 us at 9,088 with the same 1.3e5-1.6e5 edges).
    float4 nk[NS], nv[NS], nc[NS];
    auto fetch = [&](int cn) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            nk[s] = z4; nv[s] = z4; nc[s] = z4;
            if (c < H) {
                nk[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + H + c);
                nv[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + 2 * H + c);
                nc[s] = *(const float4*)(cv + (size_t)cn * ldcv + c);
            }
        }
    };
    for (int eb = e0; eb < e1; eb += 64) {
        const int ne = min(64, e1 - eb);
        int my_col = 0;
        float my_rh = 0.f, my_b = 0.f, my_dx = 0.f, my_dy = 0.f, my_dz = 0.f, my_logit = 0.f, my_cp = 0.f;
        if (lane < ne) {
            const int e = eb + lane;
            my_col = col[e];
            my_rh = rhohat[e];
            const int ri = red_idx[e];
            for (int kb = 0; kb < bias_np; ++kb) my_b += bias_red[(size_t)ri * bias_np + kb];
            my_dx = d[(size_t)e * 3]; my_dy = d[(size_t)e * 3 + 1]; my_dz = d[(size_t)e * 3 + 2];
            if (s_ext) my_cp = s_ext[e];           // FABind+: the coord_mlp is an LN-MLP evaluated outside (per-edge scalar)
        }
        fetch(__shfl(my_col, 0, 64));
        for (int j = 0; j < ne; ++j) {
            const float rh = __shfl(my_rh, j, 64);
            float lp = 0.f, cp = 0.f;
            float4 vv[NS], kk[NS], cc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { kk[s] = nk[s]; vv[s] = nv[s]; cc[s] = nc[s]; }
            if (j + 1 < ne) fetch(__shfl(my_col, j + 1, 64));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                int c = s * 256 + lane * 4;
                if (c < H) {
                    lp += qv[s].x * kk[s].x + qv[s].y * kk[s].y + qv[s].z * kk[s].z + qv[s].w * kk[s].w;
                    cp += w3v[s].x * silu_f(cc[s].x + rh * wc[s].x) + w3v[s].y * silu_f(cc[s].y + rh * wc[s].y) +
                          w3v[s].z * silu_f(cc[s].z + rh * wc[s].z) + w3v[s].w * silu_f(cc[s].w + rh * wc[s].w);
                }
            }
            lp = wave_sum(lp);
            cp = wave_sum(cp) + __shfl(my_cp, j, 64);
            const float logit = lp + rh * qw + __shfl(my_b, j, 64);
            const float mn = fmaxf(m, logit);
            const float corr = __expf(m - mn), pj = __expf(logit - mn);
            l = l * corr + pj;
            sar = sar * corr + pj * rh;
            const float t = pj * cp;
            ax = ax * corr + t * __shfl(my_dx, j, 64); ay = ay * corr + t * __shfl(my_dy, j, 64); az = az * corr + t * __shfl(my_dz, j, 64);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc[s].x = acc[s].x * corr + pj * vv[s].x; acc[s].y = acc[s].y * corr + pj * vv[s].y;
                acc[s].z = acc[s].z * corr + pj * vv[s].z; acc[s].w = acc[s].w * corr + pj * vv[s].w;
            }
            m = mn;
            if (lane == j) { my_logit = logit; my_cp = cp; }
        }
        if (lane < ne) { alpha[eb + lane] = my_logit; cvs[eb + lane] = my_cp; }
    }
    const float inv = (e1 > e0) ? 1.f / l : 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int c = s * 256 + lane * 4;
        if (c < H) {
            float4 hv = *(const float4*)(h + (size_t)r * ldh + c);
            float4 wv = *(const float4*)(w_rv + c);
            float4 o = make_float4(hv.x + (acc[s].x + sar * wv.x) * inv, hv.y + (acc[s].y + sar * wv.y) * inv,
                                   hv.z + (acc[s].z + sar * wv.z) * inv, hv.w + (acc[s].w + sar * wv.w) * inv);
            *(float4*)(h_out + (size_t)r * ldh + c) = o;
            if (h16) *(uint2*)(h16 + (size_t)r * H + c) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));   // the next contraction's operand
        }
    }
    if (lane == 0) {
        float tx = fminf(fmaxf(ax * inv, -clampv), clampv), ty = fminf(fmaxf(ay * inv, -clampv), clampv),
              tz = fminf(fmaxf(az * inv, -clampv), clampv);
        x_out[(size_t)r * 3] = x[(size_t)r * 3] + tx;
        x_out[(size_t)r * 3 + 1] = x[(size_t)r * 3 + 1] + ty;
        x_out[(size_t)r * 3 + 2] = x[(size_t)r * 3 + 2] + tz;
    }
    // the wave's own stores to alpha[] above are visible to it after this barrier-free fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = e0 + lane; e < e1; e += 64) alpha[e] = __expf(alpha[e] - m) * inv;
}

extern "C" int fabind_inter_attn_fwd(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* h,
                                     int ldh, const float* x, const float* d, const float* rhohat, const int* rowptr,
                                     const int* col, const int* red_idx, const float* bias_red, int bias_np,
                                     const float* w_rk, const float* w_rv, const float* wcr, const float* w3,
                                     float clampv, int n_rows, float* h_out, float* x_out, float* alpha, float* cvs,
                                     const float* s_ext, void* h16, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && ldqkv % 4 == 0 && ldcv % 4 == 0 && ldh % 4 == 0, "fabind_inter_attn_fwd: % 4");
    FB_REQUIRE(H <= 1024, "fabind_inter_attn_fwd: H <= 1024");
    if (n_rows <= 0) return 0;
    dim3 grid((n_rows + 3) / 4), block(256);
#define LAUNCH(NS)                                                                                                   \
    hipLaunchKernelGGL((inter_attn_fwd_kernel<NS>), grid, block, 0, stream, qkv, ldqkv, cv, ldcv, H, h, ldh, x, d,    \
                       rhohat, rowptr, col, red_idx, bias_red, bias_np, w_rk, w_rv, wcr, w3, clampv, n_rows, h_out, \
                       x_out, alpha, cvs, (bf16_t*)h16, s_ext)
    if (H <= 256) LAUNCH(1); else if (H <= 512) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LAS geometry step, row select, add
// ------------------------------------------------------------------------------------------------
// Ligand atoms (1 <= u < C): ONE WAVE per atom, lanes stride over the complex's LAS edges (154 at 40 atoms), fixed-order wave sum;
// every other node is a copy, one thread each (blocks [0, nb_copy)).  (Round 3: one THREAD per atom walking all edges serially --
// 37 us per call for 2,560 atoms, latency of 154 dependent index loads.)
// NO PACKED fp32 MATH in this kernel (`target("no-packed-fp32-ops")`, checked on the built code object by tools/isa_lint.py).  Round 5:
// with the fy / fz accumulations paired into one `v_pk_fma_f32 ... op_sel:[0,1,0]`, the y sum of lanes 48-63 was LOST in 1-3 % of the
// full-model passes when several processes shared the device (profiles/r05_contention.txt: 13 of 414 contended passes; the dumped cases
// show exactly the upper row's y terms missing; 0 of 414 with scalar FMAs; never single-process).  Unlike the weight-gradient
// contraction's mismatch (an inline-asm hazard, gemm.hip) the compiler-generated waits of this kernel are complete (the lint covers
// them) and the kernel has no LDS traffic of its own: the cause is NOT established.  Round 6 replaces the empty-asm trick that kept the
// SLP vectoriser from pairing the accumulations (it depended on the optimiser's mood) by switching packed fp32 off for the function,
// and keeps the kernel in the device-sharing stress test (tests/test_gpu_contention.py).
__global__ __launch_bounds__(256) FB_NO_PACKED_F32 void las_step_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                       const int* las_i, const int* las_j, const int* las_off,
                                                       const int* node_off, const int* c_cnt, float step, float clampv,
                                                       float* x_out, int nb_copy) {
    const int b = blockIdx.y;
    const int off = node_off[b], n = node_off[b + 1] - off, C = c_cnt[b];
    if ((int)blockIdx.x < nb_copy) {
        const int u = blockIdx.x * 256 + threadIdx.x;
        if (u >= n || (u >= 1 && u < C)) return;
        const int gu = off + u;
        x_out[(size_t)gu * 3] = x[(size_t)gu * 3];
        x_out[(size_t)gu * 3 + 1] = x[(size_t)gu * 3 + 1];
        x_out[(size_t)gu * 3 + 2] = x[(size_t)gu * 3 + 2];
        return;
    }
    const int lane = threadIdx.x & 63;
    const int u = 1 + ((int)blockIdx.x - nb_copy) * 4 + (threadIdx.x >> 6);
    if (u >= C || u >= n) return;
    const int gu = off + u;
    const float gx = x[(size_t)gu * 3], gy = x[(size_t)gu * 3 + 1], gz = x[(size_t)gu * 3 + 2];
    const float hx = x0[(size_t)gu * 3], hy = x0[(size_t)gu * 3 + 1], hz = x0[(size_t)gu * 3 + 2];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int e = las_off[b] + lane; e < las_off[b + 1]; e += 64) {
        if (las_j[e] != gu) continue;
        const int i = las_i[e];
        float dx = x[i * 3] - gx, dy = x[i * 3 + 1] - gy, dz = x[i * 3 + 2] - gz;
        float ex = x0[i * 3] - hx, ey = x0[i * 3 + 1] - hy, ez = x0[i * 3 + 2] - hz;
        float cur = dx * dx + dy * dy + dz * dz, tru = ex * ex + ey * ey + ez * ez;
        float f = 2.f * (cur - tru);
        fx = __builtin_fmaf(f, 2.f * dx, fx);
        fy = __builtin_fmaf(f, 2.f * dy, fy);
        fz = __builtin_fmaf(f, 2.f * dz, fz);
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
        x_out[(size_t)gu * 3] = gx + fminf(fmaxf(fx * step, -clampv), clampv);
        x_out[(size_t)gu * 3 + 1] = gy + fminf(fmaxf(fy * step, -clampv), clampv);
        x_out[(size_t)gu * 3 + 2] = gz + fminf(fmaxf(fz * step, -clampv), clampv);
    }
}

extern "C" int fabind_las_step(const float* x, const float* x0, const int* las_i, const int* las_j, const int* las_off,
                               const int* node_off, const int* c_cnt, int B, int max_n, float step, float clampv,
                               float* x_out, hipStream_t stream) {
    if (B <= 0) return 0;
    const int nb_copy = (max_n + 255) / 256;
    hipLaunchKernelGGL(las_step_kernel, dim3(nb_copy + (max_n + 3) / 4, B), dim3(256), 0, stream, x, x0, las_i, las_j, las_off,
                       node_off, c_cnt, step, clampv, x_out, nb_copy);
    FB_CHECK_LAUNCH();
    return 0;
}

__global__ void select_rows_kernel(const float* x, const float* z, const uint8_t* mask, int n, int width, float* out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * width) return;
    out[i] = mask[i / width] ? z[i] : x[i];
}

extern "C" int fabind_select_rows(const float* x, const float* z, const uint8_t* mask, int n, int width, float* x_out,
                                  hipStream_t stream) {
    long tot = (long)n * width;
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, x, z, mask, n, width,
                       x_out);
    FB_CHECK_LAUNCH();
    return 0;
}

__global__ void add_kernel(const float* a, const float* b, float* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

extern "C" int fabind_add(const float* a, const float* b, float* out, long n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
    FB_CHECK_LAUNCH();
    return 0;
}
