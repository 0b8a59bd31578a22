// Inter-graph edge attention (reference FABind/fabind/models/egnn.py:186-252: MC_Att_L.att_model / node_model / coord_model with
// scatter_softmax over the receiving node) with the ROWS OF THE INTER GRAPH DEALT BY DEGREE -- round 5.
//
// The one-wave-per-row kernels (attn.hip: inter_attn_fwd_kernel, bwd.hip: inter_attn_bwd_a / _b) walk a row's edges as ONE serial chain:
// three gathered 2 KB rows, two wave reductions and two exponentials per edge, one edge prefetched.  At the headline shape 7,380 of the
// 98,688 rows hold all ~1.5e5 inter edges and 5,496 of them 96 % (the ligand atoms: ~27 edges on average, up to ~80): few waves, long chains
// -- 0 % matrix-core and 16-21 % VALU utilisation, the launch time is the longest chain's.  Here the host hands over the rows SORTED BY
// DEGREE (`order`, descending; engine.Graph builds it next to the CSR, one stable sort per graph) with two counts:
//     order[0 .. n_heavy)        rows with more than IA_HEAVY edges: ONE WORK-GROUP (4 waves) per row, each wave a contiguous quarter of the
//                                row's edges, partial (max, sum, weighted sums) per wave, log-sum-exp combine through LDS in fixed wave order;
//     order[n_heavy .. n_act)    rows with 1 .. IA_HEAVY edges: one wave per row, four rows per work-group (the old form);
//     order[n_act .. n_rows)     rows without inter edges (92 % of the headline batch): pass-through copy / zero fill, one wave per row.
// Chains are 4x shorter and 4x as many waves are in flight; heavy rows start first.  Same arithmetic per edge as the one-wave kernels; the
// sums of a heavy row are taken quarter by quarter (fixed order: bit-identical repeats; not bit-identical to the one-wave kernels).
#include "common.h"
#include "fabind_hip.h"

#define IA_HEAVY 8          // rows with more edges than this are split over the four waves of a work-group (host: engine.Graph)

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void inter_attn_fwd_rows_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* __restrict__ h,
    int ldh, const float* __restrict__ x, const float* __restrict__ d, const float* __restrict__ rhohat,
    const int* rowptr, const int* col, const int* red_idx, const float* bias_red, int bias_np,
    const float* __restrict__ w_rk, const float* __restrict__ w_rv, const float* __restrict__ wcr,
    const float* __restrict__ w3, float clampv, int n_rows, float* h_out, float* x_out, float* alpha, float* cvs, bf16_t* h16,
    const float* __restrict__ s_ext, const int* __restrict__ order, int n_heavy, int n_act) {
    __shared__ float4 sAcc[4][NS * 64];
    __shared__ float sSc[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool coop = (int)blockIdx.x < n_heavy;
    const int idx = coop ? (int)blockIdx.x : n_heavy + ((int)blockIdx.x - n_heavy) * 4 + wv;
    if (idx >= n_rows) return;                                   // (only waves of the last one-wave-per-row work-group)
    const int r = order[idx];
    if (idx >= n_act) {                                          // no inter edges: h and x pass through
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
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    int ws = e0, we = e1;                                        // this wave's edges
    if (coop) {
        const int per = (e1 - e0 + 3) >> 2;
        ws = min(e1, e0 + wv * per);
        we = min(e1, ws + per);
    }
    float4 qv[NS], acc[NS], wk[NS], wc[NS], w3v[NS];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float qw = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
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
    // (per-edge scalars for up to 64 edges at once, one edge per lane, handed out with v_readlane; the rows of edge j + 1 requested before
    //  edge j is reduced: see inter_attn_fwd_kernel in attn.hip)
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
    for (int eb = ws; eb < we; eb += 64) {
        const int ne = min(64, we - eb);
        int my_col = 0;
        float my_rh = 0.f, my_b = 0.f, my_dx = 0.f, my_dy = 0.f, my_dz = 0.f, my_logit = 0.f, my_cp = 0.f;
        if (lane < ne) {
            const int e = eb + lane;
            my_col = col[e];
            my_rh = rhohat[e];
            const int ri = red_idx[e];
            for (int kb = 0; kb < bias_np; ++kb) my_b += bias_red[(size_t)ri * bias_np + kb];
            my_dx = d[(size_t)e * 3]; my_dy = d[(size_t)e * 3 + 1]; my_dz = d[(size_t)e * 3 + 2];
            if (s_ext) my_cp = s_ext[e];
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
                const int c = s * 256 + lane * 4;
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
    float inv;
    if (!coop) {
        inv = 1.f / l;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            if (c < H) {
                const float4 hv = *(const float4*)(h + (size_t)r * ldh + c);
                const float4 wvv = *(const float4*)(w_rv + c);
                const float4 o = make_float4(hv.x + (acc[s].x + sar * wvv.x) * inv, hv.y + (acc[s].y + sar * wvv.y) * inv,
                                             hv.z + (acc[s].z + sar * wvv.z) * inv, hv.w + (acc[s].w + sar * wvv.w) * inv);
                *(float4*)(h_out + (size_t)r * ldh + c) = o;
                if (h16) *(uint2*)(h16 + (size_t)r * H + c) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));
            }
        }
        if (lane == 0) {
            const float tx = fminf(fmaxf(ax * inv, -clampv), clampv), ty = fminf(fmaxf(ay * inv, -clampv), clampv),
                        tz = fminf(fmaxf(az * inv, -clampv), clampv);
            x_out[(size_t)r * 3] = x[(size_t)r * 3] + tx;
            x_out[(size_t)r * 3 + 1] = x[(size_t)r * 3 + 1] + ty;
            x_out[(size_t)r * 3 + 2] = x[(size_t)r * 3 + 2] + tz;
        }
    } else {
        // log-sum-exp combine of the four waves' partials, wave 0 .. 3 in order
#pragma unroll
        for (int s = 0; s < NS; ++s) sAcc[wv][s * 64 + lane] = acc[s];
        if (lane == 0) { sSc[wv][0] = m; sSc[wv][1] = l; sSc[wv][2] = sar; sSc[wv][3] = ax; sSc[wv][4] = ay; sSc[wv][5] = az; }
        __syncthreads();
        const float M = fmaxf(fmaxf(sSc[0][0], sSc[1][0]), fmaxf(sSc[2][0], sSc[3][0]));
        float f[4], L = 0.f, SAR = 0.f, AX = 0.f, AY = 0.f, AZ = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            f[w] = sSc[w][1] > 0.f ? __expf(sSc[w][0] - M) : 0.f;
            L += f[w] * sSc[w][1]; SAR += f[w] * sSc[w][2]; AX += f[w] * sSc[w][3]; AY += f[w] * sSc[w][4]; AZ += f[w] * sSc[w][5];
        }
        inv = 1.f / L;
        for (int g = threadIdx.x; g < NS * 64; g += 256) {
            const int c = (g >> 6) * 256 + (g & 63) * 4;
            if (c < H) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 p = sAcc[w][g];
                    a.x += f[w] * p.x; a.y += f[w] * p.y; a.z += f[w] * p.z; a.w += f[w] * p.w;
                }
                const float4 hv = *(const float4*)(h + (size_t)r * ldh + c);
                const float4 wvv = *(const float4*)(w_rv + c);
                const float4 o = make_float4(hv.x + (a.x + SAR * wvv.x) * inv, hv.y + (a.y + SAR * wvv.y) * inv,
                                             hv.z + (a.z + SAR * wvv.z) * inv, hv.w + (a.w + SAR * wvv.w) * inv);
                *(float4*)(h_out + (size_t)r * ldh + c) = o;
                if (h16) *(uint2*)(h16 + (size_t)r * H + c) = make_uint2(pack2_bf16(o.x, o.y), pack2_bf16(o.z, o.w));
            }
        }
        if (threadIdx.x == 0) {
            const float tx = fminf(fmaxf(AX * inv, -clampv), clampv), ty = fminf(fmaxf(AY * inv, -clampv), clampv),
                        tz = fminf(fmaxf(AZ * inv, -clampv), clampv);
            x_out[(size_t)r * 3] = x[(size_t)r * 3] + tx;
            x_out[(size_t)r * 3 + 1] = x[(size_t)r * 3 + 1] + ty;
            x_out[(size_t)r * 3 + 2] = x[(size_t)r * 3 + 2] + tz;
        }
        m = M;
    }
    // the wave's own stores to alpha[] above are visible to it after this barrier-free fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = ws + lane; e < we; e += 64) alpha[e] = __expf(alpha[e] - m) * inv;
}

extern "C" int fabind_inter_attn_fwd_rows(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* h,
                                          int ldh, const float* x, const float* d, const float* rhohat, const int* rowptr,
                                          const int* col, const int* red_idx, const float* bias_red, int bias_np,
                                          const float* w_rk, const float* w_rv, const float* wcr, const float* w3,
                                          float clampv, int n_rows, float* h_out, float* x_out, float* alpha, float* cvs,
                                          const float* s_ext, void* h16, const int* order, int n_heavy, int n_act, hipStream_t stream) {
    FB_REQUIRE(H % 4 == 0 && ldqkv % 4 == 0 && ldcv % 4 == 0 && ldh % 4 == 0, "fabind_inter_attn_fwd_rows: % 4");
    FB_REQUIRE(H <= 1024, "fabind_inter_attn_fwd_rows: H <= 1024");
    FB_REQUIRE(order != nullptr && 0 <= n_heavy && n_heavy <= n_act && n_act <= n_rows, "fabind_inter_attn_fwd_rows: order / counts");
    if (n_rows <= 0) return 0;
    dim3 grid(n_heavy + (n_rows - n_heavy + 3) / 4), block(256);
#define LAUNCH(NS)                                                                                                   \
    hipLaunchKernelGGL((inter_attn_fwd_rows_kernel<NS>), grid, block, 0, stream, qkv, ldqkv, cv, ldcv, H, h, ldh, x, d, \
                       rhohat, rowptr, col, red_idx, bias_red, bias_np, w_rk, w_rv, wcr, w3, clampv, n_rows, h_out, \
                       x_out, alpha, cvs, (bf16_t*)h16, s_ext, order, n_heavy, n_act)
    if (H <= 256) LAUNCH(1); else if (H <= 512) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    FB_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward, pass A (softmax adjoint per row -> dlogit[e], dcp[e], dd[e], drh[e], dq[r], per-block partials of the four vector gradients)
// Persistent work-groups: first the heavy rows, one per work-group and turn (four waves, a quarter of the edges each; S = sum alpha dalpha
// and dq meet through LDS), then the light and the empty rows, one per wave.
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void inter_attn_bwd_a_rows_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* __restrict__ d,
    const float* __restrict__ rhohat, const int* rowptr, const int* col, const float* __restrict__ w_rk,
    const float* __restrict__ w_rv, const float* __restrict__ wcr, const float* __restrict__ w3, const float* alpha,
    const float* cvs, float clampv, int n_rows, const float* __restrict__ dh_out, const float* __restrict__ dx_out,
    float* dqkv, float* dd, float* drh, float* dlogit, float* dcp, float* wpart, const int* __restrict__ order, int n_heavy, int n_act) {
    extern __shared__ float4 shp[];   // [4 quantities][4 waves][NS*64] float4 (final partials); its head doubles as the dq exchange of a heavy row
    float (*sS)[4] = (float (*)[4])(shp + 4 * NS * 64);          // [4 waves][4]: the row sums of a heavy row (right behind the dq exchange)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_wrk[NS], a_wrv[NS], a_wcr[NS], a_w3[NS], wk[NS], wvv[NS], wc[NS], w3v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
        a_wrk[s] = z4; a_wrv[s] = z4; a_wcr[s] = z4; a_w3[s] = z4; wk[s] = z4; wvv[s] = z4; wc[s] = z4; w3v[s] = z4;
        if (c < H) {
            wk[s] = *(const float4*)(w_rk + c); wvv[s] = *(const float4*)(w_rv + c);
            wc[s] = *(const float4*)(wcr + c); w3v[s] = *(const float4*)(w3 + c);
        }
    }
    const int n_hturns = (n_heavy + (int)gridDim.x - 1) / (int)gridDim.x;        // block-uniform trip count of the cooperative phase
    const int n_lturns = (n_rows - n_heavy + (int)gridDim.x * 4 - 1) / ((int)gridDim.x * 4);
    for (int turn = 0; turn < n_hturns + n_lturns; ++turn) {
        const bool coop = turn < n_hturns;
        int idx;
        if (coop) idx = turn * (int)gridDim.x + (int)blockIdx.x;
        else idx = n_heavy + ((turn - n_hturns) * (int)gridDim.x + (int)blockIdx.x) * 4 + wv;
        if (coop ? idx >= n_heavy : idx >= n_rows) continue;     // (block-uniform in a cooperative turn: all four waves share idx)
        const int r = order[idx];
        if (idx >= n_act) {                                      // no inter edges: d q = 0 (the caller does not pre-zero dqkv)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                if (c < H) *(float4*)(dqkv + (size_t)r * ldqkv + c) = z4;
            }
            continue;
        }
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        int ws = e0, we = e1;
        if (coop) {
            const int per = (e1 - e0 + 3) >> 2;
            ws = min(e1, e0 + wv * per);
            we = min(e1, ws + per);
        }
        float4 qv[NS], gh[NS], dq[NS];
        float qw = 0.f, gwv = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            qv[s] = z4; gh[s] = z4; dq[s] = z4;
            if (c < H) {
                qv[s] = *(const float4*)(qkv + (size_t)r * ldqkv + c);
                gh[s] = *(const float4*)(dh_out + (size_t)r * H + c);
                qw += qv[s].x * wk[s].x + qv[s].y * wk[s].y + qv[s].z * wk[s].z + qv[s].w * wk[s].w;
                gwv += gh[s].x * wvv[s].x + gh[s].y * wvv[s].y + gh[s].z * wvv[s].z + gh[s].w * wvv[s].w;
            }
        }
        qw = wave_sum(qw); gwv = wave_sum(gwv);
        // recompute t_r for the clamp mask (every wave of a heavy row sums the whole row: scalar loads only)
        float tx = 0.f, ty = 0.f, tz = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) {
            const float w = alpha[e] * cvs[e];
            tx += w * d[(size_t)e * 3]; ty += w * d[(size_t)e * 3 + 1]; tz += w * d[(size_t)e * 3 + 2];
        }
        tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
        const float gx = fabsf(tx) <= clampv ? dx_out[(size_t)r * 3] : 0.f;
        const float gy = fabsf(ty) <= clampv ? dx_out[(size_t)r * 3 + 1] : 0.f;
        const float gz = fabsf(tz) <= clampv ? dx_out[(size_t)r * 3 + 2] : 0.f;
        // pass 1: dalpha_e, stored in dlogit[]; S = sum alpha dalpha
        float S = 0.f;
        float4 pa[NS], pb[NS];
        auto fetch_v = [&](int cn) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                pa[s] = z4;
                if (c < H) pa[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + 2 * H + c);
            }
        };
        for (int eb = ws; eb < we; eb += 64) {
            const int ne = min(64, we - eb);
            int my_col = 0;
            float my_rh = 0.f, my_al = 0.f, my_cvs = 0.f, my_gd = 0.f, my_da = 0.f;
            if (lane < ne) {
                const int e = eb + lane;
                my_col = col[e]; my_rh = rhohat[e]; my_al = alpha[e]; my_cvs = cvs[e];
                my_gd = gx * d[(size_t)e * 3] + gy * d[(size_t)e * 3 + 1] + gz * d[(size_t)e * 3 + 2];
            }
            fetch_v(__shfl(my_col, 0, 64));
            for (int j = 0; j < ne; ++j) {
                const float rh = __shfl(my_rh, j, 64);
                float hp = 0.f;
                float4 vv[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) vv[s] = pa[s];
                if (j + 1 < ne) fetch_v(__shfl(my_col, j + 1, 64));
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    hp += gh[s].x * vv[s].x + gh[s].y * vv[s].y + gh[s].z * vv[s].z + gh[s].w * vv[s].w;
                hp = wave_sum(hp) + rh * gwv;
                const float da = hp + __shfl(my_cvs, j, 64) * __shfl(my_gd, j, 64);
                S += __shfl(my_al, j, 64) * da;
                if (lane == j) my_da = da;
            }
            if (lane < ne) dlogit[eb + lane] = my_da;
        }
        if (coop) {                                              // the row's S: the four quarters in wave order
            if (lane == 0) sS[wv][0] = S;
            __syncthreads();
            S = ((sS[0][0] + sS[1][0]) + sS[2][0]) + sS[3][0];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // pass 2
        float s_lrh = 0.f, s_arh = 0.f;
        auto fetch_kc = [&](int cn) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                pa[s] = z4; pb[s] = z4;
                if (c < H) {
                    pa[s] = *(const float4*)(qkv + (size_t)cn * ldqkv + H + c);
                    pb[s] = *(const float4*)(cv + (size_t)cn * ldcv + c);
                }
            }
        };
        for (int eb = ws; eb < we; eb += 64) {
            const int ne = min(64, we - eb);
            int my_col = 0;
            float my_rh = 0.f, my_al = 0.f, my_cvs = 0.f, my_gd = 0.f, my_da = 0.f, o_dl = 0.f, o_dc = 0.f, o_drh = 0.f;
            if (lane < ne) {
                const int e = eb + lane;
                my_col = col[e]; my_rh = rhohat[e]; my_al = alpha[e]; my_cvs = cvs[e]; my_da = dlogit[e];
                my_gd = gx * d[(size_t)e * 3] + gy * d[(size_t)e * 3 + 1] + gz * d[(size_t)e * 3 + 2];
            }
            fetch_kc(__shfl(my_col, 0, 64));
            for (int j = 0; j < ne; ++j) {
                const float rh = __shfl(my_rh, j, 64), al = __shfl(my_al, j, 64);
                float4 kks[NS], ccs[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) { kks[s] = pa[s]; ccs[s] = pb[s]; }
                if (j + 1 < ne) fetch_kc(__shfl(my_col, j + 1, 64));
                const float dl = al * (__shfl(my_da, j, 64) - S);
                const float dc = al * __shfl(my_gd, j, 64);
                float dsum = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = s * 256 + lane * 4;
                    if (c < H) {
                        const float4 kk = kks[s], cc = ccs[s];
                        dq[s].x += dl * (kk.x + rh * wk[s].x); dq[s].y += dl * (kk.y + rh * wk[s].y);
                        dq[s].z += dl * (kk.z + rh * wk[s].z); dq[s].w += dl * (kk.w + rh * wk[s].w);
                        const float ux = cc.x + rh * wc[s].x, uy = cc.y + rh * wc[s].y, uz = cc.z + rh * wc[s].z, uw = cc.w + rh * wc[s].w;
                        const float px = w3v[s].x * dsilu_f(ux), py = w3v[s].y * dsilu_f(uy), pz = w3v[s].z * dsilu_f(uz), pw = w3v[s].w * dsilu_f(uw);
                        dsum += px * wc[s].x + py * wc[s].y + pz * wc[s].z + pw * wc[s].w;
                        a_wcr[s].x += dc * rh * px; a_wcr[s].y += dc * rh * py; a_wcr[s].z += dc * rh * pz; a_wcr[s].w += dc * rh * pw;
                        a_w3[s].x += dc * silu_f(ux); a_w3[s].y += dc * silu_f(uy); a_w3[s].z += dc * silu_f(uz); a_w3[s].w += dc * silu_f(uw);
                    }
                }
                dsum = wave_sum(dsum);
                s_lrh += dl * rh; s_arh += al * rh;
                if (lane == j) { o_dl = dl; o_dc = dc; o_drh = dl * qw + al * gwv + dc * dsum; }
            }
            if (lane < ne) {
                const int e = eb + lane;
                dlogit[e] = o_dl;
                dcp[e] = o_dc;
                drh[e] = o_drh;
                const float w = my_al * my_cvs;
                dd[(size_t)e * 3] = w * gx; dd[(size_t)e * 3 + 1] = w * gy; dd[(size_t)e * 3 + 2] = w * gz;
            }
        }
        if (!coop) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s * 256 + lane * 4;
                if (c < H) {
                    *(float4*)(dqkv + (size_t)r * ldqkv + c) = dq[s];
                    a_wrk[s].x += s_lrh * qv[s].x; a_wrk[s].y += s_lrh * qv[s].y; a_wrk[s].z += s_lrh * qv[s].z; a_wrk[s].w += s_lrh * qv[s].w;
                    a_wrv[s].x += s_arh * gh[s].x; a_wrv[s].y += s_arh * gh[s].y; a_wrv[s].z += s_arh * gh[s].z; a_wrv[s].w += s_arh * gh[s].w;
                }
            }
        } else {
            // dq and the two row sums: the four quarters in wave order; wave 0 folds the row sums into its vector-gradient accumulators
#pragma unroll
            for (int s = 0; s < NS; ++s) shp[wv * NS * 64 + s * 64 + lane] = dq[s];
            if (lane == 0) { sS[wv][1] = s_lrh; sS[wv][2] = s_arh; }
            __syncthreads();
            for (int g = threadIdx.x; g < NS * 64; g += 256) {
                const int c = (g >> 6) * 256 + (g & 63) * 4;
                if (c < H) {
                    float4 t = shp[g];
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const float4 o = shp[w * NS * 64 + g];
                        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                    }
                    *(float4*)(dqkv + (size_t)r * ldqkv + c) = t;
                }
            }
            if (wv == 0) {
                const float tl = ((sS[0][1] + sS[1][1]) + sS[2][1]) + sS[3][1], ta = ((sS[0][2] + sS[1][2]) + sS[2][2]) + sS[3][2];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a_wrk[s].x += tl * qv[s].x; a_wrk[s].y += tl * qv[s].y; a_wrk[s].z += tl * qv[s].z; a_wrk[s].w += tl * qv[s].w;
                    a_wrv[s].x += ta * gh[s].x; a_wrv[s].y += ta * gh[s].y; a_wrv[s].z += ta * gh[s].z; a_wrv[s].w += ta * gh[s].w;
                }
            }
            __syncthreads();                                     // the exchange area is rewritten by the next heavy row
        }
    }
    // block partials of the four vector gradients: [gridDim.x][4][H]
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        shp[(0 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wrk[s];
        shp[(1 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wrv[s];
        shp[(2 * 4 + wv) * NS * 64 + s * 64 + lane] = a_wcr[s];
        shp[(3 * 4 + wv) * NS * 64 + s * 64 + lane] = a_w3[s];
    }
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < 4 * NS * 64; i2 += 256) {
        const int qn = i2 / (NS * 64), rem = i2 % (NS * 64);
        const int c = (rem / 64) * 256 + (rem % 64) * 4;
        if (c >= H) continue;
        float4 t = shp[(qn * 4 + 0) * NS * 64 + rem];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const float4 o = shp[(qn * 4 + w2) * NS * 64 + rem];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *(float4*)(wpart + ((size_t)blockIdx.x * 4 + qn) * H + c) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, pass B (per node n: the gradients that arrive through its edges as SENDING node -- the inter graph is symmetric, so a gather
// over the node's own row through mirror[]: dk[n], dv[n], dcv[n]; dbias_red[pair] = dlogit[e] + dlogit[mirror]).  Independent gathers (no
// serial chain), so the split only spreads a heavy row's ~2 x 2 KB per edge over four waves; the partial sums meet through LDS in wave order.
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void inter_attn_bwd_b_rows_kernel(
    const float* __restrict__ qkv, int ldqkv, const float* __restrict__ cv, int ldcv, int H, const float* rhohat,
    const int* rowptr, const int* col, const int* mirror, const int* red_idx, const float* __restrict__ wcr,
    const float* __restrict__ w3, const float* alpha, const float* dlogit, const float* dcp, int n_rows,
    const float* __restrict__ dh_out, float* dqkv, float* dcv, float* dbias_red, const int* __restrict__ order, int n_heavy, int n_act) {
    extern __shared__ float4 shb[];   // [3][4][NS*64] float4: the waves' partial dk | dv | dc of a heavy row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool coop = (int)blockIdx.x < n_heavy;
    const int idx = coop ? (int)blockIdx.x : n_heavy + ((int)blockIdx.x - n_heavy) * 4 + wv;
    if (idx >= n_rows) return;
    const int n = order[idx];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dk[NS], dv[NS], dc[NS], cvn[NS], wc[NS], w3v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s * 256 + lane * 4;
        dk[s] = z4; dv[s] = z4; dc[s] = z4; cvn[s] = z4; wc[s] = z4; w3v[s] = z4;
        if (c < H && idx < n_act) {
            cvn[s] = *(const float4*)(cv + (size_t)n * ldcv + c);
            wc[s] = *(const float4*)(wcr + c); w3v[s] = *(const float4*)(w3 + c);
        }
    }
    if (idx < n_act) {
        const int e0 = rowptr[n], e1 = rowptr[n + 1];
        int ws = e0, we = e1;
        if (coop) {
            const int per = (e1 - e0 + 3) >> 2;
            ws = min(e1, e0 + wv * per);
            we = min(e1, ws + per);
        }
        for (int eb = ws; eb < we; eb += 64) {             // per-edge scalars (through mirror[]: dependent loads) one edge per lane, ahead of the loop
            const int ne = min(64, we - eb);
            int m_l = 0;
            float dl_l = 0.f, al_l = 0.f, dcm_l = 0.f, rh_l = 0.f;
            if (lane < ne) {
                const int e = eb + lane, me = mirror[e];   // edge me = (m <- n)
                m_l = col[e];
                dl_l = dlogit[me]; al_l = alpha[me]; dcm_l = dcp[me]; rh_l = rhohat[me];
                if (n < m_l) dbias_red[red_idx[e]] = dlogit[e] + dl_l;   // ligand rows precede their proteins
            }
#pragma unroll 2
            for (int j = 0; j < ne; ++j) {                 // independent gathers: two edges in flight
                const int m = __shfl(m_l, j, 64);
                const float dl = __shfl(dl_l, j, 64), al = __shfl(al_l, j, 64), dcm = __shfl(dcm_l, j, 64), rh = __shfl(rh_l, j, 64);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = s * 256 + lane * 4;
                    if (c < H) {
                        const float4 qm = *(const float4*)(qkv + (size_t)m * ldqkv + c);
                        const float4 gm = *(const float4*)(dh_out + (size_t)m * H + c);
                        dk[s].x += dl * qm.x; dk[s].y += dl * qm.y; dk[s].z += dl * qm.z; dk[s].w += dl * qm.w;
                        dv[s].x += al * gm.x; dv[s].y += al * gm.y; dv[s].z += al * gm.z; dv[s].w += al * gm.w;
                        dc[s].x += dcm * w3v[s].x * dsilu_f(cvn[s].x + rh * wc[s].x);
                        dc[s].y += dcm * w3v[s].y * dsilu_f(cvn[s].y + rh * wc[s].y);
                        dc[s].z += dcm * w3v[s].z * dsilu_f(cvn[s].z + rh * wc[s].z);
                        dc[s].w += dcm * w3v[s].w * dsilu_f(cvn[s].w + rh * wc[s].w);
                    }
                }
            }
        }
    }
    if (!coop) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s * 256 + lane * 4;
            if (c < H) {
                *(float4*)(dqkv + (size_t)n * ldqkv + H + c) = dk[s];
                *(float4*)(dqkv + (size_t)n * ldqkv + 2 * H + c) = dv[s];
                *(float4*)(dcv + (size_t)n * ldcv + c) = dc[s];
            }
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        shb[(0 * 4 + wv) * NS * 64 + s * 64 + lane] = dk[s];
        shb[(1 * 4 + wv) * NS * 64 + s * 64 + lane] = dv[s];
        shb[(2 * 4 + wv) * NS * 64 + s * 64 + lane] = dc[s];
    }
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < 3 * NS * 64; i2 += 256) {
        const int qn = i2 / (NS * 64), g = i2 % (NS * 64);
        const int c = (g >> 6) * 256 + (g & 63) * 4;
        if (c >= H) continue;
        float4 t = shb[(qn * 4 + 0) * NS * 64 + g];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = shb[(qn * 4 + w) * NS * 64 + g];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        if (qn == 0) *(float4*)(dqkv + (size_t)n * ldqkv + H + c) = t;
        else if (qn == 1) *(float4*)(dqkv + (size_t)n * ldqkv + 2 * H + c) = t;
        else *(float4*)(dcv + (size_t)n * ldcv + c) = t;
    }
}

extern "C" int fabind_inter_attn_bwd_rows(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* d,
                                          const float* rhohat, const int* rowptr, const int* col, const int* mirror,
                                          const int* red_idx, const float* w_rk, const float* w_rv, const float* wcr,
                                          const float* w3, const float* alpha, const float* cvs, float clampv, int n_rows,
                                          const float* dh_out, const float* dx_out, float* dqkv, float* dcv, float* dd,
                                          float* drh, float* dbias_red, float* dlogit, float* dcp, float* wpart, int nblk,
                                          const int* order, int n_heavy, int n_act, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    FB_REQUIRE(H % 4 == 0 && H <= 1024, "fabind_inter_attn_bwd_rows: H % 4, H <= 1024");
    FB_REQUIRE(order != nullptr && 0 <= n_heavy && n_heavy <= n_act && n_act <= n_rows && nblk > 0, "fabind_inter_attn_bwd_rows: order / counts");
    dim3 gb(n_heavy + (n_rows - n_heavy + 3) / 4);
#define LA(NS) do {                                                                                                    \
        size_t lds = (size_t)16 * NS * 64 * sizeof(float4);                                                            \
        hipLaunchKernelGGL((inter_attn_bwd_a_rows_kernel<NS>), dim3(nblk), dim3(256), lds, stream, qkv, ldqkv, cv, ldcv, H, d, \
                           rhohat, rowptr, col, w_rk, w_rv, wcr, w3, alpha, cvs, clampv, n_rows, dh_out, dx_out, dqkv, dd, \
                           drh, dlogit, dcp, wpart, order, n_heavy, n_act);                                             \
        size_t ldsb = (size_t)12 * NS * 64 * sizeof(float4);                                                           \
        hipLaunchKernelGGL((inter_attn_bwd_b_rows_kernel<NS>), gb, dim3(256), ldsb, stream, qkv, ldqkv, cv, ldcv, H, rhohat, rowptr, \
                           col, mirror, red_idx, wcr, w3, alpha, dlogit, dcp, n_rows, dh_out, dqkv, dcv, dbias_red, order,  \
                           n_heavy, n_act);                                                                            \
    } while (0)
    if (H <= 256) LA(1); else if (H <= 512) LA(2); else LA(4);
#undef LA
    FB_CHECK_LAUNCH();
    return 0;
}
