/* fabind_hip.h -- C ABI of libfabind_hip.so: the MI355X (gfx950) kernels behind the FABind docking hot path.
 *
 * The reference (QizhiPei/FABind) is 100 % Python; its "native layer" is ATen + torch_scatter +
 * torch_geometric ops called from fabind/models/*.py.  Each entry point below replaces the group of
 * those ops named in its comment (paths relative to the reference's FABind/fabind directory).  Plain pointers
 * and sizes only: every pointer is a DEVICE pointer unless stated otherwise, `stream` is a
 * hipStream_t, dtype codes are FB_DT_F32 = 0 / FB_DT_BF16 = 1 (raw bfloat16 bits), activation
 * codes FB_ACT_NONE/SILU/RELU/SIGMOID = 0..3 (4 = FB_ACT_STORED_DERIV: the aux operand already is the derivative).  Every function returns 0 on success, non-zero on
 * error (message via fabind_last_error()); none of them synchronises or allocates.
 *
 * Node layout contract: nodes are complex-contiguous in the fixed order
 * [glb_c, ligand atoms..., glb_p, protein residues...]; node_off[b] is the first node of complex b
 * (node_off[B] = N) and c_cnt[b] = Nc_b + 1 the size of the ligand-side block (global node included).
 */
#ifndef FABIND_HIP_H
#define FABIND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

const char* fabind_last_error(void);
/* Bumped whenever an entry point's signature or an argument struct's layout changes.  History: 1 = first round-1 layout;
 * 2 = FabindGemmArgs grew {p_drop, drop_seed, row_mu, row_rs, col_c}, FabindEdgeBwdArgs / FabindPairUpdateArgs added;
 * 3 = fabind_cross_attn_bwd takes a scratch buffer instead of (n_q_rows, n_k_rows), fabind_cross_attn_bwd_scratch added;
 * 4 = fabind_pair_bias_cat takes a colpart argument (per-work-group column sums), fabind_pair_bias_cat_parts and the
 *     fabind_gemm_tn_set_waves knob added.
 * 5 = FabindGemmArgs grew {C16, ldc16}; fabind_cross_attn_mfma_fwd / fabind_cross_attn_mfma_bwd added (MFMA bf16 form of the cross attention); FabindEdgeBwdArgs grew
 *     bnd + d2scratch, fabind_gcl_edge_fused takes bnd (deterministic boundary sums instead of float atomics), fabind_gcl_edge_fused_bwd_set_variant / _variant added.
 * 6 = bf16 result copies emitted by their producers (no cast kernels): fabind_segment_sum takes (out16, ldo16), fabind_gcl_edge_fused takes agg16, fabind_inter_attn_fwd takes h16,
 *     FabindEdgeBwdArgs grew {dAB16, lddab16}; fabind_gcl_edge_fused_set_variant / _variant (forward kernel form) added.
 * 7 = fabind_multi_copy + FabindCopySeg added (the parameter pack and its adjoint as one launch each).
 * 8 = fabind_split_sum added (split-contraction partials reduced straight into fp32 or bf16).
 * 9 = fabind_zero_empty_rows added (outputs of the fused edge kernels are no longer memset whole).
 * 10 = fabind_gemm_tn_tile_n added (the host sizes the split count of fabind_gemm_tn from the output tile of the current layout: 256 x 256 on
 *     eight waves by default); knobs fabind_gemm_tn_set_exp, fabind_gemm_set_small_m; fabind_gemm_tn_set_waves accepts 16 / 20.
 * 12: fabind_mul_dact / fabind_mul_dact_colsum take `scale` (ReLU + epilogue dropout under autograd);
 *     fabind_cross_attn_fused_fwd / fabind_pair_bo_pack (pair bias recomputed inside the attention kernel), fabind_pair_hadamard_bwd_grid;
 *     FabindGemmArgs.split3 (fp32 x fp32 contracted as split bf16, three MFMAs per product term: precision mode 'bf16x3');
 *     fabind_gcl_edge_fused_x3 (the fused forward edge pipeline in that arithmetic); knob fabind_gemm_set_x3_tile.
 * 11 = fabind_gemm_tn takes with_colsum (the bias gradient rides along with the weight gradient: no separate column-sum launches);
 *     fabind_split_sum takes (n_tail, out_tail); fabind_inter_attn_bwd writes wpart as [nblk][4][H] (was [4][nblk][H]);
 *     fabind_pair_hadamard_bwd_rows added (the pair-Hadamard adjoint over the inter graph without float atomics);
 *     fabind_layernorm_rows_bwd takes nblk (the caller sizes the partial buffers; the kernel strides rows over that grid).
 * 13 = FabindGemmArgs.r_dtype (bf16 residual operand).  14 = fabind_rows_hadamard_bwd, fabind_node_chain_fwd added.
 * 15 = fabind_gemm_tn_multi + FabindTnJob (the queued weight-gradient contractions of a backward pass as one launch + one reduction);
 *     fabind_sizeof_args(3) = sizeof(FabindTnJob); fabind_gcl_edge_fused_x3 takes n_rows (4 GiB bound of its 32-bit gather offsets checked);
 *     fabind_pair_bias_finish writes per-work-group partials of dw (no float atomics; the caller sums them with fabind_colsum);
 *     fabind_pack_frag_multi + FabindPackSeg (all fragment packs of a model call in one launch); fabind_lower_bound;
 *     fabind_cross_attn_fused_fwd takes lse; fabind_cross_attn_fused_bwd + FabindAttnFusedBwdArgs, fabind_pair_bot_pack (fused backward of the
 *     cross attention: fabind_sizeof_args(4)).
 * 16 = fabind_gcl_edge_fused_train (the forward that saves M / silu'(pre2) / pre3) and FabindEdgeBwdArgs.{d2f, z3f} (the two-contraction backward).
 * 17 = round-5 retirement of knob-only kernels: fabind_gemm_set_big removed (256 x 256 GEMM tiles); fabind_gcl_edge_fused_set_variant accepts 1 only,
 *     fabind_gcl_edge_fused_bwd_set_variant 0 / 5 only, fabind_gcl_edge_fused_bwd_set_tile 64 only (their other kernels are no longer built);
 *     fabind_gcl_edge_fused_x3_train added (the split-bf16 forward that saves M / silu'(pre2) / pre3 for the two-contraction backward);
 *     fabind_inter_attn_fwd_rows / fabind_inter_attn_bwd_rows added (inter-edge attention with the rows dealt by degree: heavy rows on four waves);
 *     fabind_edge_lnfold_bwd (+ _blocks) added (FABind+: the LayerNorm-folded first edge Linear under autograd);
 *     fabind_mul_dropmask_colsum, fabind_drop_mix (+ _bwd) added, fabind_gemm's fp32 (+ residual) epilogue takes p_drop (train-mode dropout
 *     ahead of a residual).
 * 18 = round 6: fabind_loss_fwd / fabind_loss_bwd / fabind_loss_blocks (the six-term training loss and its gradient seeds, one launch each way),
 *     fabind_pocket_center_fwd / _bwd (the Gumbel-softmax pocket centre, one launch each way); fabind_gemm_tn_set_exp ignores bit 2;
 *     fabind_pair_dist_fwd / _bwd, fabind_block_hadamard_bwd (+ fabind_pair_block_tile / _chunk): the distance head's pair operations on block descriptors;
 *     FabindGemmArgs.c2_bf16; fabind_node_chain_x3_fwd (node MLP / Transition as one split-precision kernel).
 * A binding must refuse a library whose version differs from the header it was written against. */
#define FABIND_ABI_VERSION 18
int fabind_abi_version(void);
/* sizeof() of an argument struct as this library was compiled: which = 0 FabindGemmArgs, 1 FabindEdgeBwdArgs,
 * 2 FabindPairUpdateArgs, 3 FabindTnJob (-1 for an unknown index).  Lets a foreign-language mirror of the struct verify its layout. */
int fabind_sizeof_args(int which);

/* ---------------------------------------------------------------------------------------------
 * Dense contractions.  Replaces every nn.Linear / torch.cat+Linear / einsum+Linear on the path:
 * models/egnn.py:40-60,81,104,119,160-170,197-208; models/model_utils.py:83-92,103-131,168-174,
 * 196-222; models/model.py:100-101,117,132,139,358.
 *   C[M,N] = epi( pro(A|A2)[M,K] * W[N,K]^T )
 *   pro: act_pro applied to A while staging; columns [0,K1) come from A, [K1,K) from A2.
 *   epi: v = alpha*acc + bias[n]; v = act_epi(v); v *= d(act)/dx (dact_epi) at aux[m,n];
 *        v += R[r_index ? r_index[m] : m, n]; v += C_old (accumulate); C = v;
 *        dot_out[m, n_tile] = sum_n v * dotvec[n]   (n_tile = column block of 128)
 *   groups != NULL: ragged batched mode, blockIdx.z picks int32[8] {a_row0, M, w_row0, N, c_off_lo,
 *        c_off_hi, ldc, 0}; bias is indexed by w_row0 + n.
 * -------------------------------------------------------------------------------------------*/
typedef struct FabindGemmArgs {
    const void* A;
    const void* A2;
    const void* W;
    void* C;
    const float* bias;
    const void* R;
    const int* r_index;
    const float* dotvec;
    float* dot_out;
    const void* aux;
    const int* groups;
    void* C2; /* optional second output: act_epi'(pre-activation), same dtype/ld as C (saved for backward) */
    int M, N, K, K1;
    int lda, lda2, ldw, ldc, ldr, ldaux, dot_ld;
    int a_dtype, w_dtype, c_dtype, aux_dtype;
    int act_pro, act_epi, dact_epi;
    int accumulate;
    int store_preact; /* C receives the value BEFORE act_epi (row-dot still sees act_epi(v)) */
    int n_groups, max_m, max_n;
    int groups_ext; /* groups are int32[16] {M, N, K, lda, a_off lo/hi, w_off lo/hi, c_off lo/hi, ldc}: ragged A/W/C */
    int epi_fast; /* set by fabind_gemm itself: index of a specialised epilogue (0 = generic) */
    int k_splits; /* >1: split the K loop over k_splits work-groups; C must be fp32 [k_splits, M, N] partials */
    float alpha;
    float p_drop;       /* > 0: epilogue dropout (train / sampling mode): after bias + activation, before the residual, the value is
                           multiplied by keep/(1-p) with keep = [hash32(drop_seed + row*N + col) & 0xffff >= round(p*65536)] */
    unsigned drop_seed;
    /* optional LayerNorm fold (act_epi = relu writing a bf16 C, or relu + row-dot only; bf16 operands; K % 64 == 0): the accumulator of row r, column c becomes
       row_rs[r] * (acc - row_mu[r] * col_c[c]) before bias / activation -- i.e. LN(A) W^T evaluated as a GEMM on the
       UN-normalised A with W pre-scaled by the LN weight, col_c = its row sums, bias = W ln_b + b */
    const float* row_mu; const float* row_rs; const float* col_c;
    /* optional bf16 copy of an fp32 C (plain fp32 outputs with bias and/or residual only -- the residual-stream Linears): the
       epilogue writes C16[m, n] = bf16(C[m, n]) next to C, so the GEMMs that consume the result as an MFMA operand need no
       separate conversion pass over it */
    void* C16; int ldc16;
    /* != 0 with fp32 A and fp32 W: the contraction runs as split bf16 ("bf16x3": every operand element x = hi + lo with
       hi = bf16(x), lo = bf16(x - hi); three bf16 MFMAs per product term, fp32 accumulate; ~2^-17 relative operand error) instead
       of the exact fp32 MFMA.  The reference contracts in fp32 (egnn.py:68-144, model_utils.py:83-131). */
    int split3;
    /* dtype of R (FB_DT_F32 = 0: the default; FB_DT_BF16: a bf16 residual stream -- the FABind+ pair embedding, cross_att.py:41-46 --
       is added without a converted copy; row-contiguous 16-byte reads when C is bf16 with no activation, no r_index) */
    int r_dtype;
    int c2_bf16;       /* round 6: != 0: C2 (the stored activation derivative) is written as bf16 although C is fp32 (fp32 hidden layer for a
                          split-precision second Linear, bf16 derivative for the unchanged backward); C16 then carries bf16(act(pre)) */
} FabindGemmArgs;

int fabind_gemm(const FabindGemmArgs* args, hipStream_t stream);

/* Weight-gradient contraction without transposes: C_part[s][m][n] = sum_{e in split s} Y[e,m] * X[e,n]
 * (bf16 row-major operands, fp32 partials [splits, M, N]; zero_page = >= 16 zero bytes in device memory).
 * with_colsum != 0 (no groups): every split's partial is [M * N + M] floats, the last M being sum_{e in split s} Y[e,m] -- the bias
 * gradient dY^T 1 of the Linear whose weight gradient dY^T X this is (reference: autograd of nn.Linear, e.g. model_utils.py:162-175),
 * so that ONE fabind_split_sum over M * N + M elements finishes both. */
int fabind_gemm_tn(const void* Y, int ldy, const void* X, int ldx, float* C_part, int M, int N, int E, int splits,
                   const void* zero_page, const int* groups, int n_groups, int with_colsum, hipStream_t stream);
/* groups (optional, splits == 1): int32[8] per group {y_off lo/hi (elements), ldy, M, x_row0, E, c_off lo/hi}. */

/* MANY weight-gradient contractions as ONE launch + ONE reduction launch.  Replaces the per-Linear `grad_weight = grad_output^T input`,
 * `grad_bias = grad_output^T 1` of torch autograd (nn.Linear at egnn.py:40-60,81,104,119,197-208, model_utils.py:83-131,162-175) for a
 * whole backward pass: the host queues the contractions as their operands become available and flushes the queue before the first
 * consumer of a result.  On a pocket-sized batch (9,000 nodes) one contraction is 32 work-groups of the 256-CU chip and the step has 89 of
 * them; together they fill it.
 * A job: out[m, n] (out_dt, row stride ldo) = sum_e Y[e, m] X[e, n]  and, with_colsum != 0, out_tail[m] (fp32) = sum_e Y[e, m]; bf16
 * row-major operands; C_part = this job's scratch of splits * (M * N + (with_colsum ? M : 0)) floats.  The contraction kernel is
 * fabind_gemm_tn's 256 x 256 layout: job j owns work-groups [wg0, wg0 + n_wg) with n_wg = n_tiles * roundup8(splits), wg0 a multiple of
 * 8 (the XCD-aware order of the single launch holds inside every job), n_tiles = ceil(M / 256) * ceil(N / 256),
 * e_per = roundup32(ceil(E / splits)); the reduction kernel gives job j blocks [blk0, blk0 + n_blk), n_blk = ceil((M * N (+ M)) / 4096).
 * jobs_dev / wg_job_dev (int32 job index per work-group) / blk_job_dev (per reduction block) are DEVICE tables, jobs_host the HOST copy
 * of the same jobs (validated here; the device cannot be read without synchronising). */
typedef struct FabindTnJob {
    const void* Y; const void* X; float* C_part; void* out; float* out_tail;
    int ldy, ldx, M, N, E, splits, e_per, n_tiles, with_colsum, out_dt, ldo;
    int wg0, n_wg, blk0, n_blk, pad_;
} FabindTnJob;
int fabind_gemm_tn_multi(const FabindTnJob* jobs_dev, const FabindTnJob* jobs_host, int n_jobs, const int* wg_job_dev, int total_wg,
                         const int* blk_job_dev, int total_blk, const void* zero_page, hipStream_t stream);

/* out[C,R] = act(in[R,C])^T -- feeds weight-gradient contractions (autograd of the ops above). */
int fabind_transpose_act(const void* in, int in_dt, int ldi, void* out, int out_dt, int ldo, int R, int C, int act,
                         hipStream_t stream);
/* Many strided 2-D copies with dtype conversion in one launch (the differentiable parameter pack of a model call -- slices,
 * concatenations, zero padding and casts of nn.Module parameters, models/*.py __init__ layouts -> kernel operand layouts -- and its
 * adjoint).  The table lives in DEVICE memory; strides are in elements; src == NULL writes zeros; vec4 != 0 promises column-contiguous,
 * 4-element-aligned rows on both sides with cols % 4 == 0. */
typedef struct FabindCopySeg {
    const void* src;
    void* dst;
    long long src_sr, src_sc, dst_sr, dst_sc;
    int rows, cols;
    int src_dt, dst_dt;
    int vec4, pad_;
} FabindCopySeg;
int fabind_multi_copy(const FabindCopySeg* segs_dev, int n_segs, int blocks_per_seg, hipStream_t stream);
/* (round 6: src_dt = 2 = an fp32 source whose LO plane is packed, bf16(w - bf16(w)): with src_dt = 0 of the same source the hi | lo
 * fragment packs of the split-precision kernels.)
 * Many [N, K] weights (fp32 or bf16; element (n, k) at src[n * src_sr + k * src_sc], so a transposed source is a stride swap) into the
 * bf16 MFMA-fragment order [K/32][N/16][4][16][8] the fused edge / pair kernels take (fabind_gcl_edge_fused: W2p, Wcp; FabindEdgeBwdArgs:
 * W2p, Wcp, W2Tp, WcTp), one launch for all.  segs_dev: the table in DEVICE memory, segs_host: the same table on the host (validated). */
typedef struct FabindPackSeg {
    const void* src;
    void* dst;
    long long src_sr, src_sc;
    int N, K, src_dt, pad_;
} FabindPackSeg;
int fabind_pack_frag_multi(const FabindPackSeg* segs_dev, const FabindPackSeg* segs_host, int n_segs, hipStream_t stream);
/* out[i] = sum_s part[s * n + i] (fixed order), written as out_dt: the split reduction of fabind_gemm_tn / split-K partials. */
int fabind_split_sum(const float* part, int splits, long n, void* out, int out_dt, long n_tail, float* out_tail, hipStream_t stream);
/* n_tail > 0: the LAST n_tail of the n elements are written to out_tail as fp32 instead (a bias gradient behind a bf16 weight gradient). */
/* out[c] (+)= sum_r in[r,c]; scratch = float[nchunk*C] (bias gradients, deterministic two-pass). */
int fabind_colsum(const void* in, int in_dt, int ldi, float* out, int R, int C, int accumulate, float* scratch,
                  int nchunk, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Edge construction.  Replaces ComplexGraph.construct_edges + _radial_edges (models/att_model.py:
 * 37-128: dense [N,max_n] mask + nonzero) with direct row-sorted CSR emission.
 *   ctx graph  (row aggregates): ligand bonds (as given), protein-protein pairs with |xu-xv| <= cut_intra
 *              (normalised units), star edges to the segment's global node, glb_c<->glb_p.
 *   inter graph: ligand-atom <-> protein-residue pairs with |xu-xv| <= cut_inter, both directions.
 * count: fills deg_ctx[N], deg_int[N].  The caller scans them into rowptr (fabind_exclusive_scan).
 * fill : writes col / row (COO row per edge) arrays for both graphs.
 * bond_row/bond_col: ligand bonds (global node ids), complex-contiguous; bond_off[B+1] their ranges.
 * -------------------------------------------------------------------------------------------*/
int fabind_edges_count(const float* x, const int* node_off, const int* c_cnt, int B, int max_n, const int* bond_row,
                       const int* bond_off, float cut_intra, float cut_inter, int* deg_ctx, int* deg_int,
                       hipStream_t stream);
int fabind_edges_fill(const float* x, const int* node_off, const int* c_cnt, int B, int max_n, const int* bond_row,
                      const int* bond_col, const int* bond_off, float cut_intra, float cut_inter,
                      const int* rowptr_ctx, const int* rowptr_int, int* col_ctx, int* row_ctx, int* col_int,
                      int* row_int, hipStream_t stream);
/* out[0]=0, out[i+1]=out[i]+in[i], i<n (single work-group scan; n up to a few million). */
int fabind_exclusive_scan(const int* in, int* out, int n, hipStream_t stream);
/* out[k] = first index i with vals[i] >= keys[k] (n if none), vals non-decreasing int32 [n]: per-complex [start, end) offsets of a
 * complex-contiguous edge / bond / LAS list from node_off (keys = node_off [B + 1] gives out [B + 1]).  Replaces the bucketize +
 * histogram + cumsum index glue around att_model.py:37-116 / egnn.py:433-449's per-sample loops. */
int fabind_lower_bound(const int* vals, int n, const int* keys, int n_keys, int* out, hipStream_t stream);
/* Inter-edge pair bookkeeping (replaces trio_encoder's index arithmetic, models/egnn.py:286-304):
 * red_off[B+1] = prefix of ligand-row inter edges per complex; for every inter edge e, red_idx[e] =
 * index of its unordered (ligand, protein) pair in the reduced list; red_c/red_p = the pair's nodes;
 * mirror[e] = index of the reversed edge (the inter graph is symmetric). */
int fabind_inter_meta(const int* node_off, const int* c_cnt, int B, const int* rowptr_int, const int* col_int,
                      const int* row_int, int E_int, int* red_off, int* red_idx, int* red_c, int* red_p, int* mirror,
                      hipStream_t stream);
/* If the batch has no inter edge the reference inserts one fake symmetric pair (att_model.py:85-86);
 * host-side logic handles that case by calling fill with a forced pair -- see fabind_amd/graph.py. */

/* coord2radial with norm_type='per_sample' (models/egnn.py:767-787): d[e]=x[row]-x[col],
 * rho=|d|^2, rhohat = rho / sqrt(sum_{e in complex} rho^2).  Edges of a complex are contiguous:
 * [rowptr[node_off[b]], rowptr[node_off[b+1]]).  Also writes rho (un-normalised) and norm[b]. */
int fabind_edge_geom(const float* x, const int* row, const int* col, const int* rowptr, const int* node_off, int B,
                     float* d, float* rho, float* rhohat, float* norm, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MC_E_GCL pieces (models/egnn.py:68-144).
 * gcl_pre:  pre[e,:] = act( AB[row[e], 0:H] + AB[col[e], H:2H] + rhohat[e]*w_r[:] )   (act = NONE keeps the
 *           pre-activation for training; SILU is the forward-only form)
 *           (first edge_mlp Linear split column-wise: node-level projections gathered per edge).
 * segment_sum: out[r,:] = sum_{e in row r} act(Z[eidx ? eidx[e] : e,:])   (unsorted_segment_sum on a row-sorted
 *           CSR; eidx = permutation for reductions over the column index, used by backward passes)
 *           n_edges = rowptr[n_rows] (host copy).  bf16 rows are reduced edge-balanced (64 CSR positions per wave,
 *           rows cut by a chunk boundary finish with fp32 atomics); fp32 rows one wave per row, fixed order.
 * coord_mean: x_out[r] = x[r] + clamp( (1/max(deg,1)) * sum_e d[e]*s[e], +-clampv ),
 *           s[e] = sum_k s_part[e,k] (row-dot partials written by fabind_gemm).
 * -------------------------------------------------------------------------------------------*/
int fabind_gcl_pre(const void* AB, int ab_dt, int ldab, int H, const int* row, const int* col, const float* rhohat,
                   const float* w_r, void* pre, int pre_dt, int E, int act, void* dact_out, hipStream_t stream);
int fabind_segment_sum(const void* Z, int z_dt, int ldz, int H, const int* rowptr, const int* eidx, int n_rows, int n_edges,
                       int act, float* out, int ldo,
                       void* out16 /* NULL, or a bf16 copy of the result [n_rows, ldo16]; `out` may then be NULL */, int ldo16,
                       hipStream_t stream);
/* out[r, 0:C) (and out2[r, 0:C), if given) = 0 for every row r of a CSR with no entries (rowptr[r] == rowptr[r+1]): the rows the fused
 * edge kernels' segment sums (egnn.py:84,790-821 unsorted_segment_sum) do not write. */
int fabind_zero_empty_rows(const int* rowptr, int n_rows, void* out, int out_dt, int ld, int C, void* out2, int out2_dt, int ld2,
                           hipStream_t stream);
int fabind_coord_update(const float* x, const float* d, const float* s_part, int n_part, const float* weight,
                        const int* rowptr, int n_rows, int mean, float clampv, float* x_out, float* s_out,
                        hipStream_t stream);

/* The same fused forward edge pipeline (models/egnn.py:68-128) in SPLIT-bf16 arithmetic (precision mode 'bf16x3': fp32-grade results
 * on the bf16 matrix cores, three MFMAs per product term): AB = fp32 [N, 2H]; every packed weight is given as TWO fragment arrays in
 * the layout of fabind_gcl_edge_fused, W..ph = bf16(W) and W..pl = bf16(W - float(W..ph)); agg / s_out / bnd / dropout as there.
 * n_rows = rows of AB (and of agg): n_rows * ldab * 4 bytes must stay below 4 GiB (32-bit gather offsets). */
int fabind_gcl_edge_fused_x3(const float* AB, int ldab, int n_rows, int H, const int* row, const int* col, const float* rhohat,
                             const float* w_r, const void* W2ph, const void* W2pl, const float* b2, const void* Wcph,
                             const void* Wcpl, const float* bc, const float* w3, int E, float* agg, float* s_out, float p_drop,
                             unsigned seed, float* bnd, hipStream_t stream);
/* The same for a 'bf16x3' training step (round 5): additionally leaves M [E, H] bf16 row-major (= bf16 of the split-precision messages),
 * d2f / z3f = silu'(pre2) x keep and pre3 as bf16 in accumulator-fragment order -- exactly what fabind_gcl_edge_fused_train leaves, so the
 * two-contraction backward (FabindEdgeBwdArgs.d2f / z3f, bf16 copy of AB) serves this mode too (reference arithmetic: egnn.py:68-128). */
int fabind_gcl_edge_fused_x3_train(const float* AB, int ldab, int n_rows, int H, const int* row, const int* col, const float* rhohat,
                                   const float* w_r, const void* W2ph, const void* W2pl, const float* b2, const void* Wcph,
                                   const void* Wcpl, const float* bc, const float* w3, int E, float* agg, float* s_out, float p_drop,
                                   unsigned seed, float* bnd, void* M, void* d2f, void* z3f, hipStream_t stream);

/* Fused forward edge pipeline of MC_E_GCL (models/egnn.py:68-128) for 64-edge tiles, bf16:
 *   s_out[e] = w3 . silu( silu( silu(A[row]+Bc[col]+rhohat*w_r) W2^T + b2 ) Wc^T + bc ),  agg[row] += silu(.. W2^T + b2)
 * AB = bf16 [N, 2H] (A | Bc); W2p / Wcp = bf16 weights packed in MFMA fragment order [H/32][H/16][4][16][8];
 * agg must be zero-initialised (nodes without edges are not written); every node with edges gets ONE plain store -- runs that
 * span 64-edge tiles are added in tile order by a fix-up kernel through `bnd`, so the sum is deterministic.  H in {64,128,256,512}.
 * p_drop > 0 (train mode, egnn.py:82): the messages are multiplied by keep(e,c)/(1-p) before both consumers, with
 *   keep(e,c) = [ half_{e&1}( hash32(seed + (e>>1)*H + c) ) >= round(p*65536) ],
 *   hash32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16   (evaluated again by the backward). */
int fabind_gcl_edge_fused(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                          const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                          const float* w3, int E, float* agg, float* s_out, float p_drop, unsigned seed,
                          float* bnd /* scratch: ceil(E/64) x 2 x H floats (boundary runs of nodes that span tiles) */,
                          void* agg16 /* NULL, or [n_rows, H] bf16: the aggregated messages as the next contraction's operand */,
                          hipStream_t stream);

/* The same forward for a training step: additionally leaves M [E, H] bf16 row-major (the messages after dropout: operand of
 * d Wc = dT^T M), and d2f / z3f = silu'(pre2) x keep and pre3 = M Wc^T + bc as bf16 in the kernel's accumulator-fragment order
 * (ceil(E/64) x 64 x H elements each; opaque to the caller, consumed by fabind_gcl_edge_fused_bwd through FabindEdgeBwdArgs.d2f / z3f),
 * so that the backward chains two H x H contractions per edge instead of four (no forward recompute). */
int fabind_gcl_edge_fused_train(const void* AB, int ldab, int H, const int* row, const int* col, const float* rhohat,
                                const float* w_r, const void* W2p, const float* b2, const void* Wcp, const float* bc,
                                const float* w3, int E, float* agg, float* s_out, float p_drop, unsigned seed, float* bnd,
                                void* agg16, void* M, void* d2f, void* z3f, hipStream_t stream);

/* Backward of the fused edge pipeline (training, bf16): recomputes the forward per 64-edge tile and chains the four
 * H x H contractions of the adjoint (autograd of models/egnn.py:68-128).  Inputs: the forward's operands, ds [E]
 * (gradient of s), dagg [n_rows, lddagg] fp32 (gradient of agg); W2p/Wcp = fragment-packed W2 / Wc, W2Tp/WcTp =
 * fragment-packed W2^T / Wc^T.  Outputs: S1, Mm, dT, dP2 [E,H] bf16 = operands of the weight gradients
 * (dW2 = dP2^T S1, dWc = dT^T Mm via fabind_gemm_tn), dP1 [E,H] bf16 (gradient at the first Linear's output; the
 * caller reduces it over the sending node), drh [E], dABrow [n_rows, lddab] fp32 (zero-initialised by the caller:
 * receiving-side segment sum of dP1) and part [n_groups][4][H] fp32 partial column sums {d b2, d bc, d w3, d w_r}
 * (the caller adds the n_groups rows).  n_groups = number of persistent work-groups (<= number of CUs). */
typedef struct FabindEdgeBwdArgs {
    const void* AB; const int* row; const int* col; const float* rhohat; const float* w_r;
    const void* W2p; const void* Wcp; const void* W2Tp; const void* WcTp;
    const float* b2; const float* bc; const float* w3; const float* ds; const float* dagg;
    void* S1; void* Mm; void* dT; void* dP2; void* dP1;
    float* drh; float* dABrow; float* part;
    void* dbg;               /* NULL, or 12 x int64: cycle counts per phase of work-group 0 (profiling aid; two-tile variant) */
    float* bnd;              /* ceil(E / tile) x 2 x H floats of scratch (boundary runs of dABrow, added in tile order: deterministic) */
    void* d2scratch;         /* single-tile variant: n_groups x 64 x H x 2 bytes of scratch (silu'(pre2) of each work-group's tile) */
    int ldab, lddagg, lddab, E;
    float p_drop;            /* dropout probability of the messages (egnn.py:82); 0 = eval */
    unsigned seed;           /* the seed the forward call used */
    int xcd_aware;           /* filled in by the library (fabind_gcl_edge_fused_set_xcd_aware); callers pass 0 */
    int lddab16;
    void* dAB16;             /* NULL, or [N, lddab16] bf16: the receiving-side sums go to its columns [0, H) INSTEAD of dABrow (variant 5
                              * only; the caller reduces dP1 into columns [H, 2H) with fabind_segment_sum's out16) */
    const void* d2f;         /* NULL, or silu'(pre2) as saved by fabind_gcl_edge_fused_train (bf16, ceil(E/64) x 64 x H, fragment order) and */
    const void* z3f;         /* pre3 likewise: the launch then takes the TWO-contraction kernel (csrc/fused_edge_bwd4.hip), which reads these
                              * instead of recomputing them, does NOT write Mm (the forward's M is the weight gradient's operand) and needs
                              * neither d2scratch nor dbg; n_groups up to 2 x CUs at H = 512 (64 KiB LDS tile, <= 128 VGPRs) */
} FabindEdgeBwdArgs;
int fabind_gcl_edge_fused_bwd(const FabindEdgeBwdArgs* args, int H, int n_groups, hipStream_t stream);
/* Edges per tile of the backward kernel: 64 (one work-group per CU, default) or 32 (two per CU). */
int fabind_gcl_edge_fused_bwd_set_tile(int bm);
int fabind_gcl_edge_fused_bwd_tile(void);
/* Kernel variant: 0 (default) = two [64][H] LDS tiles, one work-group per CU; 1 = ONE LDS tile rewritten in place by every stage,
 * two work-groups per CU so that one work-group's elementwise epilogue can overlap the other's contraction (needs
 * FabindEdgeBwdArgs.d2scratch; n_groups up to 2 x CUs) -- measured no faster, kept as a knob.  Both
 * evaluate the same arithmetic in the same order per element (outputs equal to 1e-7). */
int fabind_gcl_edge_fused_bwd_set_variant(int v);
int fabind_gcl_edge_fused_bwd_variant(void);
int fabind_gcl_edge_fused_bwd_variant_for(int H);   /* the variant a launch at hidden size H takes: the set one, or -- when none was set explicitly -- 0 for H <= 128 (measured faster there), 5 otherwise.  The caller sizes d2scratch / n_groups from THIS value */
/* development knob of variants 3 / 4: bit mask of work the kernel SKIPS (results are then wrong -- sensitivity timing only). */
void fabind_gcl_edge_fused_bwd3_set_exp(int mask);
void fabind_gcl_edge_fused_bwd4_set_exp(int mask);   /* the saved-forward kernel: 1 dT / dP2 copy-outs, 2 S1 / dP1 stores, 4 row scan, 8 both contractions */
/* development knob: 1 (default) = XCD-aware tile order in the fused edge kernels (XCD x owns the x-th eighth of the tiles, i.e.
 * whole complexes); 0 = linear tile order (forward) / grid-stride walk (backward).  Results do not depend on it. */
void fabind_gcl_edge_fused_set_xcd_aware(int on);
/* Forward kernel variant: 1 (default) = row-wise gather + operand-swapped epilogues (csrc/fused_edge_fwd2.hip), 0 = round 1's kernel. */
int fabind_gcl_edge_fused_set_variant(int v);
int fabind_gcl_edge_fused_variant(void);

/* Fused pair-embedding update of FABind+'s CrossAttentionModule (FABind_plus/fabind/models/cross_att.py:42-44 with
 * model_utils.py InteractionModule / MLPwithLastAct), bf16, inference (no adjoint: training runs the separate launches):
 *   z1 = z_in + (T[p_node, 0:32] * T[c_node, b_off:b_off+32]) Wo^T + bo;   y = LayerNorm(z1; ln_w, ln_b, eps)
 *   z_out = drop(relu( drop(relu(y W1^T + b1)) W2^T + b2 ));   bias_out = z_out Wb^T + bb   (when Wbp != NULL)
 * z_in / z_out: bf16 [n_pairs, H] (may alias); T: fp32 [nodes, ldt]; p_node / c_node: int32 [n_pairs].
 * Wop = [H,32], W1p / W2p = [H,H], Wbp = [16,H], all bf16 packed in MFMA fragment order [K/32][N/16][4][16][8].
 * Dropout (sampling mode): keep(e,c) of fabind_gcl_edge_fused with seed (first) and seed + 0x9e3779b9 (second). */
typedef struct FabindPairUpdateArgs {
    const float* T; const int* p_node; const int* c_node;
    const void* z_in; void* z_out;
    const void* Wop; const float* bo; const float* ln_w; const float* ln_b;
    const void* W1p; const float* b1; const void* W2p; const float* b2;
    const void* Wbp; const float* bb; float* bias_out;
    int ldt, b_off, n_pairs;
    float eps, p_drop;
    unsigned seed;
} FabindPairUpdateArgs;
int fabind_pair_update_fused(const FabindPairUpdateArgs* args, int H, hipStream_t stream);

/* Ligand post-optimisation (FABind/fabind/utils/post_optim_utils.py:9-64; called per complex by fabind_inference.py:285-328):
 * `steps` Adam iterations (lr, betas 0.9 / 0.999, eps 1e-8) on x0 against
 *   sum_{(i,j) in LAS} | |x_i-x_j| - |ref_i-ref_j| | + 2 sum_{i,j} relu(1.22 - |x_i-x_j|)      (all_pairs = 0), or
 *   sum_{i,j} | |x_i-x_j| - |ref_i-ref_j| |                                                  (all_pairs = 1: no LAS edges)
 * for a BATCH of ligands in one launch (one work-group per ligand, all iterations inside the kernel).
 * x0, ref, x_out: fp32 [sum atoms, 3]; atom_off: int32 [n_ligands+1]; nbr_ptr int32 [sum atoms + 1] / nbr_idx int32 (GLOBAL
 * atom ids): for every DISTINCT directed LAS edge (i,j) the list of i holds j and the list of j holds i.
 * loss_out[l] = loss at the last iteration (before its step), rmsd_out[l] = sqrt(mean |ref - x_out|^2), like the reference's
 * return values.  max_atoms = largest ligand (<= 512). */
int fabind_post_optimize(const float* x0, const float* ref, const int* atom_off, const int* nbr_ptr, const int* nbr_idx,
                         int n_ligands, int max_atoms, int all_pairs, int steps, float lr, float* x_out, float* loss_out,
                         float* rmsd_out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Cross attention (RowAttentionBlock / Attention._attention, models/cross_att.py:118-134,
 * models/model_utils.py:21-38,96-133).  Ragged: no padding, so the -1e9 mask bias never applies.
 *   for complex b, query i, key j, head h (4 heads x 32):
 *     s = scale * q[i,h,:].k[j,h,:] + lin*sigmoid(gate)   with (lin,gate) read from
 *         bias[(pair_off[b] + i*sq[b] + j*sk[b]) * bias_ld + {lin_col, gate_col} + h]
 *     out[i,h,:] = sigmoid(gpre[i,h,:]) * sum_j softmax_j(s) v[j,h,:]
 * desc: int32[B][8] = {q_off, nq, k_off, nk, pair_off_lo, pair_off_hi, sq, sk}.
 * ksplit > 1 splits the keys over work-groups (few queries, many keys: the ligand-query block); `part` =
 * float[n_rows * ksplit * 4 * 34] scratch, merged by a second kernel (out/lse rows are the first n_rows of q).
 * -------------------------------------------------------------------------------------------*/
int fabind_cross_attn_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* gpre,
                          int ldg, const float* bias, int bias_ld, int lin_col, int gate_col, const int* desc, int B,
                          int max_nq, float scale, float* out, int ldo, float* lse, int ksplit, float* part, int n_rows,
                          hipStream_t stream);

/* Cross attention with the pair bias recomputed in the kernel (v1 stack; csrc/attn_mfma.hip, "fused" section): a work-group owns 64
 * protein rows of a complex, contracts their a0 rows (bf16 [rows, lda0], node-indexed) with the complex's packed ligand-side operand
 * `bo` (fabind_pair_bo_pack) on the matrix cores, forms bias = (lin + c) * sigmoid(gate + c') in registers and runs the attention of
 * fabind_cross_attn_mfma_fwd with it.  desc = int32[8] per complex {first protein row (node index), P, first ligand row (compact),
 * C, ...}; mode 0: queries = protein rows (q / gpre / out node-indexed, k / v compact ligand rows); mode 1: queries = ligand atoms
 * (q / gpre / out compact, k / v node-indexed), `part` = float[n_lig_rows * ceil(max_P / 64) * 4 * 34] scratch.  bconst = the eight
 * constants (lin0..3, gate0..3).  max_C <= 62.  Replaces RowAttentionBlock.forward incl. its pair-bias Linears (cross_att.py:118-134). */
/* lse (optional): receives the log-sum-exp of every (query row, head) -- what fabind_cross_attn_fused_bwd replays the weights from. */
int fabind_cross_attn_fused_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* gpre, int ldg,
                                const void* a0, int lda0, const void* bo, const int* toff, const float* bconst, const int* desc, int B,
                                int max_P, int max_C, int H, int mode, float scale, float* out, int ldo, float* part, int n_lig_rows,
                                float* lse, hipStream_t stream);
/* bo[(toff[b] * H/32 + ks * T_b + tile) * 512 + lane * 8 + e] = b0[c_index[lig0_b + atom], k] * wcomp[slot_row, k] in MFMA fragment
 * order (tile = 16 columns = 2 atoms x {lin0, gate0, lin1, gate1, lin2, gate2, lin3, gate3}; T_b = ceil(C_b / 2); toff = prefix sum of
 * T_b; tile_b[t] = complex of global tile t); wcomp = fp32 [8, H] rows lin0..3, gate0..3 (the composed weights W_{lin|gate} W_o). */
int fabind_pair_bo_pack(const float* b0, int ldb, const float* wcomp, int H, const int* c_index, const int* desc, const int* toff,
                        const int* tile_b, int n_tiles, void* out, int n_blocks, long blk_stride, hipStream_t stream);
/* (n_blocks packs in one launch: block k reads wcomp + k * 8 * H and writes out + k * blk_stride elements) */

/* Fused BACKWARD of the same block (round 4): the bias is recomputed on the matrix cores in both passes, the per-pair gradients
 * D = (d lin_h, d gate_h)_h never leave the chip as fp32 -- they overwrite the LDS bias tile as bf16, are contracted with the packed
 * operand `boT` (fabind_pair_bot_pack) for  d a0 += D Bo  in the kernel, and go out as bf16 rows of `acat` (row = compact protein
 * index, columns kcol0 + atom * 8 + slot, slot order lin0, gate0, lin1, gate1, ...) for the T = D^T a0 contraction that the ligand-side
 * and weight gradients need over ALL row tiles of a complex (fabind_batched_transpose_pad + fabind_gemm + fabind_pair_bias_finish,
 * as on the tensor path).  Replaces autograd of cross_att.py:118-134 / model_utils.py:21-38,96-159 for the v1 stack.
 * qg = q | gate pre-activations of the QUERY side [rows, ldq >= 256] (mode 0: protein rows in the node layout; mode 1: compact ligand
 * rows), kv = k | v of the KEY side, out / dout = the forward's output and its gradient [query rows, 128], lse = the forward's
 * log-sum-exp [query rows, 4]; outputs: dqg [query rows, ldq] (d q | d gate-pre), dkv [key rows, ldkv], scratch dO [query rows, 128],
 * Dv [query rows, 4]; da0 (optional: +=, fp32 [N, ldda0], columns [0, H); NULL: the caller contracts `acat` with fabind_pair_bias_btcat's operand), colpart = fabind_cross_attn_fused_bwd_parts(B, max_P) rows of ldcolpart >= 8
 * floats (per-tile column sums of D in the first 8 of a row: their sum is d bconst in slot order), part = float[fabind_cross_attn_fused_bwd_scratch(...)].
 * desc = the forward's descriptors, desc_p = int32[8] per complex whose first entry is the complex's first row in the compact protein
 * list; toff / koff = int32[B + 1] prefix sums of ceil(C / 2) / ceil(C / 4).  nsplit, B, part_rows are filled in by the library. */
typedef struct FabindAttnFusedBwdArgs {
    const float* qg; const float* kv; const void* a0; const void* bo; const void* boT; const int* toff; const int* koff;
    const float* bconst; const int* desc; const int* desc_p; const float* out; const float* lse; const float* dout;
    float* dqg; float* dkv; float* dO; float* Dv; float* da0; void* acat; float* colpart; float* part;
    int ldq, ldkv, lda0, ldda0, ldacat, kcol0, ldcolpart, kp;   /* kp: width of this block's column range of acat (>= max_C * 8; the tail is zero-filled) */
    float scale;
    int nsplit, B, part_rows;
} FabindAttnFusedBwdArgs;
int fabind_cross_attn_fused_bwd(const FabindAttnFusedBwdArgs* args, int B, int max_P, int max_C, int H, int mode, int n_lig_rows,
                                hipStream_t stream);
long fabind_cross_attn_fused_bwd_scratch(int B, int max_P, int max_C, int n_lig_rows, int mode);
int fabind_cross_attn_fused_bwd_parts(int B, int max_P);
/* boT[((koff[b] + ks) * (H / 16) + nt) * 512 + lane * 8 + e] = b0[atom][h] * wcomp[(e & 1) * 4 + (e >> 1)][h], atom = ks * 4 + (lane >> 4),
 * h = nt * 16 + (lane & 15) (zero rows past C): the A operand of the kernel's d a0 contraction.  kstep_b = int32[n_ksteps]: complex of
 * every k-step. */
int fabind_pair_bot_pack(const float* b0, int ldb, const float* wcomp, int H, const int* c_index, const int* desc, const int* koff,
                         const int* kstep_b, int n_ksteps, void* out, int n_blocks, long blk_stride, hipStream_t stream);

/* The same block with QK^T and softmax.V on the matrix cores (v_mfma_f32_16x16x32_bf16: head dim 32 = one instruction per
 * 16-key x 16-query tile; csrc/attn_mfma.hip): identical arguments, results and partials layout; q / k / v / probabilities
 * are rounded to bf16 as MFMA operands, scores, pair bias, softmax statistics and outputs stay fp32.  bias_ld, lin_col and
 * gate_col must be multiples of 4 (a pair's four heads are read as one 16-byte segment). */
int fabind_cross_attn_mfma_fwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* gpre, int ldg,
                               const float* bias, int bias_ld, int lin_col, int gate_col, const int* desc, int B, int max_nq,
                               float scale, float* out, int ldo, float* lse, int ksplit, float* part, int n_rows,
                               hipStream_t stream);

/* pair-bias precompute helpers (RowAttentionBlock.linear/linear_g applied to the initial pair
 * embedding z0 = W_o (a_i * b_j) + b_o; models/cross_att.py:125, models/att_model.py:198-206):
 * bmat[(j*NO + o), k] = b0[c_node(j), k] * wcomp[o, k]  for the ligand-side rows of one batch. */
int fabind_pair_bmat(const float* b0, int ldb, const float* wcomp, int NO, int H, const int* c_node, int n_c,
                     void* bmat, int bmat_dt, hipStream_t stream);
/* hd[e, 0:H] = a0[red_p[e]]*b0[red_c[e]],  hd[e, H:H+H2] = a1[red_p[e]]*b1[red_c[e]]  */
int fabind_pair_hadamard(const float* a0, const float* b0, int ld0, int H, const float* a1, const float* b1, int ld1,
                         int H2, const int* red_p, const int* red_c, int n_red, void* hd, int hd_dt, int ldh,
                         hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Inter-graph attention layer (MC_Att_L.att_model/node_model/coord_model, models/egnn.py:186-252)
 * on node-level projections qkv[N,3H] = [q | k | v] and cv[N,H] = coord_mlp.0(v) :
 *   logit_e = q[r].(k[c] + rhohat_e w_rk) + sum_k bias_red[red_idx[e], k]   (row-dot partials; any
 *             per-layer constant offset cancels in the softmax and is dropped)
 *   alpha   = segment softmax over the row
 *   h_out[r] = h[r] + sum_e alpha_e (v[c] + rhohat_e w_rv)
 *   x_out[r] = x[r] + clamp( sum_e d_e alpha_e * w3.silu(cv[c] + rhohat_e wcr), +-clampv )
 * -------------------------------------------------------------------------------------------*/
int fabind_inter_attn_fwd(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* h, int ldh,
                          const float* x, const float* d, const float* rhohat, const int* rowptr, const int* col,
                          const int* red_idx, const float* bias_red, int bias_np, const float* w_rk,
                          const float* w_rv, const float* wcr, const float* w3, float clampv, int n_rows,
                          float* h_out, float* x_out, float* alpha, float* cvs,
                          const float* s_ext /* optional per-edge scalar added to the coord_mlp value (FABind+: its coord_mlp is an
                                                LN-MLP evaluated outside); its gradient is the backward's dcp[] */,
                          void* h16 /* NULL, or [n_rows, H] bf16: h_out as the next contraction's operand */,
                          hipStream_t stream);
/* The same with the rows DEALT BY DEGREE (round 5, csrc/inter_attn_rows.hip): `order` [n_rows] int32 = the rows sorted by their number of
 * inter edges, descending (stable); order[0 .. n_heavy) are the rows with more than 8 edges -- one work-group of four waves each, a
 * contiguous quarter of the row's edges per wave, log-sum-exp combine in wave order --, order[n_heavy .. n_act) the other rows with edges
 * (one wave each), order[n_act .. n_rows) the rows without edges (pass-through).  Same results up to the summation order of a heavy row;
 * deterministic.  fabind_inter_attn_bwd_rows is the adjoint on the same deal. */
int fabind_inter_attn_fwd_rows(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* h, int ldh,
                               const float* x, const float* d, const float* rhohat, const int* rowptr, const int* col,
                               const int* red_idx, const float* bias_red, int bias_np, const float* w_rk,
                               const float* w_rv, const float* wcr, const float* w3, float clampv, int n_rows,
                               float* h_out, float* x_out, float* alpha, float* cvs, const float* s_ext, void* h16,
                               const int* order, int n_heavy, int n_act, hipStream_t stream);

/* LAS geometry step (models/egnn.py:433-449): x_out = x + clamp(step * sum_{(i,j): j=node} 4(|xi-xj|^2-|x0i-x0j|^2)(xi-xj)).
 * las_off[B+1]: per-complex ranges of the (complex-contiguous) LAS edge list. */
int fabind_las_step(const float* x, const float* x0, const int* las_i, const int* las_j, const int* las_off,
                    const int* node_off, const int* c_cnt, int B, int max_n, float step, float clampv, float* x_out,
                    hipStream_t stream);

/* Pocket centre (models/model.py:146-158): per complex b over its L (padded) residues
 *   s = sigmoid(logit); p = clamp([1 - s, s], 1e-6, 1 - 1e-6); y = softmax((log p + noise) / tau) over the two classes
 *   (noise: the Gumbel samples of F.gumbel_softmax in train mode, [B, L, 2], NULL in eval = utils/utils.py:687-699 gumbel_softmax_no_random);
 *   hard != 0: the straight-through one-hot; w = y[..., 1] * mask; center[b] = sum_l w_l xyz[b, l] / sum_l w_l; wsum[b] = sum_l w_l.
 * logits / mask: [B, L] (fp32 / uint8), xyz [B, L, 3].  _bwd: dlogits [B, L] from dcenter [B, 3] (the soft path's gradient; 0 at masked
 * positions).  One launch each; fixed-order sums. */
int fabind_pocket_center_fwd(const float* logits, const uint8_t* mask, const float* xyz, const float* noise, int B, int L,
                             float tau, int hard, float* center, float* wsum, hipStream_t stream);
int fabind_pocket_center_bwd(const float* logits, const uint8_t* mask, const float* xyz, const float* noise, int B, int L,
                             float tau, const float* center, const float* wsum, const float* dcenter, float* dlogits,
                             hipStream_t stream);

/* Training loss (main_fabind.py:398-417), all six terms and their sum in ONE launch:
 *   out[0] pocket_cls      = w_cls * BCEWithLogits(logits, cls) (mean over all n_cls = B * Lmax positions, padding included) * n_cls / sum(mask)
 *   out[1] pocket_center   = w_center * Huber_delta(center, center_true)                (mean over n_center = 3 B)
 *   out[2] contact         = w_pair * MSE(y_pred, dis_map)      out[3] contact_by_pred = w_pair * MSE(y_by, dis_map)
 *   out[4] distill         = w_distill * MSE(y_by, y_pred)      out[5] coord = w_coord * SmoothL1(coords, coords_true)   (n_coord = 3 * atoms)
 *   out[6] = their sum (in this order), out[7] = sum(mask); loss_out[1] / terms_out[6] (either may be NULL): copies of out[6] / out[0..5].
 * cls_dt: dtype of `cls`: 0 fp32, 1 int64, 2 int32, 3 uint8.  part: fp32 scratch [fabind_loss_blocks(n_pair, n_coord, n_cls)][8];
 * ticket: one uint32 that is ZERO before the first launch (the kernel re-arms it).  Deterministic (block partials summed in block order).
 * fabind_loss_bwd: the gradient seeds d_coords / d_y_pred / d_y_by / d_logits / d_center (any may be NULL) for the upstream gradients
 * g_loss (scalar, of out[6]) and g_terms[6] (of out[0..5]); either may be NULL (= 0).  fwd_out = the forward's out[8]. */
int fabind_loss_blocks(long n_pair, long n_coord, long n_cls);
int fabind_loss_fwd(const float* coords, const float* coords_true, long n_coord, const float* y_pred, const float* y_by,
                    const float* dis_map, long n_pair, const float* logits, const void* cls, int cls_dt, const uint8_t* mask,
                    long n_cls, const float* center, const float* center_true, long n_center, float w_coord, float w_pair,
                    float w_distill, float w_cls, float w_center, float delta, float* part, unsigned* ticket, float* out,
                    float* loss_out, float* terms_out, hipStream_t stream);
int fabind_loss_bwd(const float* coords, const float* coords_true, long n_coord, const float* y_pred, const float* y_by,
                    const float* dis_map, long n_pair, const float* logits, const void* cls, int cls_dt, long n_cls,
                    const float* center, const float* center_true, long n_center, float w_coord, float w_pair, float w_distill,
                    float w_cls, float w_center, float delta, const float* fwd_out, const float* g_loss, const float* g_terms,
                    float* d_coords, float* d_y_pred, float* d_y_by, float* d_logits, float* d_center, hipStream_t stream);

/* Pair operations of the distance-map head (models/model.py:344-365; SURVEY K19 glue) on per-complex BLOCK descriptors: the valid
 * (pocket residue i, ligand atom j) pairs of complex b are the dense block P_b x C_b, protein-major, at pair index pair_off + i * C_b + j
 * (z[z_mask] of the reference's padded [B, P, C] tensors).  desc: device array of B records
 *   struct { int64 pair_off; int32 P, C, p_row0, c_row0, tile0, pad; }      (32 bytes; tile0 = first of the complex's ceil(P / TP) tiles,
 *   TP = fabind_pair_block_tile()), n_tiles = sum_b ceil(P_b / TP).
 * fabind_pair_dist_fwd: y[pair] = clamp(scale * |xp[p_row0 + i] - xc[c_row0 + j]|, lo, hi)   (torch.cdist -> [z_mask] -> unnormalize -> clamp,
 *   model.py:349,363-365).  _bwd: dxc[c_row0 + j] (rows of every complex's atoms; fixed-order sums); part: fp32 scratch [B][8][max_C][3].
 * fabind_block_hadamard_bwd: adjoint of out[pair, :] = tp[p_row0 + i, :] * tc[c_row0 + j, :] (einsum 'bik,bjk->bijk' on the valid pairs,
 *   model.py:355) for a bf16 dout [pairs, W] (W in {64, 128, 256, 512}): dtp / dtc rows written once each, no atomics;
 *   row_b[n_crows]: complex of every row of tc; nchunk_max = ceil(max_b C_b / fabind_pair_block_chunk());
 *   part: fp32 scratch [n_tiles][nchunk_max][chunk][W]. */
int fabind_pair_block_tile(void);
int fabind_pair_block_chunk(void);
int fabind_pair_dist_fwd(const void* desc, int B, int n_tiles, const float* xp, const float* xc, float scale, float lo, float hi,
                         float* y, hipStream_t stream);
int fabind_pair_dist_bwd(const void* desc, int B, int max_C, const float* xp, const float* xc, const float* dy, float scale, float lo,
                         float hi, float* part, float* dxc, hipStream_t stream);
int fabind_block_hadamard_fwd(const void* desc, int B, int n_tiles, const float* tp, int ldtp, const float* tc, int ldtc, int W, void* out,
                              int ldo, hipStream_t stream);      /* out[pair, :] = bf16(tp[p_row0 + i, :] * tc[c_row0 + j, :]) */
int fabind_block_hadamard_bwd(const void* desc, int B, int n_tiles, const void* dout, int ldo, const float* tp, int ldtp,
                              const float* tc, int ldtc, int W, const int* row_b, int n_crows, int nchunk_max, float* part,
                              float* dtp, int lddp, float* dtc, int lddc, hipStream_t stream);

/* x_out[i] = mask[i] ? z[i] : x[i]   (X[mask] = Z[mask], models/att_model.py:236,245) */
int fabind_select_rows(const float* x, const float* z, const uint8_t* mask, int n, int width, float* x_out,
                       hipStream_t stream);

/* LayerNorm over the last dim (models/model.py:15,29): y = (x-mean)*rstd*w + b; backward returns dx and
 * per-block partial sums of dw/db ([ceil(R/4), C] each, reduce with fabind_colsum). */
/* Per-row LayerNorm statistics without writing the normalised rows: mu[r] = mean, rs[r] = rsqrt(centred variance + eps)
 * of x[r, 0:C] (two passes over registers, like fabind_layernorm_rows).  Feeds the row_mu / row_rs fold of fabind_gemm. */
int fabind_row_stats(const void* x, int x_dt, int ldx, float eps, int R, int C, float* mu, float* rs, hipStream_t stream);

/* FABind+ LN-MLPs (FABind_plus/fabind/models/model_utils.py:10-74).  fabind_layernorm_rows: row LayerNorm with free
 * input / output dtypes (FB_DT_*) and strides; output columns [C, pad_to) are zero-filled (GEMM K padding).
 * fabind_edge_ln_concat: LayerNorm over the concatenated edge input [h[row] | h[col] | rhohat] (2H+1 columns,
 * FABind_plus/fabind/models/egnn.py:52-58) evaluated while gathering. */
int fabind_layernorm_rows(const void* x, int x_dt, int ldx, const float* w, const float* b, float eps, int R, int C, void* y,
                          int y_dt, int ldy, int pad_to, hipStream_t stream);
int fabind_edge_ln_concat(const float* h, int ldh, int H, const int* row, const int* col, const float* rhohat, const float* w,
                          const float* b, float eps, int E, void* y, int y_dt, int ldy, int pad_to, hipStream_t stream);

/* First Linear of FABind+'s LN-MLP edge model (egnn.py:52-58, model_utils.py:44-46) with the LayerNorm folded into
 * per-node projections, bf16 inference:
 *   out[e,:] = drop(relu( rs_e * (AB[row[e], 0:Kp] + AB[col[e], Kp:2Kp] + (m_r-mu_e) c_r + (m_c-mu_e) c_c + (rho[e]-mu_e) w_r) + dvec ))
 * AB = bf16 [nodes, ldab] = (h - m 1^T) [W1 diag(ln_w)]_{r|c}^T (node-centred features), stat = fp32 [nodes][2] =
 * (mean m, centred sum of squares) of each node's H features; (mu_e, rs_e) are the LayerNorm statistics of
 * [h_r | h_c | rho] (2H+1 values) combined from them; c_r / c_c = row sums of the two weight blocks.  out = bf16 [E, Kp].
 * Dropout mask: hash32(seed + e*Kp + c) & 0xffff >= round(p*65536). */
int fabind_edge_lnfold(const void* AB, int ldab, int Kp, int H, const int* row, const int* col, const float* rho,
                       const float* stat, float eps, const float* w_r, const float* c_r, const float* c_c, const float* dvec,
                       int E, void* out, float p_drop, unsigned seed, hipStream_t stream);
/* Adjoint of fabind_edge_lnfold (round 5: the folded first edge Linear under autograd): out = what the forward wrote (its zeros are the
 * inactive and the dropped positions), dout = its gradient (bf16 [E, Kp]).  Writes du = rs_e x dout x mask x 1/(1-p) as bf16 [E, Kp] (the
 * caller's segment sums over the receiving / sending node give the gradient of AB's two halves), es = fp32 [E][8] = (d m_r, d Q_r, d m_c,
 * d Q_c, 0, 0, 0, 0): the gradients of the receiving / sending node's statistics through this edge, drho [E], and part = fp32 [n_blocks][4][Kp]
 * partial column sums (d dvec, d c_r, d c_c, d w_r); n_blocks = fabind_edge_lnfold_bwd_blocks(E).  Kp <= 1536. */
int fabind_edge_lnfold_bwd_blocks(int E);
int fabind_edge_lnfold_bwd(const void* AB, int ldab, int Kp, int H, const int* row, const int* col, const float* rho,
                           const float* stat, float eps, const float* w_r, const float* c_r, const float* c_c,
                           const void* out, const void* dout, int E, float p_drop, void* du, float* es,
                           float* drho, float* part, int n_blocks, hipStream_t stream);

/* coord_mlp of FABind+'s inter-edge attention layer (egnn.py:277-300; LN -> Linear -> relu -> dropout -> bias-free Linear)
 * on the value rows v_e = V[col] + rho * w_rv, LayerNorm folded into a per-node projection, bf16 inference:
 *   s_out[e] = sum_k w3[k] * drop(relu( rs_e * (P[col[e], k] + rho[e] * u[k]) + d[k] )),
 *   rs_e = rsqrt( (stat[col][0] + 2 rho stat[col][1] + rho^2 q_w) / H + eps )
 * P = bf16 [nodes, ldp] = (V - mean 1^T)(W1 diag(ln_w))^T; stat = fp32 [nodes][2] = (|Vc|^2, Vc . wc); wc = w_rv - mean(w_rv),
 * q_w = |wc|^2, u = W1 diag(ln_w) wc, d = W1 ln_b + b1.  H % 8 == 0, H <= 512.
 * Dropout mask: hash32(seed + e*H + k) & 0xffff >= round(p*65536). */
int fabind_inter_coord_fold(const void* P, int ldp, int H, const int* col, const float* rho, const float* stat, float q_w,
                            float eps, const float* u, const float* d, const float* w3, int E, float* s_out, float p_drop,
                            unsigned seed, hipStream_t stream);
/* adjoint of fabind_layernorm_rows: dx (dtype dx_dt, leading dim lddx) and per-work-group partials of dw, db ([nblk][C] each;
 * nblk = the grid the caller chooses, one wave per row strided over it -- every partial row is written; the caller adds the rows).  fabind_edge_concat: y[e] = [h[row[e]] | h[col[e]] | rhohat[e] | 0...]
 * (the un-normalised edge input of MC_E_GCL, used under autograd where LayerNorm is a separate differentiable step). */
int fabind_layernorm_rows_bwd(const void* x, int x_dt, int ldx, const float* w, const void* dy, int dy_dt, int lddy, float eps,
                              int R, int C, void* dx, int dx_dt, int lddx, float* dw_part, float* db_part, int nblk, hipStream_t stream);
int fabind_edge_concat(const float* h, int ldh, int H, const int* row, const int* col, const float* rhohat, int E, void* y,
                       int y_dt, int ldy, int pad_to, hipStream_t stream);
int fabind_layernorm_fwd(const float* x, const float* w, const float* b, float eps, int R, int C, float* y, float* mean,
                         float* rstd, hipStream_t stream);
int fabind_layernorm_bwd(const float* x, const float* w, const float* dy, const float* mean, const float* rstd, int R,
                         int C, float* dx, float* dw_part, float* db_part, hipStream_t stream);

/* elementwise: out = a + b (fp32), n elements */
int fabind_add(const float* a, const float* b, float* out, long n, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward passes (adjoints of the kernels above; they replace torch autograd's traversal of the
 * reference's ATen/torch_scatter graph for the same ops).  Contractions in backward passes go through
 * fabind_gemm / fabind_transpose_act / fabind_colsum.
 * -------------------------------------------------------------------------------------------*/
int fabind_mul_dact(const void* dy, int dy_dt, const void* y, int y_dt, int act, void* out, int out_dt, long n, float scale,
                    hipStream_t stream);
/* out = scale * dy * act'(y) (y = NULL with FB_ACT_NONE: dtype conversion) and colsum[c] = sum_r out[r,c] in one pass
 * (bias gradient of the Linear whose activation is being undone); [R,C] contiguous, scratch [nchunk, C].
 * scale: 1, or keep^-1 = 1 / (1 - p) when y is the output of a ReLU epilogue WITH dropout (fabind_gemm p_drop): y = relu(pre) keep / (1 - p),
 * so d y / d pre = [y > 0] / (1 - p) -- the dropped positions are exactly the zeros of y, no mask is stored or recomputed. */
int fabind_mul_dact_colsum(const void* dy, int dy_dt, const void* y, int y_dt, int act, void* out, int out_dt, int R, int C,
                           float* colsum, float* scratch, int nchunk, float scale, hipStream_t stream);
/* out [R,C] = dy * keep(seed, r, c) / (1 - p) and its column sums: the adjoint of fabind_gemm's epilogue dropout (key seed + r * C + c,
 * C = the GEMM's N) where the saved output does not show the mask -- dropout AHEAD of a residual (the train-mode sites of the
 * reference: nn.Dropout on the node-MLP output / the attention update before `h + ...`, FABind/fabind/models/egnn.py:106,
 * cross_att.py:128) or without an activation.  Nothing is stored by the forward: the mask is regenerated from its key. */
int fabind_mul_dropmask_colsum(const void* dy, int dy_dt, void* out, int out_dt, int R, int C, float p_drop, unsigned seed,
                               float* colsum, float* scratch, int nchunk, hipStream_t stream);
/* out = h + drop(hn - h) over n fp32 elements (nn.Dropout on the aggregated inter-edge attention message ahead of the residual,
 * FABind/fabind/models/egnn.py:236) and its adjoint (d hn = g keep / (1 - p), d h = g - d hn): the counter-based mask of the GEMM
 * epilogues keyed by (seed, linear index), regenerated by the adjoint. */
int fabind_drop_mix(const float* h, const float* hn, float* out, long n, float p_drop, unsigned seed, hipStream_t stream);
int fabind_drop_mix_bwd(const float* g, float* dhn, float* dh, long n, float p_drop, unsigned seed, hipStream_t stream);
int fabind_rowdot_bwd(const void* z, int z_dt, const float* dpart, int np, const float* u, int act, int M, int N,
                      void* dz, float* du, float* scratch, int nchunk, hipStream_t stream);
int fabind_edge_geom_bwd(const float* d, const float* rho, const float* norm, const float* dd, const float* drhohat,
                         const int* row, const int* rowptr, const int* colptr /* edges grouped by sending node */,
                         const int* perm /* their edge ids */, const int* node_off, int B, int E, int n_nodes, float* dx,
                         float* g_scratch /* [E,3] */, float* T_scratch /* [B] */, hipStream_t stream);
int fabind_gcl_pre_bwd(const void* dpre, int dt, int H, const float* rhohat, const float* w_r, int E, float* drh,
                       float* dw, float* scratch, int nchunk, hipStream_t stream);
int fabind_gather_dact(const float* dout, int ldo, const int* row, const void* Z, int z_dt, int act, void* dZ, int dz_dt,
                       int E, int H, hipStream_t stream);
int fabind_coord_update_bwd(const float* d, const float* s, const int* rowptr, int n_rows, int mean, float clampv,
                            const float* dxo, float* dd, float* ds, hipStream_t stream);
/* Adjoint of fabind_cross_attn_fwd.  Writes dqg / dkv / dbias only at the rows / pairs the block descriptors cover (the caller
 * zero-initialises them if other rows exist).  When one side has <= 256 rows per complex and the other >= 512, the long loop of
 * the short side is split over work-groups; the per-split partial sums go through `scratch` (fabind_cross_attn_bwd_scratch()
 * floats, fp32) and are added by a reduce kernel -- no float atomics, the result is deterministic. */
long fabind_cross_attn_bwd_scratch(int B, int max_nq, int max_nk);
int fabind_cross_attn_bwd(const float* qg, int ldq, const float* kv, int ldkv, const float* bias, int bias_ld,
                          int lin_col, int gate_col, const int* desc, int B, int max_nq, int max_nk, float scale,
                          const float* out, const float* lse, const float* dout, float* dqg, float* dkv, float* dbias,
                          float* dO, float* Dv, float* scratch, hipStream_t stream);
/* MFMA bf16 form of the same adjoint (csrc/attn_mfma.hip): same arguments and scratch size; dO, dS and the recomputed
 * probabilities enter the contractions as bf16, dbias / dq / dk / dv are accumulated and written in fp32. */
int fabind_cross_attn_mfma_bwd(const float* qg, int ldq, const float* kv, int ldkv, const float* bias, int bias_ld,
                               int lin_col, int gate_col, const int* desc, int B, int max_nq, int max_nk, float scale,
                               const float* out, const float* lse, const float* dout, float* dqg, float* dkv, float* dbias,
                               float* dO, float* Dv, float* scratch, hipStream_t stream);
/* Operands of the pair-bias adjoint (autograd of the pair bias of RowAttentionBlock, cross_att.py:125, all attention blocks at
 * once) concatenated along K, bf16:  Acat[poff_b + i, k*Kp + j*8 + o] = douts[k][pair(b,i,j), o],
 * BTcat[b*H + h, k*Kp + j*8 + o] = b0[ligand j of b, h] * wcomp[k][o][h]  (zeros for j >= C_b; Kp = 8 * max_C rounded up to 32).
 * douts: HOST array of nblk device pointers (fp32 [n_pairs, 8] each, NULL = no gradient); desc_p: the block descriptors of
 * fabind_cross_attn_fwd with the protein side compact; lda must be nblk*Kp (Acat rows are written contiguously).  colpart
 * (optional): fp32 [fabind_pair_bias_cat_parts(B, max_P)][nblk*8], one row of partial column sums of the douts per work-group --
 * their sum over the rows is the gradient of the pair-bias constants (no atomics, fixed order). */
long fabind_pair_bias_cat_parts(int B, int max_P);
int fabind_pair_bias_cat(const void* const* douts, int nblk, const int* desc_p, int B, int max_P, int Kp, void* Acat, int lda,
                         const float* b0, int ld_ab, const int* c_index, const float* wcomp, int H, void* BTcat, int ldb,
                         float* colpart, hipStream_t stream);
/* BTcat alone -- BTcat[b * H + h, k * Kp + j * 8 + o] = b0[c_index[lig0_b + j], h] * wcomp[k][o][h] (bf16, zero padded) -- for callers that
 * write the gradient rows Acat themselves (fabind_cross_attn_fused_bwd; pass wcomp rows in ITS slot order). */
int fabind_pair_bias_btcat(const float* b0, int ld_ab, const int* c_index, const int* desc_p, const float* wcomp, int nblk, int H, int Kp,
                           void* BTcat, int ldb, int B, hipStream_t stream);
/* Batched transpose with zero padding, bf16: for b < B, k < nsub, m < cols:
 *   out[(k*B + b)*cols + m][i] = in[desc[b][0] + i][k*cols + m]  for i < desc[b][1], 0 for i up to Pp   (out row stride Pp).
 * Turns the row-major per-complex operands of the pair-bias adjoint into the K-major, uniformly padded layout the pipelined NT GEMM
 * takes with plain groups. */
int fabind_batched_transpose_pad(const void* in, int ld_in, const int* desc, int B, int nsub, int cols, int Pp, void* out,
                                 hipStream_t stream);
int fabind_pair_hadamard_bwd(const void* dhd, int dt, int ldh, const float* a0, const float* b0, int ld0, int H,
                             const float* a1, const float* b1, int ld1, int H2, const int* red_p, const int* red_c, int n,
                             float* da0, float* db0, int ldd0, float* da1, float* db1, int ldd1, hipStream_t stream);
/* Adjoint of the pair Hadamard over the ALL-PAIRS list of a batch (pair (b, i, j) at pair_off[b] + i * C_b + j: FABind+ threads the
 * pair embedding over every protein x ligand pair, FABind_plus/fabind/models/model_utils.py:96-146) without float atomics: one wave per
 * node, one writer per element.  T = (a | b) fp32 rows; dT receives d a in columns [0, W) of protein rows and d b in [W, 2W) of ligand
 * rows.  node_b[n] = complex of node n; desc_p = the layout's int32[8] block descriptors (pair offset in words 4, 5). */
int fabind_pair_hadamard_bwd_grid(const void* dhd, int dt, int ldh, const float* T, int ldt, int W, const int* node_off,
                                  const int* c_cnt, const int* node_b, const int* desc_p, int n_nodes, float* dT, int lddt,
                                  hipStream_t stream);
/* Forward-only node-level chain over 64-row tiles with the hidden activation kept in LDS (csrc/node_chain.hip):
 *   kind 0: out = act([X1 | X2] W1^T + b1) W2^T + b2 (+ R)   hidden = H    -- MC_E_GCL.node_model, FABind/fabind/models/egnn.py:99-109
 *   kind 1: out = act(X1 W1^T + b1) W2^T + b2 (+ R)          hidden = 2 H  -- Transition, FABind/fabind/models/model_utils.py:162-175
 * X1, X2: bf16 [M, H]; every weight operand is one H x H block in the fragment order of the fused edge kernels ([K/32][N/16][64][8]):
 * kind 0: W1a / W1b = W1[:, :H] / W1[:, H:], W2a = W2 (W2b unused); kind 1: W1a / W1b = W1[:H] / W1[H:], W2a / W2b = W2[:, :H] / W2[:, H:].
 * b1 has the hidden width; R (optional) and out are fp32 [M, H]; out16 (optional) receives bf16(out).  act = FB_ACT_SILU | FB_ACT_RELU.
 * H in {128, 256, 512}. */
int fabind_node_chain_fwd(const void* X1, int ld1, const void* X2, int ld2, const void* W1a, const void* W1b, const float* b1,
                          const void* W2a, const void* W2b, const float* b2, int act, int kind, const float* R, int ldr, float* out,
                          int ldo, void* out16, int ldo16, int M, int H, hipStream_t stream);
/* The same chains in SPLIT PRECISION (round 6; bf16 mode with config.set_split_sites(3)): fp32 X rows and fp32 master weights given as
 * hi | lo bf16 fragment packs (hi = bf16(W), lo = bf16(W - hi), each in the layout of fabind_node_chain_fwd's packs); every product term
 * is three MFMAs, the hidden layer is never rounded to bf16.  X1 / X2 / R / out: fp32 rows (ld % 4 == 0, 16-byte aligned).
 * t16 (training forward; NULL otherwise): bf16 [M, hidden] of the activated hidden layer, d16 (SiLU only): bf16 of its derivative -- the
 * operands of the bf16 backward (weight gradient of the second Linear, activation adjoint inside the input-gradient GEMM). */
int fabind_node_chain_x3_fwd(const float* X1, int ld1, const float* X2, int ld2, const void* W1ah, const void* W1al, const void* W1bh,
                             const void* W1bl, const float* b1, const void* W2ah, const void* W2al, const void* W2bh, const void* W2bl,
                             const float* b2, int act, int kind, const float* R, int ldr, float* out, int ldo, void* out16, int ldo16,
                             void* t16, int ldt, void* d16, int ldd, int M, int H, hipStream_t stream);
/* Adjoint of out[e, :] = t[ia[e], :] * t[ib[e], :] (the distance-map head's LN(p_i) * LN(c_j) over every pocket residue x ligand atom
 * pair, FABind/fabind/models/model.py:355) without float atomics: the pairs of every row of t are given as a CSR (rowptr [n_rows + 1];
 * per entry the pair's row of dout and the partner's row of t); one wave per row, dT[n, :] = sum_e dout[pair_idx[e], :] * t[partner[e], :]
 * (written, not accumulated; rows without pairs receive zeros).  W <= 1024, W % 4 == 0.  Deterministic summation order. */
int fabind_rows_hadamard_bwd(const void* dout, int dt, int ldo, const float* t, int ldt, int W, const int* rowptr, const int* pair_idx,
                             const int* partner, int n_rows, float* dT, int lddt, hipStream_t stream);
/* fabind_pair_hadamard_bwd over the inter graph's reduced pairs without float atomics: one wave per node walks the inter edges of its own
 * row (pair = red_idx[e], partner = col[e]); t0 = (a | b) [N, 2H], t1 = (a | b) [N, 2 H2] (H2 may be 0), the sums are ADDED to d0 / d1
 * in the same layout.  Deterministic summation order. */
int fabind_pair_hadamard_bwd_rows(const void* dhd, int dt, int ldh, const float* t0, int ld0, int H, const float* t1, int ld1, int H2,
                                  const int* rowptr, const int* col, const int* red_idx, const int* red_c, int n_rows, float* d0,
                                  int ldd0, float* d1, int ldd1, hipStream_t stream);
int fabind_inter_attn_bwd(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* d,
                          const float* rhohat, const int* rowptr, const int* col, const int* mirror, const int* red_idx,
                          const float* w_rk, const float* w_rv, const float* wcr, const float* w3, const float* alpha,
                          const float* cvs, float clampv, int n_rows, const float* dh_out, const float* dx_out,
                          float* dqkv, float* dcv, float* dd, float* drh, float* dbias_red, float* dlogit, float* dcp,
                          float* wpart, int nblk, hipStream_t stream);
int fabind_inter_attn_bwd_rows(const float* qkv, int ldqkv, const float* cv, int ldcv, int H, const float* d,
                               const float* rhohat, const int* rowptr, const int* col, const int* mirror, const int* red_idx,
                               const float* w_rk, const float* w_rv, const float* wcr, const float* w3, const float* alpha,
                               const float* cvs, float clampv, int n_rows, const float* dh_out, const float* dx_out,
                               float* dqkv, float* dcv, float* dd, float* drh, float* dbias_red, float* dlogit, float* dcp,
                               float* wpart, int nblk, const int* order, int n_heavy, int n_act, hipStream_t stream);
int fabind_las_step_bwd(const float* x, const float* x0, const float* xo, const int* las_i, const int* las_j,
                        const int* las_off, const int* node_off, const int* c_cnt, int B, int max_n, float step,
                        float clampv, const float* dout, float* dx, hipStream_t stream);
int fabind_pair_bias_bwd(const float* dout, int NO, const float* ab, int ld, int H, const float* w, const int* desc_p,
                         int B, int max_P, int max_C, const int* p_index, const int* c_index, float* dab, float* dwk,
                         hipStream_t stream);
/* dab[c_index[jc], H + h] += sum_o w[o,h] T[(jc,o),h];  dw_part[g * ld_part + o * H + h] = sum_{jc in group g of 16} ab[c_index[jc], H + h]
 * T[(jc,o),h] for g < fabind_pair_bias_finish_parts(n_c): per-group partials of the weight gradient, WRITTEN (no float atomics); the
 * caller sums the rows (fabind_colsum over [parts, ld_part]) -- several blocks' partials side by side take one sum. */
int fabind_pair_bias_finish_parts(int n_c);
int fabind_pair_bias_finish(const float* T, const float* ab, int ld, int H, const float* w, const int* c_index, int n_c,
                            float* dab, float* dw_part, long ld_part, hipStream_t stream);
void fabind_gemm_set_config(int cfg); /* development knob: tile configuration of the bf16 LDS-DMA GEMM (default 13 = 256x128x32, 3 stages,
                                         two work-groups per CU; 0 = register-staged kernel; 1-9 = other tiles; results do not depend on it) */
void fabind_gemm_set_persistent(int on); /* development knob: persistent tile walk for long-M/short-K GEMMs */
int fabind_gemm_x3_occupancy(int wm); /* development probe: resident work-groups per CU of the split-bf16 GEMM kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor) */
void fabind_cross_attn_fused_set_dbg(void* five_int64_on_device); /* development probe: phase cycle counters of one work-group of the fused cross-attention forward (NULL = off) */
int fabind_cross_attn_fused_occupancy(int lds_bytes); /* the same for the fused cross-attention forward at the given dynamic LDS size */
void fabind_gemm_set_x3_tile(int wm); /* development knob: tile height of the split-bf16 fabind_gemm kernel in units of 64 rows (2 = default: 128x128, two 4-wave work-groups per CU; 4: 256x128, one 8-wave work-group) */
void fabind_gemm_set_small_m(int tiles); /* development knob: fabind_gemm launches with fewer 256x128 tiles than this use 128x128 tiles (default 100; 0 = never); results are bitwise equal */
void fabind_gemm_tn_set_waves(int waves); /* development knob: work-group layout of fabind_gemm_tn: 16 (default) = 256x256 tile, 8 waves, 4-stage ring; 20 = the same with 5 stages; 4 = 256x128 tile, 4 waves, two work-groups per CU; 8 = 256x128, 8 waves.  Results are bitwise equal for equal `splits` */
int fabind_gemm_tn_tile_n(void);          /* 256 or 128: columns of an output tile under the current layout (the host sizes `splits` from the tile count) */
void fabind_gemm_tn_set_exp(int mask);    /* probe knob (tools/probes/gemm_tn_bench.py): 1 = no operand loads after the pipeline fill, 4 = operand loads re-read the first rows (cache hits); 2 is retired (ignored); 0 in production */

#ifdef __cplusplus
}
#endif
#endif
