"""GPU: cross attention as MFMA bf16 tiles (csrc/attn_mfma.hip; north_star: "cross-attention QK^T / softmax.V as MFMA bf16
tiles") against (a) the CPU oracle's mha at the headline block shapes (P, C) = (1501, 41) and (41, 1501), and (b) the fp32
VALU kernels it replaces in bf16 mode -- forward, log-sum-exp and every gradient (d q, d gate, d k, d v, d pair bias), incl.
the split-key / split-query paths and ragged blocks.  Reference: FABind/fabind/models/cross_att.py:118-134,
model_utils.py:21-38,96-133."""
import math

import numpy as np
import pytest
import torch

import fabind_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _desc(nqs, nks, q_major=True):
    """Block descriptors of B ragged blocks: queries / keys compact, pair = pair_off + i*sq + j*sk."""
    nqs, nks = np.asarray(nqs, np.int64), np.asarray(nks, np.int64)
    qo, ko = np.cumsum(nqs) - nqs, np.cumsum(nks) - nks
    po = np.cumsum(nqs * nks) - nqs * nks
    sq, sk = (nks, np.ones_like(nks)) if q_major else (np.ones_like(nqs), nqs)
    d = np.stack([qo, nqs, ko, nks, po & 0xFFFFFFFF, po >> 32, sq, sk], 1).reshape(-1)
    return torch.from_numpy((d & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).view(-1, 8).to(DEV), int((nqs * nks).sum())


def _inputs(nqs, nks, seed, q_major=True):
    g = torch.Generator().manual_seed(seed)
    desc, n_pairs = _desc(nqs, nks, q_major)
    qg = torch.randn(int(sum(nqs)), 256, generator=g).to(DEV)
    kv = torch.randn(int(sum(nks)), 256, generator=g).to(DEV)
    bias = torch.randn(n_pairs, 8, generator=g).to(DEV)
    return desc, qg, kv, bias


def _run(mfma, desc, qg, kv, bias, B, mq, mk, cot=None):
    from fabind_amd import kernels as K
    from fabind_amd import ops
    K.CROSS_ATTN_MFMA = mfma
    try:
        a, b, c = qg.clone().requires_grad_(True), kv.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        out = ops.cross_attn(a, b, c, 0, 4, desc, B, mq, mk, 1.0 / math.sqrt(32.0))
        if cot is None:
            return out.detach()
        (out * cot).sum().backward()
        return out.detach(), a.grad, b.grad, c.grad
    finally:
        K.CROSS_ATTN_MFMA = None


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("nqs,nks,q_major", [
    ([50, 33, 64, 17], [9, 12, 41, 5], True),            # protein queries, few keys, ragged (no split)
    ([1501, 1400], [41, 37], True),                      # headline p-block: the query pass of the backward is split over queries
    ([41, 37], [1501, 1400], False),                     # headline c-block: keys split over work-groups (partials + combine)
    ([130, 70], [300, 260], False),                      # mid sizes, several key chunks with online rescaling, no split
])
def test_mfma_cross_attention_matches_fp32_kernels(nqs, nks, q_major):
    desc, qg, kv, bias = _inputs(nqs, nks, 5, q_major)
    B, mq, mk = len(nqs), max(nqs), max(nks)
    cot = torch.randn(qg.shape[0], 128, generator=torch.Generator().manual_seed(6)).to(DEV)
    ref = _run(False, desc, qg, kv, bias, B, mq, mk, cot)
    got = _run(True, desc, qg, kv, bias, B, mq, mk, cot)
    names = ("out", "d qg", "d kv", "d bias")
    errs = [_rel(g_, r_) for g_, r_ in zip(got, ref)]
    print("MFMA vs fp32 cross attention", nqs[:2], nks[:2], dict(zip(names, ["%.2e" % e for e in errs])))
    # bf16 operands (q, k, v, p, dO, dS: 2^-9 relative each) against exact fp32: a few 1e-3 of the largest entry
    for n_, e in zip(names, errs):
        assert e < 2e-2, (n_, e)
    # a transposed / permuted operand would give O(1): demand the bulk to be much tighter
    for g_, r_ in zip(got, ref):
        assert float((g_ - r_).norm() / r_.norm()) < 6e-3


@pytest.mark.parametrize("Q,Kn", [(1501, 41), (41, 1501)])
def test_mfma_attention_vs_oracle_mha_at_headline_block_shapes(Q, Kn):
    """Attention.forward through the MFMA kernels vs oracle.mha."""
    from fabind_amd import engine
    from fabind_amd import kernels as K
    from fabind_amd.models.model_utils import Attention
    torch.manual_seed(0)
    H, B = 64, 2
    att = Attention(H, H, H, 32, 4).eval()
    g = torch.Generator().manual_seed(1)
    q_x, kv_x = torch.randn(B, Q, H, generator=g), torch.randn(B, Kn, H, generator=g)
    pair_bias = torch.randn(B, 4, Q, Kn, generator=g)
    sd = {k: v.detach().clone() for k, v in att.state_dict().items()}
    ref = orc.mha(sd, "", q_x, kv_x, [pair_bias])
    engine.set_precision("fp32")                      # projections exact; only the attention core runs in bf16 MFMA
    K.CROSS_ATTN_MFMA = True
    try:
        got = att.to(DEV)(q_x.to(DEV), kv_x.to(DEV), [pair_bias.to(DEV)])
    finally:
        K.CROSS_ATTN_MFMA = None
    err = float((got.cpu() - ref).abs().max() / ref.abs().max())
    print("Attention (MFMA core) vs oracle.mha at Q=%d K=%d: max rel err %.3e" % (Q, Kn, err))
    assert err < 1e-2 and float((got.cpu() - ref).norm() / ref.norm()) < 4e-3


@pytest.mark.parametrize("H", [512, 128])
@pytest.mark.parametrize("sizes", [[(1500, 40)], [(70, 9), (130, 33), (63, 61), (200, 1)], [(1500, 40), (1100, 27), (333, 41)]])
def test_fused_attention_recomputes_the_pair_bias_in_the_kernel(sizes, H):
    """csrc/attn_mfma.hip, fused section: the RowAttention block with its pair bias lin * sigmoid(gate) of z0 = W_o (a_i . b_j) + b_o
    contracted on the matrix cores inside the attention kernel (64 protein rows per work-group), against the tensor path -- the
    ragged [pairs, 8] bias GEMM + cross_attn_mfma_fwd -- on the same operands: both blocks (protein queries / ligand queries with the
    row-tile partials + combine), ragged complexes incl. a one-atom ligand, the limit of 62 ligand-side nodes (61 atoms + the global
    node), a protein side of exactly one 64-row tile (63 residues + the global node)."""
    from fabind_amd import config, engine, ops, synthetic
    config.set_precision("bf16")
    try:
        inp = synthetic.make_stack_batch(sizes, 8, seed=3)
        lay = engine.Layout(inp["batch_id"].to(DEV), inp["segment_id"].to(DEV))
        g = torch.Generator().manual_seed(len(sizes) * 100 + H)
        N = lay.N
        a0b0 = (torch.randn(N, 2 * H, generator=g) * 0.5).to(DEV)
        wcomp = (torch.randn(2, 8, H, generator=g) / H ** 0.5).to(DEV)
        bconst = torch.randn(2, 8, generator=g).to(DEV)
        scale = 1.0 / math.sqrt(32.0)
        with torch.no_grad():
            pb = ops.PairBias(a0b0, H, wcomp, bconst, lay)
            assert pb.can_fuse()
            bias = ops.pair_bias(a0b0, H, wcomp, bconst, lay)
            # protein-query block: q / gate rows node-indexed, k / v compact ligand rows
            qg_p = torch.randn(N, 256, generator=g).to(DEV)
            kv_p = torch.randn(lay.sumC, 256, generator=g).to(DEV)
            ref_p = ops.cross_attn(qg_p, kv_p, bias[0], 0, 4, lay.desc_pf, lay.B, lay.max_P, lay.max_C, scale)
            got_p = ops.cross_attn_fused(qg_p, kv_p, pb, 0, 0, lay, scale)
            # ligand-query block: q / gate compact, k / v node-indexed
            qg_c = torch.randn(lay.sumC, 256, generator=g).to(DEV)
            kv_c = torch.randn(N, 256, generator=g).to(DEV)
            ref_c = ops.cross_attn(qg_c, kv_c, bias[1], 0, 4, lay.desc_cf, lay.B, lay.max_C, lay.max_P, scale)
            got_c = ops.cross_attn_fused(qg_c, kv_c, pb, 1, 1, lay, scale)
        prot = lay.p_index64
        e_p, e_c = _rel(got_p[prot], ref_p[prot]), _rel(got_c, ref_c)
        print("fused attention vs tensor path, sizes %s H=%d: protein-query block %.2e, ligand-query block %.2e" % (sizes, H, e_p, e_c))
        assert torch.isfinite(got_p).all() and torch.isfinite(got_c).all()
        assert e_p < 3e-3 and e_c < 3e-3
        assert float(got_p[lay.c_index64].abs().max()) == 0.0          # rows outside the block stay zero
    finally:
        config.set_precision("fp32")


@pytest.mark.parametrize("H", [512, 128])
@pytest.mark.parametrize("sizes", [[(200, 40)], [(70, 9), (130, 33), (63, 61), (200, 1)], [(1500, 40), (1100, 27), (333, 41)]])
def test_fused_attention_backward_recomputes_the_pair_bias_in_the_kernels(sizes, H):
    """csrc/attn_mfma.hip, fused backward (round 4): ops.cross_attn_fused_train -- forward with the log-sum-exp saved, backward recomputing
    lin / gate on the matrix cores in both passes, d a0 contracted in the kernel, the per-pair gradients leaving only as bf16 rows for the
    T = D^T a0 GEMM -- against the TENSOR path (ragged [pairs, 8] bias GEMM under autograd + cross_attn_mfma fwd / bwd + the pair-bias
    adjoint chain) on the same operands: outputs and the gradients of q | gate, k | v, the factored pair embedding a0 | b0, the composed
    weights and the constants, both blocks, ragged complexes incl. a one-atom ligand and the 62-node limit."""
    from fabind_amd import config, engine, ops, synthetic
    config.set_precision("bf16")
    try:
        inp = synthetic.make_stack_batch(sizes, 8, seed=3)
        lay = engine.Layout(inp["batch_id"].to(DEV), inp["segment_id"].to(DEV))
        g = torch.Generator().manual_seed(len(sizes) * 100 + H + 1)
        N = lay.N
        base = dict(a0b0=(torch.randn(N, 2 * H, generator=g) * 0.5), wcomp=(torch.randn(2, 8, H, generator=g) / H ** 0.5),
                    bconst=torch.randn(2, 8, generator=g), qg_p=torch.randn(N, 256, generator=g), kv_p=torch.randn(lay.sumC, 256, generator=g),
                    qg_c=torch.randn(lay.sumC, 256, generator=g), kv_c=torch.randn(N, 256, generator=g))
        cot_p, cot_c = torch.randn(N, 128, generator=g).to(DEV), torch.randn(lay.sumC, 128, generator=g).to(DEV)
        scale = 1.0 / math.sqrt(32.0)
        prot = lay.p_index64

        def run(fused):
            t = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
            a0b0 = ops.shared_grad(t["a0b0"] * 1.0)
            ops.FUSED_ATTN_TRAIN, ops.FUSED_ATTN_TRAIN_MIN_TILES = fused, 0       # (small test batches: the size threshold of the dispatch off)
            pb = ops.PairBias(a0b0, H, t["wcomp"], t["bconst"], lay)
            assert pb.can_fuse_train() == fused and pb.has_tensors() == (not fused)
            if fused:
                o_p = ops.cross_attn_fused_train(t["qg_p"], t["kv_p"], pb, 0, 0, scale)
                o_c = ops.cross_attn_fused_train(t["qg_c"], t["kv_c"], pb, 1, 1, scale)
            else:
                bias = pb.tensors()
                o_p = ops.cross_attn(t["qg_p"], t["kv_p"], bias[0], 0, 4, lay.desc_pf, lay.B, lay.max_P, lay.max_C, scale)
                o_c = ops.cross_attn(t["qg_c"], t["kv_c"], bias[1], 0, 4, lay.desc_cf, lay.B, lay.max_C, lay.max_P, scale)
            ((o_p * cot_p).sum() + (o_c * cot_c).sum()).backward()
            return o_p.detach(), o_c.detach(), {k: v.grad.clone() for k, v in t.items()}

        try:
            rp, rc, gr = run(False)
            fp, fc, gf = run(True)
        finally:
            ops.FUSED_ATTN_TRAIN, ops.FUSED_ATTN_TRAIN_MIN_TILES = True, 512
        assert _rel(fp[prot], rp[prot]) < 3e-3 and _rel(fc, rc) < 3e-3
        assert float(fp[lay.c_index64].abs().max()) == 0.0
        errs = {}
        for k in base:
            a, b = gf[k], gr[k]
            assert torch.isfinite(a).all(), k
            if k in ("qg_p", "kv_c"):                # node-layout arrays: the block covers the protein rows, the ligand rows carry zero gradient
                assert float(a[lay.c_index64].abs().max()) == 0.0, k
            errs[k] = _rel(a, b)
        print("fused attention fwd+bwd vs tensor path, sizes %s H=%d:" % (sizes, H), {k: "%.1e" % v for k, v in errs.items()})
        for k, v in errs.items():
            assert v < (2e-2 if k in ("wcomp", "bconst") else 1.2e-2), (k, v)
    finally:
        config.set_precision("fp32")


def test_fused_attention_fwd_bwd_vs_oracle_mha():
    """The fused training path (pair bias recomputed on the matrix cores in the forward AND the backward kernels) against the ORACLE's
    `mha` itself (the CPU restatement of Attention.forward, model_utils.py:96-159) -- not against this library's tensor path (VERDICT r3
    weak 5): per complex, oracle.mha runs on [q | gate-pre] / [k | v] rows with selector weights (linear_q = [I 0], linear_g = [0 I],
    linear_k = [I 0], linear_v = [0 I], linear_o = I), its pair bias lin * sigmoid(gate) built from the factored pair embedding in
    float64; outputs and the autograd gradients of every input of both blocks are compared."""
    from fabind_amd import config, engine, ops, synthetic
    sizes, H = [(70, 9), (130, 33), (200, 1), (63, 40)], 128
    config.set_precision("bf16")
    try:
        inp = synthetic.make_stack_batch(sizes, 8, seed=5)
        lay = engine.Layout(inp["batch_id"].to(DEV), inp["segment_id"].to(DEV))
        g = torch.Generator().manual_seed(17)
        N = lay.N
        base = dict(a0b0=(torch.randn(N, 2 * H, generator=g) * 0.5), wcomp=(torch.randn(2, 8, H, generator=g) / H ** 0.5),
                    bconst=torch.randn(2, 8, generator=g), qg_p=torch.randn(N, 256, generator=g), kv_p=torch.randn(lay.sumC, 256, generator=g),
                    qg_c=torch.randn(lay.sumC, 256, generator=g), kv_c=torch.randn(N, 256, generator=g))
        cot_p, cot_c = torch.randn(N, 128, generator=g), torch.randn(lay.sumC, 128, generator=g)
        scale = 1.0 / math.sqrt(32.0)
        # ---- oracle (CPU, fp32 attention on a float64-built bias), complex by complex
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        eye, zer = torch.eye(128), torch.zeros(128, 128)
        sd = {"linear_q.weight": torch.cat([eye, zer], 1), "linear_g.weight": torch.cat([zer, eye], 1), "linear_g.bias": torch.zeros(128),
              "linear_k.weight": torch.cat([eye, zer], 1), "linear_v.weight": torch.cat([zer, eye], 1), "linear_o.weight": eye,
              "linear_o.bias": torch.zeros(128)}
        cidx, pidx = lay.c_index64.cpu(), lay.p_index64.cpu()
        loss = 0.0
        outs_p, outs_c = {}, {}
        for b in range(lay.B):
            P_, C_ = int(lay.P[b]), int(lay.C[b])
            poff = sum(int(x) for x in lay.P[:b])
            prow = pidx[poff:poff + P_]
            crow = cidx[int(lay.coff[b]):int(lay.coff[b]) + C_]
            a0, b0 = t["a0b0"][prow, :H].double(), t["a0b0"][crow, H:].double()
            def bias_of(k):
                z = torch.einsum("ih,jh,oh->ijo", a0, b0, t["wcomp"][k].double()) + t["bconst"][k].double()
                return (z[..., :4] * torch.sigmoid(z[..., 4:])).float()                       # [P, C, 4 heads]
            bp = bias_of(0).permute(2, 0, 1)[None]                                          # protein queries: [1, 4, P, C]
            bc = bias_of(1).permute(2, 1, 0)[None]                                          # ligand queries:  [1, 4, C, P]
            lo = int(lay.coff[b])
            o_p = orc.mha(sd, "", t["qg_p"][prow][None], t["kv_p"][lo:lo + C_][None], [bp])[0]
            o_c = orc.mha(sd, "", t["qg_c"][lo:lo + C_][None], t["kv_c"][prow][None], [bc])[0]
            outs_p[b], outs_c[b] = (prow, o_p.detach()), (lo, o_c.detach())
            loss = loss + (o_p * cot_p[prow]).sum() + (o_c * cot_c[lo:lo + C_]).sum()
        loss.backward()
        # ---- the fused kernels
        d = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
        a0b0 = ops.shared_grad(d["a0b0"] * 1.0)
        ops.FUSED_ATTN_TRAIN_MIN_TILES = 0
        pb = ops.PairBias(a0b0, H, d["wcomp"], d["bconst"], lay)
        ops.FUSED_ATTN_TRAIN_MIN_TILES = 512
        assert not pb.has_tensors()
        f_p = ops.cross_attn_fused_train(d["qg_p"], d["kv_p"], pb, 0, 0, scale)
        f_c = ops.cross_attn_fused_train(d["qg_c"], d["kv_c"], pb, 1, 1, scale)
        ((f_p * cot_p.to(DEV)).sum() + (f_c * cot_c.to(DEV)).sum()).backward()
        for b in range(lay.B):
            prow, rp = outs_p[b]
            lo, rc = outs_c[b]
            assert _rel(f_p.detach().cpu()[prow], rp) < 1e-2 and _rel(f_c.detach().cpu()[lo:lo + rc.shape[0]], rc) < 1e-2
        errs = {k: _rel(d[k].grad.cpu(), t[k].grad) for k in base}
        print("fused attention fwd+bwd vs oracle.mha autograd:", {k: "%.1e" % v for k, v in errs.items()})
        for k, v in errs.items():
            assert v < 2.5e-2, (k, v)
    finally:
        config.set_precision("fp32")


def test_fused_attention_serves_one_differentiable_pass_at_a_time():
    """the blocks of one differentiable pass share state on their PairBias.  A second differentiable forward of a block
    whose backward has not run is refused -- also after an abandoned pass --, and a new context reproduces the first pass's gradients."""
    from fabind_amd import config, engine, ops, synthetic
    H = 128
    config.set_precision("bf16")
    old = (ops.FUSED_ATTN_TRAIN, ops.FUSED_ATTN_TRAIN_MIN_TILES)
    try:
        ops.FUSED_ATTN_TRAIN, ops.FUSED_ATTN_TRAIN_MIN_TILES = True, 0
        inp = synthetic.make_stack_batch([(70, 9), (130, 33)], 8, seed=3)
        lay = engine.Layout(inp["batch_id"].to(DEV), inp["segment_id"].to(DEV))
        g = torch.Generator().manual_seed(11)
        N = lay.N
        t = {k: v.to(DEV).requires_grad_(True) for k, v in dict(
            a0b0=torch.randn(N, 2 * H, generator=g) * 0.5, wcomp=torch.randn(2, 8, H, generator=g) / H ** 0.5, bconst=torch.randn(2, 8, generator=g),
            qg_p=torch.randn(N, 256, generator=g), kv_p=torch.randn(lay.sumC, 256, generator=g), qg_c=torch.randn(lay.sumC, 256, generator=g),
            kv_c=torch.randn(N, 256, generator=g)).items()}
        scale = 1.0 / math.sqrt(32.0)
        pb = ops.PairBias(ops.shared_grad(t["a0b0"] * 1.0), H, t["wcomp"], t["bconst"], lay)

        def one_pass():
            o_p = ops.cross_attn_fused_train(t["qg_p"], t["kv_p"], pb, 0, 0, scale)
            o_c = ops.cross_attn_fused_train(t["qg_c"], t["kv_c"], pb, 1, 1, scale)
            return o_p, o_c

        o_p, o_c = one_pass()
        with pytest.raises(RuntimeError, match="ONE differentiable pass"):
            ops.cross_attn_fused_train(t["qg_p"], t["kv_p"], pb, 0, 0, scale)
        (o_p.square().sum() + o_c.square().sum()).backward(retain_graph=False)
        first = {k: v.grad.clone() for k, v in t.items() if k in ("wcomp", "bconst", "qg_p", "kv_c")}
        assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in first.values())
        # an abandoned pass (forward only, graph dropped) keeps its context refused; a new context reproduces the gradients
        pb2 = ops.PairBias(ops.shared_grad(t["a0b0"] * 1.0), H, t["wcomp"], t["bconst"], lay)
        ops.cross_attn_fused_train(t["qg_p"], t["kv_p"], pb2, 0, 0, scale)
        with pytest.raises(RuntimeError, match="ONE differentiable pass"):
            ops.cross_attn_fused_train(t["qg_p"], t["kv_p"], pb2, 0, 0, scale)
        for v in t.values():
            v.grad = None
        pb = ops.PairBias(ops.shared_grad(t["a0b0"] * 1.0), H, t["wcomp"], t["bconst"], lay)
        o_p, o_c = one_pass()
        (o_p.square().sum() + o_c.square().sum()).backward()
        for k, v in first.items():
            assert _rel(t[k].grad, v) < 1e-5, k
    finally:
        ops.FUSED_ATTN_TRAIN, ops.FUSED_ATTN_TRAIN_MIN_TILES = old
        config.set_precision("fp32")
