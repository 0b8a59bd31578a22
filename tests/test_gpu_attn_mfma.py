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
