"""PROBE: the forward node-level chains as one kernel (csrc/node_chain.hip) against the two GEMM launches they replace, at the bench
shape's node count (98,688 rows, H = 512) and the pocket shape's (9,088)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabind_amd import config, kernels as K
dev = torch.device("cuda:0")
config.set_precision("bf16")
H = 512


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6


for M in (98688, 9088, 2624):
    g = torch.Generator().manual_seed(0)
    for kind, act in ((0, K.ACT_SILU), (1, K.ACT_RELU)):
        hid = H if kind == 0 else 2 * H
        kin = 2 * H if kind == 0 else H
        X = torch.randn(M, kin, generator=g).bfloat16().to(dev)
        W1 = (torch.randn(hid, kin, generator=g) / kin ** 0.5).bfloat16().to(dev)
        W2 = (torch.randn(H, hid, generator=g) / hid ** 0.5).bfloat16().to(dev)
        b1, b2 = torch.randn(hid, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
        R = torch.randn(M, H, generator=g).to(dev)
        packs = K.node_chain_pack(W1, W2, kind)
        X1, X2 = (X[:, :H].contiguous(), X[:, H:].contiguous()) if kind == 0 else (X, None)
        t_buf = torch.empty(M, hid, dtype=torch.bfloat16, device=dev)
        o32, o16 = torch.empty(M, H, device=dev), torch.empty(M, H, dtype=torch.bfloat16, device=dev)

        def two():
            K.gemm(X1, W1, bias=b1, A2=X2, act_epi=act, out=t_buf)
            K.gemm(t_buf, W2, bias=b2, residual=R, out=o32, out16=o16)
        us2 = timeit(two)
        us1 = timeit(lambda: K.node_chain_fwd(X1, X2, packs, b1, b2, act, kind, residual=R, want16=True))
        print("NODECHAIN M=%6d kind %d (hidden %4d): two GEMMs %7.1f us   one kernel %7.1f us" % (M, kind, hid, us2, us1))
