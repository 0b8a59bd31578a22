"""Node-level GEMM shapes (M ~ 98k rows, K/N in {512,1024,1536}): time vs M (tile quantisation) and tile config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabind_amd import kernels as K, _lib
from gemm_bench import timeit
dev = torch.device("cuda:0")
lib = _lib.load()
for (N, Kd) in ((512, 512), (1024, 512), (512, 1024), (1536, 512)):
    for M in (98304, 98688, 196608):
        A = torch.randn(M, Kd, device=dev).bfloat16()
        W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o32 = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * Kd
        line = "M=%6d N=%4d K=%4d:" % (M, N, Kd)
        for cfg in (13, 3, 6, 1, 7):
            lib.fabind_gemm_set_config(cfg)
            ms = timeit(lambda: K.gemm(A, W, out=o16), 20)
            line += "  cfg%d bf16 %.0f us (%.0f TF)" % (cfg, ms * 1e3, fl / ms / 1e9)
        lib.fabind_gemm_set_config(K.GEMM_DEFAULT_CONFIG)
        ms = timeit(lambda: K.gemm(A, W, bias=b, out=o32), 20)
        line += "  | default cfg bias->f32 %.0f us (%.0f TF)" % (ms * 1e3, fl / ms / 1e9)
        print(line)
