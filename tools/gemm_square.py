import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabind_amd import kernels as K, _lib
from gemm_bench import timeit
dev = torch.device("cuda:0")
lib = _lib.load()
for (M, N, Kd) in [(4096, 4096, 4096), (8192, 8192, 8192), (1556480, 512, 512), (1556480, 512, 2048), (98688, 1024, 512), (98688, 512, 1024)]:
    A = torch.randn(M, Kd, device=dev).bfloat16(); W = torch.randn(N, Kd, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for cfg in (3, 6, 4):
        lib.fabind_gemm_set_config(cfg)
        ms = timeit(lambda: K.gemm(A, W, out=o), n=5)
        print("M=%d N=%d K=%d cfg %d: %.3f ms %.1f TFLOP/s" % (M, N, Kd, cfg, ms, 2.0 * M * N * Kd / ms / 1e9))
