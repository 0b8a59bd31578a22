"""One shape of the split-bf16 GEMM, 10 launches (for rocprofv3 --pmc): argv = M N K [tile rows / 64]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabind_amd import _lib, config, kernels as K
dev = torch.device("cuda:0")
M, N, Kd = (int(v) for v in sys.argv[1:4])
config.set_precision("bf16x3")
if len(sys.argv) > 4:
    _lib.load().fabind_gemm_set_x3_tile(int(sys.argv[4]))
A = torch.randn(M, Kd, device=dev)
W = torch.randn(N, Kd, device=dev) / Kd ** 0.5
b = torch.randn(N, device=dev)
o = torch.empty(M, N, device=dev)
for _ in range(10):
    K.gemm(A, W, bias=b, out=o)
config.set_precision("bf16")
A16, W16 = A.bfloat16(), W.bfloat16()
for _ in range(10):
    K.gemm(A16, W16, bias=b, out=o)
torch.cuda.synchronize()
