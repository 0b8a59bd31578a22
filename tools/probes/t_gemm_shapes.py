import sys
sys.path.insert(0, '/root/repo')
import torch
from fabind_amd import kernels as K
dev = torch.device("cuda:0")
for (M, N, Kd) in ((1556480, 1088, 1088), (1556480, 512, 1088), (1556480, 320, 320), (1556480, 128, 320), (200000, 1088, 1088)):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, out=out)
    torch.cuda.synchronize()
    ref = torch.relu(A[:256].float() @ W.float().T + b)
    print(M, N, Kd, "ok, err", float((out[:256].float() - ref).abs().max()), float((out[-256:].float() - torch.relu(A[-256:].float() @ W.float().T + b)).abs().max()))
    del A, W, out
