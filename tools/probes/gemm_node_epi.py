"""The node-MLP first Linear as the step runs it -- [h | agg] (two K = 512 operands), SiLU, value + derivative as bf16 -- against the same
contraction with plainer operands / epilogues: where do 226 us go when the bytes say 120?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fabind_amd import kernels as K
from gemm_bench import timeit
dev = torch.device("cuda:0")
M, H = 98688, 512
g = torch.Generator(device=dev).manual_seed(0)
A1 = torch.randn(M, H, device=dev, generator=g).bfloat16()
A2 = torch.randn(M, H, device=dev, generator=g).bfloat16()
Acat = torch.cat([A1, A2], 1).contiguous()
W = (torch.randn(H, 2 * H, device=dev, generator=g) / 32).bfloat16()
b = torch.randn(H, device=dev, generator=g)
o16 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
d16 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
o32 = torch.empty(M, H, device=dev)
fl = 2.0 * M * H * 2 * H
cases = {
    "cat A, no bias, bf16 out": lambda: K.gemm(Acat, W, out=o16),
    "cat A, bias + silu, bf16 out": lambda: K.gemm(Acat, W, bias=b, act_epi=K.ACT_SILU, out=o16),
    "cat A, bias + silu, bf16 out + derivative": lambda: K.gemm(Acat, W, bias=b, act_epi=K.ACT_SILU, out=o16, out2=d16),
    "A | A2, no bias, bf16 out": lambda: K.gemm(A1, W, A2=A2, out=o16),
    "A | A2, bias + silu, bf16 out + derivative (the step's form)": lambda: K.gemm(A1, W, bias=b, A2=A2, act_epi=K.ACT_SILU, out=o16, out2=d16),
    "cat A, bias, fp32 out": lambda: K.gemm(Acat, W, bias=b, out=o32),
}
for name, fn in cases.items():
    ms = timeit(fn, 30)
    print("%-62s %7.1f us  %6.0f TFLOP/s" % (name, ms * 1e3, fl / ms / 1e9), flush=True)
