"""Node-level GEMM shapes of the headline step under the default tile config: bf16 output, fp32 output with bias, fp32 residual epilogue --
for A/B builds of csrc/gemm.hip (FABIND_LIB)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fabind_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
print("== node GEMMs, lib=[%s]" % os.environ.get("FABIND_LIB", ""))
M = 98688
for (N, Kd) in ((512, 512), (1024, 512), (512, 1024), (512, 1536)):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    line = "M=%6d N=%4d K=%4d:" % (M, N, Kd)
    for name, fn in (("bf16 out", lambda: K.gemm(A, W, out=o16)), ("bias -> f32", lambda: K.gemm(A, W, bias=b, out=o32)),
                     ("bias + relu -> bf16", lambda: K.gemm(A, W, bias=b, act_epi=K.ACT_RELU, out=o16)),
                     ("bias + residual -> f32 (+bf16 copy)", lambda: K.gemm(A, W, bias=b, residual=res, out=o32, out16=o16))):
        try:
            for _ in range(3):
                fn()
        except TypeError:
            continue
        ts = []
        for _ in range(10):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts.sort()
        line += "  %s %.0f us" % (name, ts[len(ts) // 2])
    print(line)
