"""PROBE: the split-bf16 ('bf16x3') NT GEMM on the node- and edge-level shapes of the headline step, both tile heights, the epilogues
the fp32-storage path runs (bias -> fp32, bias + residual, SiLU + stored derivative, activation adjoint through an fp32 aux tile),
next to the bf16 LDS-DMA kernel and the exact-fp32 MFMA kernel on the same shape.  Prints microseconds and USEFUL TFLOP/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fabind_amd import _lib, config, kernels as K  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def med(fn, n=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


shapes = [(98688, 512, 512), (98688, 1024, 512), (98688, 512, 1024), (98688, 1536, 512), (98688, 512, 128), (2624, 512, 512)]
if len(sys.argv) > 1 and sys.argv[1] == "edge":
    shapes = [(1539196, 512, 512)]
for (M, N, Kd) in shapes:
    A = torch.randn(M, Kd, device=dev)
    W = torch.randn(N, Kd, device=dev) / Kd ** 0.5
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev)
    o, o2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    A16, W16 = A.bfloat16(), W.bfloat16()
    fl = 2.0 * M * N * Kd
    forms = (("bias->f32", lambda: K.gemm(A, W, bias=b, out=o)),
             ("bias+res", lambda: K.gemm(A, W, bias=b, residual=R, out=o)),
             ("silu+deriv", lambda: K.gemm(A, W, bias=b, act_epi=K.ACT_SILU, out=o, out2=o2)),
             ("x relu'(aux)", lambda: K.gemm(A, W, aux=aux, dact=K.ACT_RELU, out=o)))
    for name, fn in forms:
        line = "X3GEMM M=%7d N=%4d K=%4d %-13s" % (M, N, Kd, name)
        config.set_precision("bf16x3")
        for wm in (2, 4):
            lib.fabind_gemm_set_x3_tile(wm)
            t = med(fn)
            line += "  x3/%d %7.1f us %6.1f TF" % (64 * wm, t, fl / t / 1e6)
        lib.fabind_gemm_set_x3_tile(2)
        if name == "bias->f32":
            config.set_precision("fp32")
            t = med(fn, 5)
            line += "  fp32-mfma %8.1f us" % t
            config.set_precision("bf16")
            t = med(lambda: K.gemm(A16, W16, bias=b, out=o))
            line += "  bf16 %6.1f us %6.1f TF" % (t, fl / t / 1e6)
        print(line, flush=True)
config.set_precision("bf16")
