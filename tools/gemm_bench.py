"""Micro-benchmark of the GEMM kernel variants on the GPU (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fabind_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    M, N, Kd = 1556480, 512, 512
    if len(sys.argv) > 3:
        M, N, Kd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    A16 = torch.randn(M, Kd, device=dev).bfloat16()
    A32 = torch.randn(M, Kd, device=dev)
    W16 = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    u = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * Kd
    cases = {
        "bf16A plain -> bf16": lambda: K.gemm(A16, W16, out=o16),
        "bf16A bias+silu -> bf16": lambda: K.gemm(A16, W16, bias=b, act_epi=K.ACT_SILU, out=o16),
        "bf16A bias+silu rowdot only": lambda: K.gemm(A16, W16, bias=b, act_epi=K.ACT_SILU, dotvec=u, want_out=False),
        "bf16A plain -> f32": lambda: K.gemm(A16, W16, out=o32),
        "bf16A silu-prologue (old kernel) -> bf16": lambda: K.gemm(A16, W16, act_pro=K.ACT_SILU, out=o16),
        "f32A plain -> f32": lambda: K.gemm(A32, W16, out=o32),
        "f32A plain -> bf16": lambda: K.gemm(A32, W16, out=o16),
    }
    import ctypes
    from fabind_amd import _lib
    lib = _lib.load()
    ref = (A16[:512].float() @ W16.float().T)
    for cfg in (3, 103):
        lib.fabind_gemm_set_config(3)
        lib.fabind_gemm_set_persistent(1 if cfg > 100 else 0)
        K.gemm(A16, W16, out=o16)
        err = float((o16[:512].float() - ref).abs().max())
        for name in ("bf16A plain -> bf16", "bf16A bias+silu -> bf16", "bf16A bias+silu rowdot only"):
            ms = timeit(cases[name])
            print("cfg %d %-40s %8.3f ms  %7.1f TFLOP/s  (err %.3g)" % (cfg, name, ms, fl / ms / 1e9, err))
    lib.fabind_gemm_set_config(0)            # the register-staged kernel for the cases below
    for name in ("bf16A silu-prologue (old kernel) -> bf16", "f32A plain -> f32", "f32A plain -> bf16"):
        ms = timeit(cases[name])
        print("%-45s %8.3f ms  %7.1f TFLOP/s" % (name, ms, fl / ms / 1e9))


if __name__ == "__main__":
    main()
