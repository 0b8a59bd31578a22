"""PROBE: node-level GEMM shapes (M = 98,688 rows) under occupancy-bounded builds of the LDS-DMA pipelined kernel.
cfg 3  = 256x128 tile, 8 waves, 213 VGPRs -> ONE work-group per CU                                  [shipped]
cfg 6  = 128x128 tile, 4 waves, 213 VGPRs + 64 AGPRs -> 1 wave/SIMD, one work-group per CU
cfg 16 = cfg 6 compiled with a 2-waves-per-SIMD bound: 213 VGPRs, no AGPRs, no scratch -> two work-groups per CU
cfg 17 = cfg 6 with a 3-waves-per-SIMD bound: 168 VGPRs + 176 B scratch -> three per CU
cfg 13 = cfg 3 with a 4-waves-per-SIMD bound: 128 VGPRs + 336 B scratch -> two per CU
Epilogues timed: bias -> fp32 (the dominant one), bias + residual -> fp32, plain -> bf16."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fabind_amd import _lib, kernels as K  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)


def med(fn, n=30):
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


M = 98688
CFGS = (3, 13, 6)          # 16 / 17 (128x128 tile under 2- / 3-waves-per-SIMD bounds) were probe-only builds: results in profiles/r01_bench_noise.txt
for rnd in range(2):
    for (N, Kd) in ((512, 512), (1024, 512), (512, 1024), (1536, 512)):
        A = torch.randn(M, Kd, device=dev).bfloat16()
        W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev)
        o32, o16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ref = None
        for name, fn in (("bias->f32", lambda: K.gemm(A, W, bias=b, out=o32)),
                         ("bias+res->f32", lambda: K.gemm(A, W, bias=b, residual=R, out=o32)),
                         ("->bf16", lambda: K.gemm(A, W, out=o16))):
            line = "NODEOCC r%d N=%4d K=%4d %-14s" % (rnd, N, Kd, name)
            base = None
            for cfg in CFGS:
                lib.fabind_gemm_set_config(cfg)
                t = med(fn)
                out = (o32 if "f32" in name else o16).clone()
                if cfg == 3:
                    base, ref = t, out
                same = bool(torch.equal(out, ref))
                line += "  cfg%-2d %6.1f us (%+5.1f%%)%s" % (cfg, t, 100.0 * (t / base - 1.0), "" if same else " !=")
            print(line)
lib.fabind_gemm_set_config(K.GEMM_DEFAULT_CONFIG)
