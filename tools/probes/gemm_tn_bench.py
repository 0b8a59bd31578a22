"""Stand-alone timing of the TN (weight-gradient) contraction at the step's shapes, with a check against torch.
usage: python tools/probes/gemm_tn_bench.py [reps]   (run from a tree root: times THAT tree's library)"""
import sys
import os
sys.path.insert(0, os.getcwd())
import torch
from fabind_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:                      # probe mask (fabind_gemm_tn_set_exp); results are then garbage, timings only
    from fabind_amd import _lib
    _lib.load().fabind_gemm_tn_set_exp(int(sys.argv[2]))
    print("probe mask", sys.argv[2])
if len(sys.argv) > 3:
    from fabind_amd import _lib
    _lib.load().fabind_gemm_tn_set_waves(int(sys.argv[3]))
    print("waves", sys.argv[3])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [(1539196, 512, 512), (98688, 512, 512), (98688, 1024, 512), (98688, 1536, 512), (98688, 512, 1024), (78837, 1024, 576),
          (98688, 256, 512), (2624, 512, 512)]
for E, M, N in SHAPES:
    Y = (torch.randn(E, M, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    X = (torch.randn(E, N, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    out = K.gemm_tn(Y, X)
    if E <= 100000:
        ref = Y.float().t() @ X.float()
        err = float((out - ref).abs().max() / ref.abs().max())
    else:
        ref = Y[:65536].float().t() @ X[:65536].float()
        err = float((K.gemm_tn(Y[:65536], X[:65536]) - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        K.gemm_tn(Y, X)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        K.gemm_tn(Y, X)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    print("E=%8d M=%5d N=%5d  %8.1f us (incl. split reduction)  %7.1f TFLOP/s  %5.2f TB/s operands  rel err %.2e"
          % (E, M, N, us, 2.0 * E * M * N / us / 1e6, (E * (M + N) * 2) / us / 1e6, err), flush=True)
    del Y, X
