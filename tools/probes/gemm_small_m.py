"""fabind_gemm on few-row operands (pocket-sized batches, ligand rows): 256x128 tiles (FABIND small-M switch off) against the switch."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from fabind_amd import kernels as K, _lib
from gemm_bench import timeit
dev = torch.device("cuda:0")
lib = _lib.load()
for M in (2624, 9088, 16384):
    for (N, Kd) in ((512, 512), (1024, 512), (512, 1024), (512, 128)):
        A = torch.randn(M, Kd, device=dev).bfloat16()
        W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        o32 = torch.empty(M, N, device=dev)
        line = "M=%6d N=%4d K=%4d:" % (M, N, Kd)
        outs = []
        for sm in (0, 192, 100000):
            lib.fabind_gemm_set_small_m(sm)
            ms = timeit(lambda: K.gemm(A, W, bias=b, out=o32), 50)
            outs.append(o32.clone())
            line += "  small_m=%d: %.1f us" % (sm, ms * 1e3)
        print(line, " equal:", bool((outs[0] == outs[2]).all()))
lib.fabind_gemm_set_small_m(192)
