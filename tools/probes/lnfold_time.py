"""Stand-alone time of FABind+'s LayerNorm-folded first edge Linear (csrc/norm.hip: edge_lnfold forward / adjoint) at the two shapes of a
training step: the hidden-128 pocket model on the whole-protein graph (E = 1.54 M, Kp = 320) and the hidden-512 stack on the pocket
(E = 226 k, Kp = 1088).  FABIND_LIB selects an A/B build (tools/probes/build_variants.sh)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fabind_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
print("== lnfold kernels, lib=[%s]" % os.environ.get("FABIND_LIB", ""))
for N, H, E, Kp in ((98688, 128, 1539196, 320), (18053, 512, 225774, 1088)):
    g = torch.Generator().manual_seed(0)
    row = torch.sort(torch.randint(0, N, (E,), generator=g)).values.to(torch.int32).to(dev)
    col = torch.randint(0, N, (E,), generator=g).to(torch.int32).to(dev)
    AB = torch.randn(N, 2 * Kp, generator=g).to(dev).to(torch.bfloat16)
    stat = torch.stack([0.1 * torch.randn(N, generator=g), H * (0.5 + torch.rand(N, generator=g))], 1).contiguous().to(dev)
    rho = torch.rand(E, generator=g).to(dev)
    vec = lambda: torch.randn(Kp, generator=g).to(dev)
    w_r, c_r, c_c, dvec = vec(), vec(), vec(), vec()
    dout = torch.randn(E, Kp, generator=g).to(dev).to(torch.bfloat16)
    for name, fn in (("forward  p=0.1", lambda: K.edge_lnfold(AB, Kp, H, row, col, rho, stat, 1e-5, w_r, c_r, c_c, dvec, 0.1, 7)),
                     ("forward  p=0  ", lambda: K.edge_lnfold(AB, Kp, H, row, col, rho, stat, 1e-5, w_r, c_r, c_c, dvec, 0.0, 0)),
                     ("backward p=0.1", None)):
        if fn is None:
            out = K.edge_lnfold(AB, Kp, H, row, col, rho, stat, 1e-5, w_r, c_r, c_c, dvec, 0.1, 7)
            fn = lambda: K.edge_lnfold_bwd(AB, Kp, H, row, col, rho, stat, 1e-5, w_r, c_r, c_c, out, dout, 0.1)
        for _ in range(3):
            fn()
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print("H=%-4d E=%-8d Kp=%-5d %s  median %8.1f us  min %8.1f" % (H, E, Kp, name, ts[len(ts) // 2], ts[0]))
