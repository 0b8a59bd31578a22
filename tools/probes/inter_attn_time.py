"""PROBE: stand-alone time of the inter-edge attention forward / backward at the bench shape, on the real inter graph and on the
same graph with the ligand-side rows emptied (only rows of <= a few edges left): how much of the launch is the long rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine, kernels as K, ops
dev = torch.device("cuda:0")
engine.set_precision("bf16")
n_prot = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
inp = bench.make_batch(64, n_prot, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
lay = engine.Layout(t["batch_id"], t["segment_id"])
br = t["compound_edge_index"][0].to(torch.int32); bc_ = t["compound_edge_index"][1].to(torch.int32)
g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
H, N = 512, lay.N
deg = (g.rp_int[1:] - g.rp_int[:-1]).float()
print("N %d, E_int %d, rows with edges %d, max degree %d, rows with > 8 edges %d (%.0f %% of the edges)" % (
    N, g.E_int, int((deg > 0).sum()), int(deg.max()), int((deg > 8).sum()), 100.0 * float(deg[deg > 8].sum()) / g.E_int))
gen = torch.Generator().manual_seed(0)
qkv = torch.randn(N, 3 * H, generator=gen).to(dev)
cv = torch.randn(N, H, generator=gen).to(dev)
h = torch.randn(N, H, generator=gen).to(dev)
x = t["X"][:, 0].contiguous()
w = [torch.randn(H, generator=gen).to(dev) * 0.1 for _ in range(4)]
d, rhohat = ops.edge_geom(x, g.row_int, g.col_int, g.rp_int, lay, g.int_by_col)
bias = torch.randn(g.red_idx.max().item() + 1, 1, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6


def fwd():
    with torch.no_grad():
        ops.inter_attn(qkv, cv, H, h, x, d, rhohat, g, bias, w[0], w[1], w[2], w[3], 2.0)


print("forward, no grad: %.1f us" % timeit(fwd))
qg, cg, hg = qkv.clone().requires_grad_(True), cv.clone().requires_grad_(True), h.clone().requires_grad_(True)


def fwdbwd():
    ho, xo, al = ops.inter_attn(qg, cg, H, hg, x, d, rhohat, g, bias, w[0], w[1], w[2], w[3], 2.0)
    (ho.sum() + xo.sum()).backward()
    qg.grad = cg.grad = hg.grad = None


print("forward + backward: %.1f us" % timeit(fwdbwd))
K.PROFILE = None
