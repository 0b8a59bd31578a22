"""Probe: the fused edge backward / forward at H = 128 on the headline graph (the pocket model of the full IaBNet: 1 layer + out layer on the
whole 1500-residue proteins, E = 1.5 M edges) across the kernel variants; and the sending-side segment sum at the pocket-sized shape
(plain vs permuted rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from fabind_amd import engine, kernels as K, _lib
dev = torch.device("cuda:0")
engine.set_precision("bf16")
lib = _lib.load()
lib.fabind_gcl_edge_fused_bwd_set_variant.argtypes = [__import__("ctypes").c_int]
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def graph(n_prot):
    inp = bench.make_batch(64, n_prot, 40, 512, 0)
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    lay = engine.Layout(t["batch_id"], t["segment_id"])
    br = t["compound_edge_index"][0].to(torch.int32); bc_ = t["compound_edge_index"][1].to(torch.int32)
    return lay, engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
lay, g = graph(1500)
E, N = g.E_ctx, lay.N
colptr, perm = g.ctx_by_col()
for H in (128, 512):
    AB = torch.randn(N, 2 * H, device=dev).bfloat16(); rh = torch.rand(E, device=dev)
    w_r, b2, bc, w3 = [torch.randn(H, device=dev) * 0.5 for _ in range(4)]
    W2 = (torch.randn(H, H, device=dev) / H ** 0.5).bfloat16(); Wc = (torch.randn(H, H, device=dev) / H ** 0.5).bfloat16()
    ds = torch.randn(E, device=dev); dagg = torch.randn(N, H, device=dev)
    for v in (5, 0, 1, 3, 4):
        lib.fabind_gcl_edge_fused_bwd_set_variant(v)
        try:
            us = timeit(lambda: K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm, rowptr=g.rp_ctx))
            print("H=%d E=%d edge backward (+ segment sum + 2 weight-gradient contractions) variant %d: %.0f us" % (H, E, v, us), flush=True)
        except Exception as e:
            print("H=%d variant %d: %s" % (H, v, str(e)[:100]))
    lib.fabind_gcl_edge_fused_bwd_set_variant(5)
    W2p, Wcp = K.pack_frag(W2), K.pack_frag(Wc)
    print("H=%d forward: %.0f us" % (H, timeit(lambda: K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N, rowptr=g.rp_ctx))))
# segment sum at the pocket-sized shape
lay, g = graph(100)
E, N, H = g.E_ctx, lay.N, 512
colptr, perm = g.ctx_by_col()
Z = torch.randn(E, H, device=dev).bfloat16()
out16 = torch.empty(N, 2 * H, dtype=torch.bfloat16, device=dev)
deg = (colptr[1:] - colptr[:-1]).float()
print("pocket graph: N=%d E=%d; sending-side degree mean %.1f, max %d, rows > 32: %d, > 128: %d" % (N, E, deg.mean(), int(deg.max()), int((deg > 32).sum()), int((deg > 128).sum())))
print("segment_sum permuted rows (by sender): %.1f us" % timeit(lambda: K.segment_sum(Z, colptr, N, eidx=perm, out16=out16[:, H:]), 20))
print("segment_sum plain rows (by receiver): %.1f us" % timeit(lambda: K.segment_sum(Z, g.rp_ctx, N, out16=out16[:, H:]), 20))
ident = torch.arange(E, device=dev, dtype=torch.int32)
print("segment_sum identity permutation: %.1f us" % timeit(lambda: K.segment_sum(Z, g.rp_ctx, N, eidx=ident, out16=out16[:, H:]), 20))
print("segment_sum fp32 out: %.1f us" % timeit(lambda: K.segment_sum(Z, colptr, N, eidx=perm), 20))
