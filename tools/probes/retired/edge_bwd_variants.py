"""Probe: per-launch time of the fused edge backward at the bench shape, two-tile kernel (variant 0) vs single in-place tile
with two work-groups per CU (variant 1), 3 rounds interleaved; also the forward kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, kernels as K, _lib
from gemm_bench import timeit
dev = torch.device("cuda:0")
engine.set_precision("bf16")
inp = bench.make_batch(64, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
lay = engine.Layout(t["batch_id"], t["segment_id"])
br = t["compound_edge_index"][0].to(torch.int32); bc_ = t["compound_edge_index"][1].to(torch.int32)
g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
E, N, H = g.E_ctx, lay.N, 512
AB = torch.randn(N, 2 * H, device=dev).bfloat16()
rh = torch.rand(E, device=dev)
w_r, b2, bc, w3 = [torch.randn(H, device=dev) * 0.5 for _ in range(4)]
W2 = (torch.randn(H, H, device=dev) / H ** 0.5).bfloat16(); Wc = (torch.randn(H, H, device=dev) / H ** 0.5).bfloat16()
ds = torch.randn(E, device=dev); dagg = torch.randn(N, H, device=dev)
colptr, perm = g.ctx_by_col()
run = lambda: K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm)
W2p, Wcp = K.pack_frag(W2), K.pack_frag(Wc)
fwd = lambda: K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N)
lib = _lib.load()
outs = {}
fouts = {}
for rnd in range(3):
    for v in (0, 3, 5):
        lib.fabind_gcl_edge_fused_bwd_set_variant(v)
        K.PROFILE = {}
        timeit(run, 5)
        for k, evs in K.PROFILE.items():
            if "bwd" in k:
                print("round %d variant %d  %-60s %.3f ms" % (rnd, v, k[:60], sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)), flush=True)
        K.PROFILE = None
        outs[v] = run()
    for fv in (0, 1):
        lib.fabind_gcl_edge_fused_set_variant(fv)
        K.PROFILE = {}
        timeit(fwd, 5)
        for k, evs in K.PROFILE.items():
            print("round %d forward %d  %-60s %.3f ms" % (rnd, fv, k[:60], sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)), flush=True)
        K.PROFILE = None
        fouts[fv] = fwd()
lib.fabind_gcl_edge_fused_bwd_set_variant(1)
for ng in (128, 256, 384, 512):                       # does a second work-group per CU add throughput?
    K.EDGE_BWD_GROUPS = ng
    K.PROFILE = {}
    timeit(run, 5)
    for k, evs in K.PROFILE.items():
        if "bwd" in k:
            print("variant 1 with %d persistent work-groups: %.3f ms" % (ng, sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)), flush=True)
    K.PROFILE = None
K.EDGE_BWD_GROUPS = 0
lib.fabind_gcl_edge_fused_bwd_set_variant(0)
for v in (3, 5):
    for n, a, b in zip(("dAB", "drh", "dw_r", "dW2", "db2", "dWc", "dbc", "dw3"), outs[0], outs[v]):
        print("variant %d vs 0  %-5s max rel diff %.2e" % (v, n, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))
for n, a, b in zip(("agg", "s"), fouts[0], fouts[1]):
    print("forward variant 1 vs 0  %-4s max rel diff %.2e" % (n, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))
