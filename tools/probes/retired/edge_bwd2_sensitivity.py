"""Probe: per-launch time of the row-wise fused edge backward (variant 3 / 4) at the bench shape with parts of its work SKIPPED
(fabind_gcl_edge_fused_bwd2_set_exp bit mask; the results are wrong, only the times mean something): which of the operand stores,
scratch round trips, gathers, dagg loads and the row scan the launch time is sensitive to.  usage: edge_bwd2_sensitivity.py [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, kernels as K, _lib
from gemm_bench import timeit
dev = torch.device("cuda:0")
engine.set_precision("bf16")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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
lib = _lib.load()
lib.fabind_gcl_edge_fused_bwd_set_variant(variant)
names = {0: "everything", 1: "no operand stores", 2: "no scratch stores", 3: "no stores at all", 4: "gathers collapsed", 8: "dagg collapsed",
         16: "no row scan", 28: "gathers + dagg collapsed, no scan", 31: "all of the above"}
for rnd in range(int(os.environ.get("MASK_ROUNDS", "2"))):
    for m, nm in names.items():
        lib.fabind_gcl_edge_fused_bwd2_set_exp(m)
        K.PROFILE = {}
        timeit(run, 5)
        for k, evs in K.PROFILE.items():
            if "bwd" in k:
                print("round %d mask %2d %-36s %.3f ms" % (rnd, m, nm, sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)), flush=True)
        K.PROFILE = None
for rnd in range(2):
    for skew in (0,):            # cycles per step of the start skew (16 steps)
        for m in ((0, 1, 16, 17) if variant >= 5 else (0, 1, 2, 16, 19)):
            (lib.fabind_gcl_edge_fused_bwd3_set_exp if variant >= 5 else lib.fabind_gcl_edge_fused_bwd2_set_exp)(m | (skew << 8))
            K.PROFILE = {}
            timeit(run, 5)
            for k, evs in K.PROFILE.items():
                if "bwd" in k:
                    print("round %d skew %5d mask %d  %.3f ms" % (rnd, skew, m, sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)), flush=True)
            K.PROFILE = None
lib.fabind_gcl_edge_fused_bwd2_set_exp(0)
lib.fabind_gcl_edge_fused_bwd_set_variant(0)
