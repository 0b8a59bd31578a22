"""Per-phase cycle counts of gcl_edge_fused_bwd_kernel (work-group 0) at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, kernels as K
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
from fabind_amd import _lib
import sys as _s
bm = int(_s.argv[1]) if len(_s.argv) > 1 else 32
variant = int(_s.argv[2]) if len(_s.argv) > 2 else 0      # the phase counters live in the two-tile kernel (0) and the row-wise one (3 / 4)
_lib.load().fabind_gcl_edge_fused_bwd_set_variant(variant)
if variant == 0:
    _lib.load().fabind_gcl_edge_fused_bwd_set_tile(bm)
print("tile =", bm)
K.PROFILE = {}
ms = timeit(run, 5)
for k, evs in K.PROFILE.items():
    print("%-90s %.3f ms" % (k[:90], sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)))
K.PROFILE = None
print("whole backward op (kernel + 2 weight-gradient contractions + sending-side reduction): %.3f ms" % ms)
if len(_s.argv) > 3:
    _lib.load().fabind_gcl_edge_fused_bwd3_set_exp(int(_s.argv[3]))          # skip mask (variant 5; bits 2-3 = weight-stream experiment)
K.EDGE_BWD_TIMES = torch.zeros(12, dtype=torch.int64, device=dev)
run(); torch.cuda.synchronize()
tt = K.EDGE_BWD_TIMES.cpu().tolist()
names = ["P0 gather+silu", "tile stores (5)", "contractions (4)", "P1 epilogue", "P2 epilogue", "P3 dagg loads", "P3 epilogue",
         "P4 epilogue", "P5 gather+dsilu", "P6 column scan"]
n_tiles = (E + bm - 1) // bm
ngr = {0: 512 if bm == 32 else 256, 3: 256, 4: 512, 5: 256}[variant]
per = (n_tiles + ngr - 1) // ngr
tot = sum(tt)
for n, c in zip(names, tt):
    print("%-20s %10.0f cycles/tile  %5.1f%%" % (n, c / per, 100.0 * c / max(tot, 1)))
print("total %.0f cycles/tile, %d tiles per work-group" % (tot / per, per))
