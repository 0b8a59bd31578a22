"""One variant of the fused edge backward at the bench shape, 6 launches (for rocprofv3 --pmc): argv[1] = variant (0 .. 5), argv[2] = experiment mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine, kernels as K, _lib
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
_lib.load().fabind_gcl_edge_fused_bwd_set_variant(int(sys.argv[1]))
if len(sys.argv) > 2:                                      # experiment mask of the store-wave kernel (32 = non-temporal operand stores)
    _lib.load().fabind_gcl_edge_fused_bwd3_set_exp(int(sys.argv[2]))
W2p, Wcp = K.pack_frag(W2), K.pack_frag(Wc)
for _ in range(6):
    K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm)
    K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N)
torch.cuda.synchronize()
