"""HBM-roofline check of the standalone message-passing kernels (segment-sum, gather) at the bench shape."""
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
br = t["compound_edge_index"][0].to(torch.int32); bc = t["compound_edge_index"][1].to(torch.int32)
g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc, lay.ranges(br), 1.6, 2.0)
E, N, H = g.E_ctx, lay.N, 512
Z = torch.randn(E, H, device=dev).bfloat16()
AB = torch.randn(N, 2 * H, device=dev)
rh = torch.rand(E, device=dev); w_r = torch.randn(H, device=dev)
ms = timeit(lambda: K.segment_sum(Z, g.rp_ctx, N))
by = E * H * 2 + E * 4 + N * H * 4
print("segment_sum      E=%d H=%d: %.3f ms  %.2f TB/s algorithmic (%.0f%% of 8 TB/s)" % (E, H, ms, by / ms / 1e9, by / ms / 1e9 / 8 * 100))
colptr, perm = g.ctx_by_col()
ms = timeit(lambda: K.segment_sum(Z, colptr, N, eidx=perm))
print("segment_sum(perm) (sending side): %.3f ms  %.2f TB/s" % (ms, by / ms / 1e9))
ms = timeit(lambda: K.gcl_pre(AB, H, g.row_ctx, g.col_ctx, rh, w_r, torch.bfloat16, K.ACT_SILU))
by2 = E * H * 2 + E * 12 + N * 2 * H * 4
print("gcl_pre fp32 AB   : %.3f ms  %.2f TB/s algorithmic (compulsory: write E*H*2, read AB once)" % (ms, by2 / ms / 1e9))
AB16 = AB.bfloat16()
ms = timeit(lambda: K.gcl_pre(AB16, H, g.row_ctx, g.col_ctx, rh, w_r, torch.bfloat16, K.ACT_SILU))
by2 = E * H * 2 + E * 12 + N * 2 * H * 2
print("gcl_pre bf16 AB   : %.3f ms  %.2f TB/s algorithmic" % (ms, by2 / ms / 1e9))
ms = timeit(lambda: K.gcl_pre(AB16, H, g.row_ctx, g.col_ctx, rh, w_r, torch.bfloat16, K.ACT_SILU, want_dact=True))
by2 = 2 * E * H * 2 + E * 12 + N * 2 * H * 2
print("gcl_pre bf16 +dact: %.3f ms  %.2f TB/s algorithmic (training forward: value + stored derivative)" % (ms, by2 / ms / 1e9))
