"""Fused edge kernels at the bench shape (64 x 1500/40, H = 512): live HIP-event time per launch of the plain / saving forward and of the
recompute / saved-forward backward kernels (kernels._profiled labels), n launches each.  FABIND_LIB selects an A/B build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine, kernels as K
dev = torch.device("cuda:0")
engine.set_precision("bf16")
n_prot = int(os.environ.get("N_PROT", "1500"))
inp = bench.make_batch(64, n_prot, 40, 512, 0)
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
W2p, Wcp = K.pack_frag(W2), K.pack_frag(Wc)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
which = sys.argv[2] if len(sys.argv) > 2 else "all"
for rep in range(2):
    K.PROFILE = {} if rep else None
    for _ in range(n if rep else 2):
        if which in ("all", "old"):
            K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N, rowptr=g.rp_ctx)
            K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm, dab_bf16=True, rowptr=g.rp_ctx)
        if which in ("all", "new"):
            fw = K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N, rowptr=g.rp_ctx, save=True)
            K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm, dab_bf16=True, rowptr=g.rp_ctx, saved=fw[-1])
            del fw
    torch.cuda.synchronize()
for label, evs in K.PROFILE.items():
    if not label.startswith("gcl_edge"):
        continue
    ts = sorted(s_.elapsed_time(e_) * 1e3 for s_, e_, _ in evs)
    print("%-60s E=%d  median %7.1f us  min %7.1f  (%d launches)" % (label.split(" E=")[0] + (" saving" if "saving" in label else ""), E, ts[len(ts) // 2], ts[0], len(ts)))
