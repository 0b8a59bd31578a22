"""Stand-alone time per launch of the default fused edge backward and forward at the bench shape (run from a tree root: times THAT
tree's library).  usage: python tools/probes/edge_bwd_time.py [reps]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from fabind_amd import engine, kernels as K
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
engine.set_precision("bf16")
inp = bench.make_batch(64, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
lay = engine.Layout(t["batch_id"], t["segment_id"])
br = t["compound_edge_index"][0].to(torch.int32); bc_ = t["compound_edge_index"][1].to(torch.int32)
g = engine.Graph(lay, t["X"][:, 0].contiguous(), br, bc_, lay.ranges(br), 1.6, 2.0)
E, N, H = g.E_ctx, lay.N, 512
gen = torch.Generator(device=dev).manual_seed(0)
AB = torch.randn(N, 2 * H, device=dev, generator=gen).bfloat16()
rh = torch.rand(E, device=dev, generator=gen)
w_r, b2, bc, w3 = [torch.randn(H, device=dev, generator=gen) * 0.5 for _ in range(4)]
W2 = (torch.randn(H, H, device=dev, generator=gen) / H ** 0.5).bfloat16(); Wc = (torch.randn(H, H, device=dev, generator=gen) / H ** 0.5).bfloat16()
ds = torch.randn(E, device=dev, generator=gen); dagg = torch.randn(N, H, device=dev, generator=gen)
colptr, perm = g.ctx_by_col()
W2p, Wcp = K.pack_frag(W2), K.pack_frag(Wc)
bwd = lambda: K.gcl_edge_fused_bwd(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2, b2, Wc, bc, w3, ds, dagg, colptr, perm)
fwd = lambda: K.gcl_edge_fused(AB, H, g.row_ctx, g.col_ctx, rh, w_r, W2p, b2, Wcp, bc, w3, N)
out = bwd()
chk = [float(o.float().abs().sum()) for o in out if torch.is_tensor(o)]
for name, fn in (("backward", bwd), ("forward", fwd)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s E=%d: %.3f ms per launch" % (name, E, e0.elapsed_time(e1) / reps), flush=True)
K.PROFILE = {}
for _ in range(reps):
    bwd()
torch.cuda.synchronize()
for label, ev in K.PROFILE.items():
    print("  %-60s %8.1f us avg over %d" % (label[:60], sum(a.elapsed_time(b) for a, b, _ in ev) * 1e3 / len(ev), len(ev)))
K.PROFILE = None
print("backward output checksums", ["%.6e" % c for c in chk])
