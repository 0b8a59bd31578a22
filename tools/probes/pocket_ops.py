"""Probe: device kernels of one pocket-shape step (B=64, 100/40 nodes) by name: count and time -- what the launch budget is spent on."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine
from torch.profiler import profile, ProfilerActivity
n_prot = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
engine.set_precision("bf16")
model = bench.build_model(512, 4, 1).to(dev); model.eval()
inp = bench.make_batch(64, n_prot, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    X, Hh = model(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
    ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
cnt, tm = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        n = e.name.split("(")[0][:70]
        cnt[n] += 1; tm[n] += e.device_time
print("%d launches, %.2f ms" % (sum(cnt.values()), sum(tm.values()) / 1e3))
for n, c in cnt.most_common(32):
    print("%5d x %8.1f us total  %s" % (c, tm[n], n))
