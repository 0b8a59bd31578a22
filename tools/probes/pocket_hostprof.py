"""Probe: the pocket-realistic stack step (B=64, 100 protein / 40 ligand nodes, H=512, fwd+bwd, n_iter=1): wall per step, GPU kernel time
per step (torch.profiler), launches per step, and the cProfile top of the host side.  usage: pocket_hostprof.py [n_prot] [batch]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine
n_prot = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
engine.set_precision("bf16")
model = bench.build_model(512, 4, 1).to(dev); model.eval()
inp = bench.make_batch(B, n_prot, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    X, Hh = model(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
    ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print("wall per step %.2f ms = %.0f complexes/s" % (wall * 1e3, B / wall))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print("GPU kernel time of one step: %.2f ms in %d launches" % (sum(e.device_time for e in ev) / 1e3, len(ev)))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28); print(st.getvalue()[:5500])
