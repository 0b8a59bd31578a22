"""Probe: host-side profile of the pocket-sized fwd+bwd step INCLUDING the backward pass's Python (autograd runs it in a device worker
thread, which cProfile does not see: multithreading is switched off here so that everything runs in the calling thread).
usage: pocket_hostprof2.py [n_prot] [batch]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine, kernels as K
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
def wall(n=10):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
w = wall()
print("multithreaded backward: wall per step %.2f ms = %.0f complexes/s" % (w * 1e3, B / w))
K.TN_STATS.update(queued=0, flushes=0, direct=0)
step(); torch.cuda.synchronize()
print("weight-gradient queue per step:", dict(K.TN_STATS))
# host-only time of a step: launches queued, no waiting (the device is far behind after a few steps; measure the host side alone)
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); h = time.perf_counter() - t0
torch.cuda.synchronize()
print("host time of one step (launch queue empty at its start): %.2f ms" % (h * 1e3))
torch.autograd.set_multithreading_enabled(False)
w = wall()
print("single-threaded backward: wall per step %.2f ms = %.0f complexes/s" % (w * 1e3, B / w))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumulative"):
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats(key).print_stats(45); print(st.getvalue()[:9000])
