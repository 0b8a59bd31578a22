"""PROBE: host side of the stack's fwd+bwd step (the bench's default step function, fresh index tensors per step) under cProfile:
wall per step and the functions the host spends its time in.  usage: stack_hostprof.py [n_prot] [steps]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine
n_prot = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
engine.set_precision("bf16")
model = bench.build_model(512, 4, 1).to(dev)
model.train(False)
batches = []
for r_ in range(2):
    inp = bench.make_batch(64, n_prot, 40, 512, seed=1000 * r_)
    batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()})
params = list(model.parameters())
k = [0]


def step():
    t = dict(batches[k[0] % 2])
    k[0] += 1
    t["batch_id"], t["segment_id"] = t["batch_id"].clone(), t["segment_id"].clone()
    X0 = t["X"].clone()
    for p in params:
        p.grad = None
    X, Hh = model(X0, t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"],
                  t["coord_LAS"])
    loss = (X * X).mean() + (Hh * Hh).mean() * 1e-6
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("wall %.2f ms per step (n_prot %d)" % ((time.time() - t0) / steps * 1e3, n_prot))
t0 = time.time()
for _ in range(steps):
    step()
t_host = time.time() - t0
torch.cuda.synchronize()
print("host-only (no final sync in the window) %.2f ms per step" % (t_host / steps * 1e3))
if len(sys.argv) > 3 and sys.argv[3] == "1":                # backward in the calling thread: cProfile sees the Function.backward bodies
    torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(sys.argv[4] if len(sys.argv) > 4 else "tottime").print_stats(60)
print(s.getvalue()[:14000])
