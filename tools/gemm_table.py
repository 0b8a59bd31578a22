"""Per-shape table of every profiled launch in one fwd+bwd bench step (label, calls, total ms, TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, kernels as K
dev = torch.device("cuda:0")
engine.set_precision("bf16")
model = bench.build_model(512, 4, 1).to(dev); model.eval()
inp = bench.make_batch(64, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    X, Hh = model(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
    ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
step(); step(); torch.cuda.synchronize()
K.PROFILE = {}
step(); torch.cuda.synchronize()
rows = []
for label, evs in K.PROFILE.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in evs)
    rows.append((ms, len(evs), evs[0][2], label))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("profiled launches: %.1f ms" % tot)
for ms, n, fl, label in rows[:45]:
    print("%8.3f ms %4d x  %7.1f TF/s  %s" % (ms, n, fl * n / ms / 1e9 if ms > 0 else 0, label[:110]))
