"""Which torch (non-fabind) device ops run in one fwd+bwd step: name, input shapes, count, device time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("aten device time total %.2f ms" % sum(r[0] for r in rows))
for r in rows[:40]:
    print("%7.3f ms %4d  %-22s %s" % r)
