"""Probe: which torch (aten) device ops run in one pocket-sized fwd+bwd step, and FROM WHERE: op name, shapes, count, device time,
and the innermost fabind_amd frame of the Python stack that issued them (backward ops: the autograd node's name).
usage: aten_sites.py [n_prot]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine
from torch.profiler import profile, ProfilerActivity
n_prot = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
engine.set_precision("bf16")
torch.autograd.set_multithreading_enabled(False)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "device_time", 0) or 0
    # only ops that launch something themselves (self device time), not wrappers
    sdt = getattr(e, "self_device_time_total", 0) or 0
    if sdt <= 0:
        continue
    site = "?"
    for fr in (e.stack or []):
        if "fabind_amd" in fr or "bench.py" in fr or "pocket" in fr:
            site = fr.split("fabind_amd/")[-1][:70]
            break
    if site == "?":
        p_ = e.cpu_parent
        while p_ is not None:
            if "Backward" in p_.name or "autograd::engine" in p_.name:
                site = "<autograd> " + p_.name[:60]
                break
            p_ = p_.cpu_parent
    k = (e.name, str(e.input_shapes)[:60], site)
    rows[k][0] += 1
    rows[k][1] += sdt
tot = sum(v[0] for v in rows.values())
print("%d launching aten ops, %.2f ms device" % (tot, sum(v[1] for v in rows.values()) / 1e3))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])[:70]:
    print("%4d x %7.1f us  %-28s %-60s %s" % (v[0], v[1], k[0], k[1], k[2]))
