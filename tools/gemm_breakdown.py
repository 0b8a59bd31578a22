"""Per-shape GEMM time breakdown of one fwd+bwd step (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, kernels as K
dev = torch.device("cuda:0")
engine.set_precision("bf16")
mode = sys.argv[1] if len(sys.argv) > 1 else "fwdbwd"
model = bench.build_model(512, 4, 1).to(dev).eval()
inp = bench.make_batch(64, 1500, 40, 512, 0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
def step():
    if mode == "fwd":
        with torch.no_grad():
            model(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
        return
    for p in model.parameters(): p.grad = None
    X, H = model(t["X"].clone(), t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
    ((X * X).mean() + (H * H).mean() * 1e-6).backward()
step(); torch.cuda.synchronize()
K.PROFILE = {}
step(); torch.cuda.synchronize()
rows = []
for label, evs in K.PROFILE.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in evs)
    rows.append((ms, len(evs), evs[0][2], label))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total profiled MFMA-kernel time %.1f ms" % tot)
for ms, n, fl, label in rows[:16]:
    print("%7.2f ms  n=%3d  avg %7.1f us  %6.1f TF  %s" % (ms, n, ms / n * 1e3, fl * n / ms / 1e9, label))
