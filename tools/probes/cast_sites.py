"""Probe: where the bf16 cast kernels (aten::to / copy_ with a bf16 result) of one fwd+bwd step of the bench come from -- Python
source line of the innermost fabind_amd frame, count and device time."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine
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
sites = collections.Counter(); elems = collections.Counter()
orig_to = torch.Tensor.to
def spy_to(self, *a, **k):
    r = orig_to(self, *a, **k)
    if r is not self and r.dtype == torch.bfloat16 and self.dtype == torch.float32 and self.is_cuda:
        fr = [f for f in traceback.extract_stack() if "fabind_amd" in f.filename]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-6:][::-1] if "ops.py:116" not in ("%s:%d" % (os.path.basename(f.filename), f.lineno)))
        sites[key] += 1; elems[key] += self.numel()
    return r
torch.Tensor.to = spy_to
step(); torch.cuda.synchronize()
torch.Tensor.to = orig_to
print("fp32 -> bf16 casts through Tensor.to in one step: %d, %.1f M elements" % (sum(sites.values()), sum(elems.values()) / 1e6))
for k, c in sorted(sites.items(), key=lambda kv: -elems[kv[0]])[:25]:
    print("%4d x %8.2f M elements  %s" % (c, elems[k] / 1e6, k))
