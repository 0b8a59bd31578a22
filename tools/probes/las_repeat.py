"""Is fabind_las_step (one wave per ligand atom, lanes stride over the complex's LAS edges, fixed-order wave sum) bit-reproducible when
several PROCESSES share the device?  Fixed synthetic inputs, N launches, every result compared with the first.
usage: las_repeat.py [launches] [tag]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fabind_amd import kernels as K  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
g = np.random.RandomState(0)
B, n, C = 16, 190, 30
node_off = np.arange(B + 1) * n
li, lj = [], []
for b in range(B):
    for a in range(1, C):
        for c in range(1, C):
            if a != c:
                li.append(b * n + a); lj.append(b * n + c)
las_i, las_j = np.array(li, np.int32), np.array(lj, np.int32)
las_off = np.arange(B + 1) * ((C - 1) * (C - 2))
x0 = g.randn(B * n, 3).astype(np.float32)
x = x0 + 0.01 * g.randn(B * n, 3).astype(np.float32)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
args = (t(x, torch.float32), t(x0, torch.float32), t(las_i, torch.int32), t(las_j, torch.int32), t(las_off, torch.int32), t(node_off, torch.int32),
        t(np.full(B, C), torch.int32), B, n, 0.05, 3.0)
ref = K.las_step(*args).clone()
bad = 0
for it in range(n_it):
    out = K.las_step(*args)
    if it % 64 == 63 or it == n_it - 1:
        torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        d = (out - ref).abs()
        if bad <= 5:
            print("%s launch %d differs: %d elements, max %.3e" % (tag, it, int((d > 0).sum()), float(d.max())), flush=True)
print("%s: %d of %d launches differ" % (tag, bad, n_it), flush=True)
