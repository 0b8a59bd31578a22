"""Throughput of the batched ligand post-optimisation kernel (1000 Adam steps per ligand) next to the CPU oracle.
usage: python tools/post_optim_bench.py [n_ligands] [atoms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fabind_amd.utils.post_optim_utils import post_optimize_compound_coords_batched
from oracle.post_optim_oracle import post_optimize_compound_coords as orc

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
refs, preds, las, batch = [], [], [], []
idx = np.arange(n)
ii, jj = np.nonzero((np.abs(idx[:, None] - idx[None]) <= 2) & (idx[:, None] != idx[None]))
for li in range(L):
    st = rng.normal(size=(n, 3)); st /= np.linalg.norm(st, axis=1, keepdims=True)
    r = np.cumsum(1.5 * st, 0).astype(np.float32)
    refs.append(r); preds.append((r + rng.normal(scale=0.5, size=r.shape)).astype(np.float32))
    las.append(np.stack([ii, jj]) + li * n); batch += [li] * n
ref, pred = torch.from_numpy(np.concatenate(refs)).to(dev), torch.from_numpy(np.concatenate(preds)).to(dev)
e, b = torch.from_numpy(np.concatenate(las, 1)).to(dev), torch.tensor(batch, device=dev)
post_optimize_compound_coords_batched(ref, pred, b, 1000, e)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    x, loss, rmsd = post_optimize_compound_coords_batched(ref, pred, b, 1000, e)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print("GPU: %d ligands x %d atoms x 1000 steps: %.2f ms per batch = %.0f ligands/s" % (L, n, dt * 1e3, L / dt))
t0 = time.time()
k = 3
for li in range(k):
    orc(refs[li], preds[li], 1000, las[li] - li * n)
dc = (time.time() - t0) / k
print("CPU oracle (numpy, 1 core): %.2f s per ligand = %.2f ligands/s  -> GPU/CPU %.0fx" % (dc, 1 / dc, (L / dt) * dc))
