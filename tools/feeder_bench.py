"""Throughput of the device batch builder / double-buffered feeder at the bench shape (PCIe-inclusive: raw arrays start in
pageable host memory).  usage: python tools/feeder_bench.py [batch] [n_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fabind_amd.data import DeviceFeeder, build_batch, pack_samples

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
L, nc = 1500, 40
idx = np.arange(nc)
bonds = np.stack([np.concatenate([idx[:-1], idx[1:]]), np.concatenate([idx[1:], idx[:-1]])], 1)
ii, jj = np.nonzero((np.abs(idx[:, None] - idx[None]) <= 2) & (idx[:, None] != idx[None]))
def sample():
    prot = rng.normal(size=(L, 3)).astype(np.float32) * 18.0
    walk = np.cumsum(1.5 * rng.normal(size=(nc, 3)) / np.sqrt(3.0), 0).astype(np.float32)
    return dict(protein_node_xyz=prot, protein_esm2_feat=rng.normal(size=(L, 1280)).astype(np.float32), coords=walk - walk.mean(0) + 6.0,
                compound_node_features=rng.normal(size=(nc, 56)).astype(np.float32), input_atom_edge_list=bonds,
                LAS_edge_index=np.stack([ii, jj]), rdkit_coords=walk - walk.mean(0))
groups = [[sample() for _ in range(B)] for _ in range(2)] * (NB // 2)
build_batch(groups[0], dev); torch.cuda.synchronize()
t0 = time.time(); p = pack_samples(groups[0]); t1 = time.time()
b = build_batch(None, dev, packed=p); torch.cuda.synchronize(); t2 = time.time()
print("one batch of %d (1500 residues x 1280 feats, 40 atoms): host pack %.1f ms (%.0f MB), copy + device assembly %.1f ms" % (
    B, (t1 - t0) * 1e3, p[0].numel() * 4 / 1e6, (t2 - t1) * 1e3))
t0 = time.time(); n = 0
for batch in DeviceFeeder(groups, dev):
    n += B
torch.cuda.synchronize()
dt = time.time() - t0
print("feeder: %d complexes in %.2f s = %.0f complexes/s (pack thread + side-stream copy/assembly overlapped)" % (n, dt, n / dt))
