"""Shared loaders for the golden fixtures (tests/golden/*.npz, made by oracle/make_golden.py)."""
import os

import numpy as np
import torch

from fabind_amd.synthetic import HeteroBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def weights(g, dtype=torch.float32):
    return {k[2:]: torch.from_numpy(v).to(dtype) if v.dtype.kind == "f" else torch.from_numpy(v)
            for k, v in g.items() if k.startswith("w_")}


def stack_inputs(g, dtype=torch.float32):
    d = {}
    for k, v in g.items():
        if k.startswith("in_"):
            t = torch.from_numpy(v)
            d[k[3:]] = t.to(dtype) if t.is_floating_point() and k[3:] != "segment_id" else t
    return d


def hetero_from_npz(g):
    data = HeteroBatch()
    for k, v in g.items():
        if not k.startswith("d_"):
            continue
        key, field = k[2:].split("::")
        t = torch.from_numpy(v)
        if key == "":
            setattr(data, field, t)
        else:
            kk = tuple(key.split("|")) if "|" in key else key
            data[kk][field] = t
    return data


def rmsd(a, b):
    return float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(-1).mean()))


def fused_edge_keep_mask(seed, E, H, p_drop):
    """The counter-based dropout mask of csrc/fused_edge.hip (include/fabind_hip.h: fabind_gcl_edge_fused), as a
    float [E, H] tensor of keep/(1-p) factors."""
    import torch
    thr = int(p_drop * 65536.0 + 0.5)
    e = torch.arange(E, dtype=torch.int64)[:, None]
    c = torch.arange(H, dtype=torch.int64)[None, :]
    M = 0xFFFFFFFF
    x = (seed + (e >> 1) * H + c) & M
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & M
    x = x ^ (x >> 16)
    half = torch.where((e & 1) == 1, x >> 16, x & 0xFFFF)
    return (half >= thr).float() / (1.0 - thr / 65536.0)
