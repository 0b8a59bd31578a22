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
