"""Probe: find the first intermediate that differs between two identical forward calls at the bench shape (bf16), using the
engine's per-layer capture; replicates tests/test_gpu_headline.py::test_bench_batch_properties (fresh uploads per call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import test_gpu_headline as T
from fabind_amd import engine, kernels as K, synthetic
dev = torch.device("cuda:0")
snap = (sys.argv[1] != "nosnap") if len(sys.argv) > 1 else True
inp = synthetic.make_stack_batch([(1500, 40)] * 64, 512, seed=0, snap=snap)


def once(m):
    cap = {}
    engine.DEBUG_CAPTURE = cap
    try:
        X, H, _ = T._hip(m, inp, dev)
    finally:
        engine.DEBUG_CAPTURE = None
    cap["X"], cap["H"] = X.clone(), H.clone()
    return cap


for pre_fp32 in (False, True):
    for mfma in (True, False):
        if pre_fp32:
            engine.set_precision("fp32")
            T._hip(T._model(1).to(dev), inp, dev)
        engine.set_precision("bf16")
        K.CROSS_ATTN_MFMA = mfma
        m = T._model(1).to(dev)
        a = once(m)
        for rep in range(3):
            b = once(m)
            diffs = [(k, float((a[k].float() - b[k].float()).abs().max())) for k in a if float((a[k].float() - b[k].float()).abs().max()) > 0]
            print("snap=%s fp32-run-first=%s MFMA=%s rerun %d: %s" % (snap, pre_fp32, mfma, rep,
                  "bit-identical" if not diffs else "first differing: %s; all: %s" % (diffs[0], [d[0] for d in diffs])), flush=True)
K.CROSS_ATTN_MFMA = None
engine.set_precision("fp32")
