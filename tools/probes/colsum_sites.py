"""Which call sites still launch a separate column sum in one fwd+bwd step of the default bench workload (after the warm-up)."""
import collections, os, sys, traceback
sys.path.insert(0, os.getcwd())
import torch
import bench
from fabind_amd import engine, kernels as K, ops
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "1", "--warmup", "1"]
sites = collections.Counter()
orig = K.colsum
def traced(x, *a, **k):
    fr = [f for f in traceback.extract_stack()[:-1] if "fabind_amd" in f.filename][-2:]
    sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)) + "  shape %s %s" % (tuple(x.shape), x.dtype)] += 1
    return orig(x, *a, **k)
K.colsum = traced
bench.main()
for s, n in sites.most_common():
    print("%4d  %s" % (n, s))
