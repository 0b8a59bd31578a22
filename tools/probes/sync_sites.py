"""PROBE: every host-synchronising call of one step (torch.cuda.set_sync_debug_mode("warn")), grouped by the fabind_amd line that made it.
usage: sync_sites.py [stack|model|plus_train|fwd] [n_prot]"""
import collections, os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "stack"
sys.argv = ["bench.py", "--mode", {"stack": "fwdbwd", "fwd": "fwd"}.get(mode, mode), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
            "--no-extras"] + (["--n-prot", sys.argv[2]] if len(sys.argv) > 2 else []) + (["--precision", sys.argv[3]] if len(sys.argv) > 3 else [])
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message).lower():
        st = traceback.extract_stack()
        mine = [f for f in st if "/fabind_amd/" in f.filename or f.filename.endswith("bench.py")]
        key = " <- ".join("%s:%d" % (os.path.relpath(f.filename, ROOT), f.lineno) for f in mine[-3:][::-1]) or "%s:%d" % (filename, lineno)
        sites[key] += 1
        return
    orig(message, category, filename, lineno, file, line)


warnings.showwarning = show
warnings.simplefilter("always")
# the bench's own timing syncs are not of interest: enable the mode inside the step only
real_timed = None
import fabind_amd.kernels as K
state = {"n": 0}
_sync = torch.cuda.synchronize


def patched_sync(*a, **k):
    torch.cuda.set_sync_debug_mode(0)
    r = _sync(*a, **k)
    state["n"] += 1
    if state["n"] >= 2:                       # after the warm-up's synchronize: the timed step runs under the debug mode
        torch.cuda.set_sync_debug_mode(1)
    return r


torch.cuda.synchronize = patched_sync
try:
    bench.main()
finally:
    torch.cuda.set_sync_debug_mode(0)
print("---- synchronising calls of the timed step, by call site (count)")
for k, v in sites.most_common(60):
    print("%4d  %s" % (v, k))
