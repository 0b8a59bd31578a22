"""Does any kernel of the stack forward read memory it did not write?  Runs the B = 64 bench batch forward once on a clean allocator,
then again after the caching allocator's free blocks have been filled with a poison value (NaN / 1e30 / -7): a kernel that reads a
torch.empty buffer before writing it shows up as a changed (or non-finite) result.  The engine's per-layer captures name the first
layer that differs.  argv[1] = precision (bf16x3 | bf16 | fp32), argv[2] = 'grad' to include a backward pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fabind_amd import engine, synthetic
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
grad = len(sys.argv) > 2 and sys.argv[2] == "grad"
engine.set_precision(prec)
m = bench.build_model(512, 4, 1).eval().to(dev)
inp = synthetic.make_stack_batch([(1500, 40)] * 64, 512, seed=0)
t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}


def fwd():
    Hin = t["H"].clone().requires_grad_(grad)
    with torch.set_grad_enabled(grad):
        X, Hh = m(t["X"].clone(), Hin, t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"],
                  t["LAS_edge_index"], t["coord_LAS"])
    return X, Hh, Hin


def run():
    engine.DEBUG_CAPTURE = {}
    for p in m.parameters():
        p.grad = None
    X, Hh, Hin = fwd()
    out = {"X": X.detach().clone(), "H": Hh.detach().clone()}
    if grad:
        ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
        out["dHin"] = Hin.grad.clone()
        for n_, p in m.named_parameters():
            if p.grad is not None:
                out["grad:" + n_] = p.grad.clone()
    cap = {k: v.clone() for k, v in engine.DEBUG_CAPTURE.items()}
    engine.DEBUG_CAPTURE = None
    torch.cuda.synchronize()
    return out, cap


def poison(val):
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    blocks = []
    for gb in (16, 8, 4, 2, 1, 1, 0.5, 0.25, 0.25, 0.1, 0.1, 0.05, 0.05, 0.02, 0.02, 0.01, 0.01):      # big and small blocks of the cache
        for _ in range(3):
            blocks.append(torch.full((int(gb * (1 << 28)),), val, dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    del blocks


ref, cap0 = run()
print(prec, "grad" if grad else "fwd", "reference finite:", all(bool(torch.isfinite(v).all()) for v in ref.values()))
for val in (float("nan"), 1e30, -7.0, 0.0):
    poison(val)
    out, cap = run()
    bad = [(k, float((out[k].float() - ref[k].float()).abs().max())) for k in ref if not torch.equal(out[k], ref[k])]
    first = [k for k in cap0 if not torch.equal(cap[k], cap0[k])]
    print("poison %-6s: %d of %d results differ %s; first differing capture: %s" % (val, len(bad), len(ref), bad[:4], first[:3]))
