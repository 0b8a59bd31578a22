"""Probe: which torch (aten) device ops run in one full-model (IaBNet stage 1, six-term loss) fwd+bwd step of bench.py --mode model, and
FROM WHERE (innermost fabind_amd / bench frame, or the autograd node); the batch arrival (clone + plan_stage1) is profiled separately."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine, synthetic
from fabind_amd.models import get_model
from fabind_amd.models.model import compute_loss
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
engine.set_precision("bf16")
torch.autograd.set_multithreading_enabled(False)
torch.manual_seed(0)
class _Log:
    def log_message(self, *a, **k):
        pass


model = get_model(bench.stack_args(512, 4, 1), _Log(), dev).to(dev)
synthetic.condition_for_large_graphs(model)
model.eval()
hb = synthetic.make_hetero_batch([(1500, 40)] * 64, seed=0).to(dev)
params = list(model.parameters())


def arrive():
    data = hb.clone()
    return data, model.plan_stage1(data)


def step(data, plan):
    for p in params:
        p.grad = None
    out = model(data, stage=1, train=False, plan=plan)
    loss, _ = compute_loss(out, data)
    loss.backward()


for _ in range(3):
    step(*arrive())
torch.cuda.synchronize()


def report(title, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        fn(); torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if not e.name.startswith("aten::"):
            continue
        sdt = getattr(e, "self_device_time_total", 0) or 0
        if sdt <= 0:
            continue
        site = "?"
        for fr in (e.stack or []):
            if "fabind_amd" in fr or "bench.py" in fr or "aten_sites_model" in fr:
                site = fr.split("fabind_amd/")[-1][:80]
                break
        if site == "?":
            p_ = e.cpu_parent
            while p_ is not None:
                if "Backward" in p_.name or "autograd::engine" in p_.name:
                    site = "<autograd> " + p_.name[:60]
                    break
                p_ = p_.cpu_parent
        k = (e.name, str(e.input_shapes)[:50], site)
        rows[k][0] += 1
        rows[k][1] += sdt
    print("== %s: %d launching aten ops, %.2f ms device" % (title, sum(v[0] for v in rows.values()), sum(v[1] for v in rows.values()) / 1e3))
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])[:60]:
        print("%4d x %7.1f us  %-26s %-50s %s" % (v[0], v[1], k[0], k[1], k[2]))


box = []
report("batch arrival (clone + plan_stage1)", lambda: box.append(arrive()))
report("model step (forward + loss + backward)", lambda: step(*box[0]))
