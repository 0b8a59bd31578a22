"""Probe: where the HOST time of a production-shaped training step goes (full IaBNet, B complexes of ~300 residues / 30 atoms, the
pocket crop leaves ~100 residues for the 4-layer stack, n_iter = 8, train mode): wall per step, GPU kernel time per step (events
around the step are wall too; kernel time comes from torch.profiler), and the cProfile top of the host side.
usage: host_profile.py [B] [n_iter]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fabind_amd import engine, synthetic
from fabind_amd.models import get_model
from fabind_amd.models.model import compute_loss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
engine.set_precision("bf16")


class _Log:
    def log_message(self, m):
        pass


torch.manual_seed(0)
model = get_model(bench.stack_args(512, 4, n_iter), _Log(), dev).to(dev)
model.train()
hb = synthetic.make_hetero_batch([(300 + 7 * (i % 5), 25 + (i % 9)) for i in range(B)], seed=0).to(dev)
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    data = hb.clone()
    out = model(data, stage=1, train=True)
    loss, _ = compute_loss(out, data)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    step()
torch.cuda.synchronize()
wall = (time.time() - t0) / 5
print("B=%d n_iter=%d: %.1f ms per training step = %.0f complexes/s" % (B, n_iter, 1e3 * wall, B / wall))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages()
ktime = sum(e.device_time_total for e in ev) / 1e3
print("GPU kernel time of one step: %.1f ms in %d launches" % (ktime, sum(e.count for e in ev)))
for e in sorted(ev, key=lambda e: -e.device_time_total)[:32]:
    print("  %7.2f ms %6d x %7.1f us  %s" % (e.device_time_total / 1e3, e.count, e.device_time_total / max(e.count, 1), e.key[:110]))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
