"""Probe: throughput of the CPU oracle (bench.cpu_baseline, forward and forward+backward) vs torch thread count on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
print("host cpus:", os.cpu_count(), "torch default threads:", torch.get_num_threads())
for n in (8, 16, 32, 64, torch.get_num_threads()):
    torch.set_num_threads(n)
    f = bench.cpu_baseline(512, 4, 1, 1500, 40, budget_s=6.0, backward=False)
    print("threads %3d  forward %.4f complexes/s" % (n, f["value"]), flush=True)
for n in (16, 32):
    torch.set_num_threads(n)
    b = bench.cpu_baseline(512, 4, 1, 1500, 40, budget_s=12.0, backward=True)
    print("threads %3d  forward+backward %.4f complexes/s" % (n, b["value"]), flush=True)
