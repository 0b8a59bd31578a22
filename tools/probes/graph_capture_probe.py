"""Probe: do launches made through the C ABI on torch's current stream get captured by torch.cuda.CUDAGraph (hipGraph) and replay?
A chain of K.gemm launches on static buffers, captured once, replayed 100 times; wall per replayed launch vs eager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fabind_amd import engine, kernels as K
engine.set_precision("bf16")
dev = torch.device("cuda:0")
M, H, n = 9088, 512, 40
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
Ws = [torch.randn(H, H, device=dev).to(torch.bfloat16) * 0.04 for _ in range(n)]
bufs = [torch.empty(M, H, dtype=torch.bfloat16, device=dev) for _ in range(2)]
def chain():
    cur = x
    for i, W in enumerate(Ws):
        K.gemm(cur, W, out=bufs[i % 2])
        cur = bufs[i % 2]
    return cur
for _ in range(3): chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): chain()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 20 / n
ref = chain().float().clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = chain()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
err = (out.float() - ref).abs().max().item()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / 20 / n
print("eager %.1f us per launch, graph replay %.1f us per launch, max |diff| %.3g" % (eager * 1e6, rep * 1e6, err))
