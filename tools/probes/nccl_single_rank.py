"""Probe: the RCCL (backend "nccl") path of fabind_amd.parallel on a real GPU with ONE rank -- the world-size-2 tests run on
gloo and stage through the host, so the device-tensor collective branch never executes there.  A one-rank SUM is the
identity, so calling allreduce_gradients(params, world=2) must leave exactly grad / 2 (bucketing, async wait, copy-back,
missing gradients as zeros).  Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1
--master-port 29511 tools/probes/nccl_single_rank.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from fabind_amd import parallel

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="env://", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = torch.Generator().manual_seed(0)
shapes = [(512, 512), (1024,), (2048, 512), (3,), (512, 1025), (4096, 512)]          # ~13 MB fp32 -> several 4 MB buckets
params = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
for i, p in enumerate(params):
    if i != 3:
        p.grad = torch.randn(*p.shape, generator=g).to(dev)                          # params[3] has no gradient: counts as zeros
ref = [None if p.grad is None else p.grad.clone() for p in params]
parallel.allreduce_gradients(params, world=2, bucket_bytes=4 << 20)
torch.cuda.synchronize()
for p, r in zip(params, ref):
    want = torch.zeros_like(p) if r is None else r / 2
    assert p.grad is not None and torch.equal(p.grad, want), "all-reduce result differs"
total = parallel.clip_grad_norm_(params, 1.0)
assert torch.isfinite(total)
dist.barrier()
dist.destroy_process_group()
print("RCCL single-rank path ok: %d tensors in %d-byte buckets, grad norm before clip %.3f" % (len(params), 4 << 20, float(total)))
