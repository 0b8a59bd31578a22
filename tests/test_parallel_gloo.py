"""CPU, world_size 2 over gloo: the data-parallel gradient all-reduce keeps DDP semantics
(mean of per-rank gradients, unused parameters as zeros, clip after the all-reduce)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fabind_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    used = torch.nn.Parameter(torch.randn(5, 3))
    big = torch.nn.Parameter(torch.randn(300, 40))
    unused = torch.nn.Parameter(torch.randn(7))            # never gets a gradient (cf. att_i.inter_layer.*)
    half = torch.nn.Parameter(torch.randn(4))              # gets a gradient on rank 0 only
    x = torch.arange(15, dtype=torch.float32).reshape(5, 3) * (rank + 1)
    loss = (used * x).sum() + (big ** 2).sum() * (rank + 1) + (half.sum() if rank == 0 else 0.0)
    loss.backward()
    params = [used, big, unused, half]
    parallel.allreduce_gradients(params, world, bucket_bytes=1 << 12)   # force several buckets
    total = parallel.clip_grad_norm_(params, 1.0)
    q.put((rank, [p.grad.numpy().tolist() for p in params], float(total)))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_is_mean_of_rank_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    used, big = torch.randn(5, 3), torch.randn(300, 40)
    x = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    exp = [(x * 1 + x * 2) / 2, (2 * big * 1 + 2 * big * 2) / 2, torch.zeros(7), torch.ones(4) / 2]
    norm = torch.sqrt(sum((e ** 2).sum() for e in exp))
    coef = min(1.0, 1.0 / (float(norm) + 1e-6))
    for rank, grads, total in res:
        assert abs(total - float(norm)) <= 1e-4 * float(norm)
        for g, e in zip(grads, exp):
            assert torch.allclose(torch.tensor(g), e * coef, rtol=1e-5, atol=1e-6)
    # both ranks hold identical gradients after the collective
    for a, b in zip(res[0][1], res[1][1]):
        assert a == b


def test_shard_complexes_partitions_the_batch():
    for n in (1, 7, 16, 64, 129):
        for world in (1, 2, 4, 8):
            parts = [parallel.shard_complexes(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
