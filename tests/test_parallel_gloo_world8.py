"""CPU, gloo at the REAL world sizes (4 and 8 ranks): what only shows with more than two ranks -- rank 0's arrival
order broadcast and the bucket rebuild when every rank saw a DIFFERENT order in the discovery step, the never-used set staying collective
when a late parameter fires on one rank of eight, `shard_complexes(weights = P x C)` at BASELINE config 4's shape (16 complexes per GPU x 8),
and the DDP mean (main_fabind.py:194-195, 289-296, 419-423)."""
import hashlib
import os
import socket

import numpy as np
import pytest
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


def _run(worker, world, *args, timeout=300):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _order_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (700, 40, 900, 13, 650, 300, 1100, 5)]       # 8 tensors, several per 8 KB bucket
    late = torch.nn.Parameter(torch.randn(9))             # no gradient in the discovery step; fires on the LAST rank only, in step 2
    dead = torch.nn.Parameter(torch.randn(31))            # never
    params = ps + [late, dead]
    red = parallel.GradReducer(params, world, bucket_bytes=1 << 13)
    out, layouts = [], []
    for step in range(5):
        for p in params:
            p.grad = None
        # discovery step: EVERY RANK produces its gradients in a different order (one backward per tensor makes the order exact);
        # later steps: one common order that differs from rank 0's discovery order
        idx = list(range(len(ps)))
        order = idx[rank % len(ps):] + idx[:rank % len(ps)] if step == 0 else list(reversed(idx))
        for i in order:
            (ps[i] * float(rank + 1 + step)).sum().backward()
        if step == 2 and rank == world - 1:
            (late * 8.0).sum().backward()
        red.finish()
        out.append([p.grad.clone() for p in params])
        layouts.append(hashlib.sha1(repr([[red.index[id(p)] for p in b] for b in red.buckets]).encode()).hexdigest() + ":%d" % len(red.unused))
        assert (id(late) in red.unused) == (step < 2) and id(dead) in red.unused
    red.close()
    q.put((rank, [[g.tolist() for g in s] for s in out], layouts, [red.index[id(p)] for b in red.buckets for p in b]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_reducer_order_broadcast_rebuild_and_late_parameter_at_world(world):
    res = _run(_order_worker, world)
    ref = res[0]
    for rank, steps, layouts, flat_order in res:
        assert steps == ref[1], rank                          # every rank holds the same averaged gradients after every step
        assert layouts == ref[2], rank                        # ... and the same buckets (rank 0's discovery order), before and after the late parameter
    # rank 0's discovery order (0, 1, ..., 7) became the bucket order everywhere; never-used tensors trail in reverse registration order
    assert ref[3] == [0, 1, 2, 3, 4, 5, 6, 7, 9, 8]
    mean_scale = lambda step: sum(r + 1 + step for r in range(world)) / world
    for step, grads in enumerate(ref[1]):
        for g in grads[:8]:
            assert all(abs(v - mean_scale(step)) < 1e-5 for v in g)
        exp_late = 8.0 / world if step == 2 else 0.0          # fired on one rank of `world`: the mean; zeros before and after
        assert all(abs(v - exp_late) < 1e-6 for v in grads[8]), (step, grads[8][:3])
        assert all(v == 0.0 for v in grads[9])


def _mean_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(257, 33))
    b = torch.nn.Parameter(torch.randn(33))
    x = torch.randn(19, 257, generator=torch.Generator().manual_seed(100 + rank))
    ((x @ w + b) ** 2).mean().backward()
    local = [w.grad.clone(), b.grad.clone()]
    parallel.allreduce_gradients([w, b], world, bucket_bytes=1 << 12)
    total = parallel.clip_grad_norm_([w, b], 1.0)
    q.put((rank, [t.tolist() for t in local], [w.grad.tolist(), b.grad.tolist()], float(total)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_allreduce_is_the_ddp_mean_and_clip_follows_it_at_world(world):
    res = _run(_mean_worker, world)
    mw = sum(torch.tensor(r[1][0]) for r in res) / world
    mb = sum(torch.tensor(r[1][1]) for r in res) / world
    norm = float(torch.sqrt((mw ** 2).sum() + (mb ** 2).sum()))
    coef = min(1.0, 1.0 / (norm + 1e-6))
    for rank, _, got, total in res:
        assert abs(total - norm) <= 1e-5 * norm                                   # the norm of the AVERAGED gradient: clip after the all-reduce
        assert torch.allclose(torch.tensor(got[0]), mw * coef, rtol=1e-5, atol=1e-7) and torch.allclose(torch.tensor(got[1]), mb * coef, rtol=1e-5, atol=1e-7)
        assert got == res[0][2]


def test_weighted_sharding_at_config4_shape_16_per_gpu_times_8():
    """BASELINE config 4: batch 16 per GPU x 8 GPUs = 128 complexes per step, PDBbind-like sizes: every rank gets exactly 16, the deal is a
    partition, every rank computes the same deal, and the slowest rank's pair work (sum of P x C: what the pair path costs) is within 5 %
    of the mean -- against up to ~1.5x for the contiguous split."""
    rng = np.random.default_rng(7)
    for trial in range(5):
        P, C = rng.integers(50, 800, 128), rng.integers(8, 90, 128)
        w = P * C
        parts = [parallel.shard_complexes(128, r, 8, weights=w) for r in range(8)]
        assert [len(p) for p in parts] == [16] * 8
        assert sorted(np.concatenate(parts).tolist()) == list(range(128))
        loads = np.array([w[p].sum() for p in parts], dtype=np.float64)
        contiguous = np.array([w[16 * r:16 * r + 16].sum() for r in range(8)], dtype=np.float64)
        assert loads.max() / loads.mean() <= 1.05, loads.max() / loads.mean()
        assert loads.max() <= contiguous.max()
        assert all((parallel.shard_complexes(128, r, 8, weights=w) == parts[r]).all() for r in range(8))
