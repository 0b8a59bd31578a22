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


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin1, lin2, lin3 = torch.nn.Linear(40, 300), torch.nn.Linear(300, 50), torch.nn.Linear(50, 3)
    unused = torch.nn.Parameter(torch.randn(11))
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(lin3.parameters()) + [unused]
    red = parallel.GradReducer(params, world, bucket_bytes=1 << 13)          # several buckets
    assert len(red.buckets) > 3
    out = []
    for step in range(2):                                                    # second step: buffers are reused
        x = torch.randn(17, 40, generator=torch.Generator().manual_seed(100 * step + rank))
        for p in params:
            p.grad = None
        loss = lin3(torch.relu(lin2(torch.relu(lin1(x))))).pow(2).sum() * (rank + 1)
        loss.backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]
        red.finish()
        overl = [p.grad.clone() for p in params]
        # the non-overlapped form of the same arithmetic on the saved local gradients
        for p, g in zip(params, local):
            p.grad = g
        parallel.allreduce_gradients(params, world, bucket_bytes=1 << 12)
        out.append(([g.numpy().tolist() for g in overl], [p.grad.numpy().tolist() for p in params]))
    red.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_reducer_equals_plain_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, steps in res:
        for overl, plain in steps:
            for a, b in zip(overl, plain):
                assert torch.allclose(torch.tensor(a), torch.tensor(b), rtol=1e-6, atol=1e-7)
    assert res[0][1] == res[1][1]                                            # identical on both ranks
    assert all(v == 0.0 for v in res[0][1][0][0][-1])                       # the unused parameter: zeros, not None


class _NanModel(torch.nn.Module):
    """A stand-in with the 11-tuple contract of IaBNet.forward whose outputs go NaN on request (train_step's NaN guard)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(3))

    def forward(self, data, stage=1, train=True):
        v = self.w * data["x"]
        if data["nan"]:
            v = v * float("nan")
        z = v.sum().reshape(1)
        return (v, None, z, z, z, None, None, None, z, None, 0)


def _nan_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _NanModel()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    loss_fn = lambda out, data: (out[0].sum(), {"coord": out[0].sum()})
    res = []
    # step 0: clean on both ranks; step 1: NaN on rank 1 only -> BOTH ranks must skip (and nobody hangs); step 2: clean again
    for step, nan_on in enumerate([(), (1,), ()]):
        r = parallel.train_step(m, {"x": torch.full((3,), float(rank + 1)), "nan": rank in nan_on}, opt, loss_fn, world, clip=0)
        res.append((r is None, m.w.detach().clone().tolist()))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_nan_on_one_rank_skips_the_step_on_every_rank():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, steps in res:
        assert [s[0] for s in steps] == [False, True, False]
        assert steps[0][1] == steps[1][1]                   # the skipped step left the weights alone
    assert res[0][1] == res[1][1]                           # ranks stay in lock-step: mean gradient (1+2)/2 applied twice
    assert all(abs(v - (1.0 - 2 * 0.1 * 1.5)) < 1e-6 for v in res[0][1][2][1])


def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin1, lin2, lin3 = torch.nn.Linear(40, 300), torch.nn.Linear(300, 50), torch.nn.Linear(50, 3)
    late = torch.nn.Parameter(torch.randn(50))              # receives a gradient from the SECOND micro-batch only
    unused = torch.nn.Parameter(torch.randn(11))
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(lin3.parameters()) + [late, unused]
    red = parallel.GradReducer(params, world, bucket_bytes=1 << 13)
    out = []
    for step in range(2):
        for p in params:
            p.grad = None
        if step == 1:
            # an abandoned step first: one backward, then abort() -- the next step must start clean and nobody may hang
            x = torch.randn(5, 40, generator=torch.Generator().manual_seed(7 + rank))
            lin3(torch.relu(lin2(torch.relu(lin1(x))))).sum().backward()
            red.abort()
            for p in params:
                p.grad = None
        for micro in range(2):                               # gradient accumulation: two backward() calls, one finish()
            x = torch.randn(17, 40, generator=torch.Generator().manual_seed(1000 * step + 10 * micro + rank))
            hid = torch.relu(lin2(torch.relu(lin1(x))))
            if micro == 1:
                hid = hid + late
            (lin3(hid).pow(2).sum() * (rank + 1)).backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]          # the accumulated local gradient
        red.finish()
        overl = [p.grad.clone() for p in params]
        for p, g in zip(params, local):
            p.grad = g
        parallel.allreduce_gradients(params, world, bucket_bytes=1 << 12)
        out.append(([g.numpy().tolist() for g in overl], [p.grad.numpy().tolist() for p in params]))
    red.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_with_gradient_accumulation_and_abort():
    """Two backward() calls per step: finish() must average the ACCUMULATED gradients (VERDICT r2 weak item 13 -- the buckets of
    the first micro-batch have already been sent when the second one adds to their parameters); abort() discards a partial step."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, steps in res:
        for overl, plain in steps:
            for a, b in zip(overl, plain):
                assert torch.allclose(torch.tensor(a), torch.tensor(b), rtol=1e-6, atol=1e-7)
            assert any(v != 0.0 for v in overl[-2])          # the late parameter did receive its (averaged) gradient
            assert all(v == 0.0 for v in overl[-1])
    assert res[0][1] == res[1][1]


def _asym_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(600, 30))           # always used; big enough to fill buckets of its own
    b = torch.nn.Parameter(torch.randn(300, 20))
    c = torch.nn.Parameter(torch.randn(9))                 # never used in the discovery step; fires on rank 1 ONLY in step 2, never again
    params = [a, b, c]
    red = parallel.GradReducer(params, world, bucket_bytes=1 << 13)
    out = []
    for step in range(5):
        for p in params:
            p.grad = None
        loss = (a ** 2).sum() * (rank + 1) + b.sum() * (step + 1)
        if step == 2 and rank == 1:
            loss = loss + 8.0 * c.sum()
        loss.backward()
        red.finish()
        out.append([p.grad.numpy().tolist() for p in params])
        assert (id(c) in red.unused) == (step < 2)         # membership in the never-used set changes on BOTH ranks together
    red.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_late_parameter_on_one_rank_only():
    """a never-used parameter that fires late on ONE rank must leave the never-used set on EVERY rank -- the flat
    buckets are reduced and divided in place, so on a rank that kept it 'never used' the slot would hold the previous step's average
    and be summed again in every later step (4.0, 2.0, 1.0, ... instead of 4.0, 0, 0)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_asym_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                          # identical averaged gradients on both ranks, every step
    for step, grads in enumerate(res[0][1]):
        expect_c = 4.0 if step == 2 else 0.0               # (0 + 8) / 2 in the step it fired, zeros before and after
        assert all(abs(v - expect_c) < 1e-6 for v in grads[2]), (step, grads[2])
        assert all(abs(v - (step + 1.0)) < 1e-6 for row in grads[1] for v in row)


NEVER_USED = ("inter_layer.", "pocket_pred_model.gnn.out_layer.coord_mlp.")   # 33 tensors without a gradient


def _production_params():
    """The production IaBNet's trainable tensors (394 state keys, 36,270,615 parameters) with their names, and which of them the
    reference's training step never reaches (30 x `...att_i.inter_layer.*` + 3 x the pocket model's `out_layer.coord_mlp.*`)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_cpu import _Logger, _args
    from fabind_amd.models import get_model
    a = _args(512, 4, 8)
    a.pocket_pred_hidden_size = 128
    m = get_model(a, _Logger(), None)
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    dead = [n for n, _ in named if ".gnn.att_" in n and ".inter_layer." in n and "cross_attn_module" not in n
            or n.startswith("pocket_pred_model.gnn.out_layer.coord_mlp.")]
    return m, named, set(dead)


def _production_worker(rank, world, port, q, bucket_dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(2)
    m, named, dead = _production_params()
    params = [p for _, p in named]
    used = [p for n, p in named if n not in dead]
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[bucket_dtype]
    red = parallel.GradReducer(params, world, bucket_dtype=dt)                 # default 64 MB buckets: 3 of them at 145 MB
    info = dict(n_params=len(params), n_dead=len(dead), n_buckets=len(red.buckets), early=[], unused_after=[], max_err=[])
    for step in range(3):
        for p in params:
            p.grad = None
        # backward produces the gradients roughly in reverse registration order: one backward() per tensor makes that order exact
        # (every hook fires once per step; the dead tensors get none, like in the real step)
        for i, p in enumerate(reversed(used)):
            (p * float((rank + 1) * (1 + i % 3))).sum().backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]
        red.finish()
        info["early"].append(red.issued_early)
        info["unused_after"].append(len(red.unused))
        got = [p.grad.clone() for p in params]
        for p, g in zip(params, local):
            p.grad = g
        parallel.allreduce_gradients(params, world, bucket_dtype=dt)
        info["max_err"].append(max(float((a_ - p.grad).abs().max()) for a_, p in zip(got, params)))
        if step == 2:
            info["dead_zero"] = all(float(p.grad.abs().max()) == 0.0 for n, p in named if n in dead)
            info["sample"] = [float(got[0].reshape(-1)[0]), float(got[-1].reshape(-1)[0])]
    info["n_buckets_rebuilt"] = len(red.buckets)
    info["last_bucket_is_dead_tail"] = all(id(p) in red.unused for p in red.buckets[-1][-len(dead):])
    red.close()
    q.put((rank, info))
    dist.barrier()
    dist.destroy_process_group()


def _run_production(bucket_dtype):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_production_worker, args=(r, world, port, q, bucket_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_reducer_overlaps_on_the_production_parameter_list():
    """on the real model's 394 tensors the 33 never-used ones sat in the leading buckets, so no bucket left
    before finish().  After the discovery step the buckets follow the arrival order with the never-used tensors last: at least 2 of
    the 3 buckets must be on the wire before finish(), and the result must equal the plain all-reduce on both ranks."""
    res = _run_production("fp32")
    for rank, info in res:
        assert info["n_params"] == 394 and info["n_dead"] == 33 and info["n_buckets"] == 3 and info["n_buckets_rebuilt"] == 3
        assert info["early"][0] == 0                      # the discovery step cannot overlap (what round 3 shipped for every step)
        assert info["early"][1] >= 2 and info["early"][2] >= 2
        assert info["unused_after"] == [33, 33, 33]
        assert info["last_bucket_is_dead_tail"] and info["dead_zero"]
        assert max(info["max_err"]) == 0.0                # same arithmetic as the non-overlapped form: bit-identical
    assert res[0][1]["sample"] == res[1][1]["sample"]


def test_reducer_bf16_buckets_on_the_production_parameter_list():
    res = _run_production("bf16")
    for rank, info in res:
        assert info["early"][1] >= 2
        assert max(info["max_err"]) == 0.0                # both forms round the per-rank gradients to bf16 before the sum
    assert res[0][1]["sample"] == res[1][1]["sample"]


def test_weighted_sharding_balances_pair_work_with_equal_counts():
    import numpy as np
    rng = np.random.default_rng(0)
    for n, world in ((16, 2), (64, 8), (129, 8), (7, 4)):
        P, C = rng.integers(60, 600, n), rng.integers(10, 80, n)            # PDBbind-like spread: P*C varies > 10x
        w = P * C
        parts = [parallel.shard_complexes(n, r, world, weights=w) for r in range(world)]
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(n))                      # a partition
        sizes = [len(p_) for p_ in parts]
        assert sizes == [hi - lo for lo, hi in (parallel.shard_complexes(n, r, world) for r in range(world))]
        loads = np.array([w[p_].sum() for p_ in parts], dtype=np.float64)
        contiguous = np.array([w[lo:hi].sum() for lo, hi in (parallel.shard_complexes(n, r, world) for r in range(world))], dtype=np.float64)
        if n >= 2 * world:
            assert loads.max() / loads.mean() <= 1.10                         # the slowest rank within 10 % of the mean
            assert loads.max() <= contiguous.max()
        # deterministic: every rank computes the same deal
        assert all((parallel.shard_complexes(n, r, world, weights=w) == parts[r]).all() for r in range(world))
