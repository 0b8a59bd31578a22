"""GPU, two ranks on ONE device over gloo (RCCL needs a GPU per rank; the gradients are staged through the host): BASELINE config 4's
per-GPU shape -- a ragged batch of 16 pocket-sized complexes per rank -- trained for three data-parallel steps of the full IaBNet with
the six-term loss, sharded by `parallel.shard_complexes(weights=P*C)`, gradients through the overlapped `GradReducer`, clip after the
all-reduce (main_fabind.py:392-426).  Checked against a single-process emulation of the same arithmetic: per step, the gradient of
every rank's shard computed one after the other on the same weights, their MEAN (DDP semantics), clip, optimizer step
(config 4 has no dataset and no 8-GPU node here; this is its per-rank step on real kernels)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N_GLOBAL, WORLD, STEPS, LR = 32, 2, 3, 1e-3


def _sizes():
    g = np.random.RandomState(3)
    return [(int(g.randint(150, 420)), int(g.randint(10, 41))) for _ in range(N_GLOBAL)]


class _Logger:
    def log_message(self, s):
        pass


def _model(dev):
    from fabind_amd import synthetic
    from fabind_amd.models import get_model
    from test_gpu_stack import _args
    a = _args(512, 4, 1)
    a.pocket_pred_hidden_size = 128
    a.random_n_iter = False
    torch.manual_seed(0)
    m = get_model(a, _Logger(), None).eval()       # eval(): no dropout / Gumbel noise, stage fixed -- the step is a deterministic function of its batch
    synthetic.condition_for_large_graphs(m)
    return synthetic.condition_model_inputs(m).to(dev)


def _shard(rank, dev):
    from fabind_amd import parallel, synthetic
    sizes = _sizes()
    idx = parallel.shard_complexes(N_GLOBAL, rank, WORLD, weights=[p * c for p, c in sizes])
    return idx, synthetic.make_hetero_batch([sizes[i] for i in idx], seeds=[500 + int(i) for i in idx]).to(dev)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, q):
    import torch.distributed as dist
    from fabind_amd import engine, parallel
    from fabind_amd.models.model import compute_loss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)
    engine.set_precision("bf16")
    m = _model(dev)
    idx, data = _shard(rank, dev)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=LR)
    red = parallel.GradReducer(params, WORLD)
    losses = []
    for _ in range(STEPS):
        r = parallel.train_step(m, data.clone(), opt, compute_loss, WORLD, clip=1.0, stage=1, reducer=red)
        assert r is not None
        losses.append(float(r[0]))
    early = red.issued_early
    red.close()
    q.put((rank, [int(i) for i in idx], losses, early, [p.detach().float().cpu().numpy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_of_ragged_pocket_batches_equals_mean_of_shard_gradients():
    import torch.multiprocessing as mp
    from fabind_amd import engine, parallel
    from fabind_amd.models.model import compute_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    # the single-process emulation runs while the two ranks train (same device)
    dev = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)
    engine.set_precision("bf16")
    try:
        m = _model(dev)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=LR)
        shards = [_shard(r, dev) for r in range(WORLD)]
        counts = [len(s[0]) for s in shards]
        assert counts == [N_GLOBAL // WORLD] * WORLD                               # equal counts, like the reference's DistributedSampler
        sizes = _sizes()
        work = [sum(sizes[i][0] * sizes[i][1] for i in s[0]) for s in shards]
        assert max(work) <= 1.05 * min(work), work                                 # pair work balanced by the weighted deal
        ref_losses = [[] for _ in range(WORLD)]
        for _ in range(STEPS):
            grads = []
            for r, (_, data) in enumerate(shards):
                for p in params:
                    p.grad = None
                dc = data.clone()
                out = m(dc, stage=1, train=True)
                loss, _ = compute_loss(out, dc)
                loss.backward()
                ref_losses[r].append(float(loss.detach()))
                grads.append([None if p.grad is None else p.grad.detach().clone() for p in params])
            for i, p in enumerate(params):
                gs = [g[i] if g[i] is not None else torch.zeros_like(p) for g in grads]
                p.grad = (gs[0].float() + gs[1].float()).div_(WORLD).to(p.dtype)
            parallel.clip_grad_norm_(params, 1.0)
            opt.step()
        want = [p.detach().float().cpu().numpy() for p in params]
        names = [n for n, p in m.named_parameters() if p.requires_grad]
    finally:
        engine.set_precision("fp32")
    res = sorted([q.get(timeout=900) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, idx, losses, early, got in res:
        assert idx == [int(i) for i in shards[rank][0]]
        assert np.allclose(losses, ref_losses[rank], rtol=1e-5), (rank, losses, ref_losses[rank])
        gaps = sorted(((float(np.abs(g - w).max()) / max(1e-6, float(np.abs(w).max())), n) for g, w, n in zip(got, want, names)), reverse=True)
        worst = gaps[0][0]
        if worst > 1e-5:
            print("rank %d: parameters that differ from the emulation:" % rank, [(n, "%.2e" % v) for v, n in gaps[:8] if v > 0])
        print("rank %d: 3 DP steps, losses %s; parameters vs single-process mean-of-shards emulation: worst relative gap %.2e; "
              "buckets on the wire before the end of the last backward: %d" % (rank, ["%.5f" % v for v in losses], worst, early))
        assert worst <= 1e-5
        assert early >= 1                                                          # the discovery step is over: the reducer overlaps
    assert all(np.array_equal(a, b) for a, b in zip(res[0][4], res[1][4]))         # ranks hold identical weights
    assert ref_losses[0][-1] != ref_losses[0][0]                                   # the weights did move
