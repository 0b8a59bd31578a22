"""Data-parallel training step: one process per GPU, complexes sharded across ranks, one gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

Replaces accelerate -> torch DDP (reference main_fabind.py:194-195, 289-296, 419-423):
  * DDP semantics are kept: the result is the MEAN over ranks of the per-rank gradients (each rank's loss
    is a mean over its own atoms / pairs / residues -- SURVEY.md section 4, 8(e)); do not "fix" this.
  * 33 parameter tensors never receive a gradient (att_i.inter_layer.*, pocket model out_layer.coord_mlp.*;
    the reference needs find_unused_parameters=True).  Here every rank packs the SAME flat buffer over
    all trainable parameters and fills missing gradients with zeros, so no graph inspection is needed.
  * `GradReducer` overlaps the collective with backward (buckets are issued as they fill).  The first step is the discovery step, as
    in DDP: it records the order in which gradients arrive and which parameters never get one, then rebuilds the buckets in arrival
    order with the never-used tensors in a trailing bucket that no bucket waits for -- from the second step on every bucket but the
    last leaves while backward is still running (the first step reduces after backward).  `allreduce_gradients` is the
    non-overlapped form of the same arithmetic.
  * the NaN-skip of a step (main_fabind.py:394-396) is agreed on collectively, so no rank waits in a collective alone.
  * xGMI is point-to-point (7 links x ~153 GB/s): the 145 MB fp32 gradient is sent as a few large buckets
    (default 64 MB) so that RCCL can use reduce-scatter + all-gather across all links; the collective is
    issued on RCCL's own stream and overlaps with the packing of the next bucket.  `bucket_dtype=torch.bfloat16` halves the
    bytes on the links (73 MB; the sum over ranks is then rounded to bf16 -- SURVEY 8(e) allows it, fp32 is the default).
  * `shard_complexes(..., weights=P*C)` deals the complexes of a global batch to the ranks length-bucketed: equal counts (+-1, what the
    reference's DistributedSampler gives, main_fabind.py:289-296) and balanced pair work, so that no rank waits for the one that drew
    the large proteins.
  * gradient clipping (max_norm 1.0) happens after the all-reduce on the full gradient (main_fabind.py:420-423).
"""
import torch
import torch.distributed as dist


def shard_complexes(n_complexes, rank, world, weights=None):
    """Shard of a global batch of complexes for `rank`.
    weights=None: the contiguous, balanced range [lo, hi).
    weights = per-complex work (P*C: the pair path is 62 % of the flops): -> sorted int64 index array of this rank's
    complexes -- every rank gets n // world complexes (+1 for the first n % world ranks, like the contiguous form) and the sums of
    weights are balanced greedily: complexes in decreasing weight, each to the rank with the least work among those that still have
    room (longest-processing-time rule with a count cap).  Deterministic (stable sort, ties to the lower rank), so every rank
    computes the same deal without communicating."""
    base, rem = divmod(n_complexes, world)
    if weights is None:
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    import numpy as np
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    assert w.shape[0] == n_complexes, "shard_complexes: one weight per complex"
    room = [base + (1 if r < rem else 0) for r in range(world)]
    load = [0.0] * world
    mine = []
    for i in np.argsort(-w, kind="stable"):
        r = min((r_ for r_ in range(world) if room[r_] > 0), key=lambda r_: (load[r_], r_))
        room[r] -= 1
        load[r] += float(w[i])
        if r == rank:
            mine.append(int(i))
    return np.sort(np.asarray(mine, dtype=np.int64))


def _buckets(params, bucket_bytes):
    cur, size = [], 0
    for p in params:
        n = p.numel() * 4
        if cur and size + n > bucket_bytes:
            yield cur
            cur, size = [], 0
        cur.append(p)
        size += n
    if cur:
        yield cur


def _staged(flat, group):
    """The tensor the collective backend can reduce for `flat`: itself under RCCL; under gloo (the CPU-only backend of the tests) a host
    copy, widened to fp32 when the bucket is bf16 (gloo has no bf16 sum; the operands are bf16-rounded all the same).  One helper for
    the overlapped and the plain path, so that both reduce the same numbers on every backend."""
    if dist.get_backend(group) == "gloo":
        if flat.is_cuda:
            flat = flat.cpu()
        if flat.dtype == torch.bfloat16:
            flat = flat.float()
    return flat


def allreduce_gradients(params, world=None, bucket_bytes=64 << 20, group=None, bucket_dtype=torch.float32):
    """Average .grad of `params` over all ranks (flat buckets of bucket_dtype, fp32 by default; missing grads count as zeros)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    params = [p for p in params if p.requires_grad]
    if world == 1:
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        return
    pending = []
    for bucket in _buckets(params, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(bucket_dtype) for p in bucket])
        dev = flat.device
        flat = _staged(flat, group)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending.append((bucket, flat, work, dev))
    for bucket, flat, work, dev in pending:
        work.wait()
        flat = flat.to(dev).float()
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def clip_grad_norm_(params, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ semantics on the (already all-reduced) gradients."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float()) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef.to(g.dtype))
    return total


class GradReducer:
    """Gradient all-reduce overlapped with backward (what DDP's reducer does for the reference, main_fabind.py:194-195,
    419-423), sized for xGMI: the trainable parameters are cut into a few LARGE flat buckets (default 64 MB of fp32 -- ring
    collectives over 7 point-to-point links are per-link bound, so few big messages beat many small ones).  A post-accumulate
    hook on every parameter copies its gradient into its bucket; a bucket's all-reduce is issued asynchronously (RCCL runs it
    on its own stream) the moment its last EXPECTED gradient has arrived, while backward continues with the earlier layers.
    `finish()` issues what is left, waits, divides by the world size and writes the averaged gradients back.

    Which gradients to expect is learned, as in DDP (`find_unused_parameters=True` + the bucket rebuild after the first
    iteration): the FIRST step runs on provisional buckets in reverse registration order and cannot overlap on the real model --
    33 of its tensors never receive a gradient (`att_i.inter_layer.*`, the pocket model's `out_layer.coord_mlp.*`)
    and they sit in the leading buckets, whose count therefore never reaches zero before `finish()`.  At the end of that step
    rank 0's arrival order is broadcast and every rank rebuilds the same buckets: parameters in the order their gradients
    arrived, then the never-used ones in trailing bucket(s) that are zero-filled once and that no bucket waits for.  From the
    second step on a bucket leaves as soon as its used parameters have reported.  A parameter from the never-used set that does
    get a gradient later (another stage, another loss) -- on ANY rank -- is copied like any other; if its bucket has already left
    it is reduced again in `finish()`, and the parameter is expected from then on ON EVERY RANK: membership in the never-used set
    is collective (the per-parameter "fired late" flags travel in the same tiny MAX all-reduce as the stale-bucket mask), so a
    rank on which it never fires zero-fills its slot in every later step instead of re-sending the previous step's average
    (the buffers are reduced and divided in place).

    Every rank issues the same collectives in the same order: bucket k is issued only after buckets 0..k-1, and the set of
    buckets reduced a second time (gradient accumulation over several backward() calls, late parameters) is agreed on by one
    tiny MAX all-reduce in `finish()`.
    bucket_dtype: torch.float32 (default) or torch.bfloat16 (half the bytes on the links; the sum is rounded to bf16).
    check_late=False: the caller promises ONE backward per step and no late parameters after the discovery step -- `finish()` then
    skips the agreement collective (and its host read-back); a second accumulation or a late parameter raises instead."""

    def __init__(self, params, world=None, bucket_bytes=64 << 20, group=None, bucket_dtype=torch.float32, rebuild=True, check_late=True):
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.params = [p for p in params if p.requires_grad]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.bucket_bytes, self.bucket_dtype = bucket_bytes, bucket_dtype
        self.rebuild = bool(rebuild)
        self.check_late = bool(check_late)
        self.steps_done = 0
        self.unused = set()            # ids of parameters no bucket waits for (never fired in the discovery step)
        self.issued_early = 0          # buckets of the LAST finished step whose collective left before finish() (diagnostic, tests)
        self._layout(list(reversed(self.params)))
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params] if self.world > 1 else []
        self._reset()

    def _layout(self, ordered):
        """Cut `ordered` (every trainable parameter once) into buckets; parameters in self.unused are not waited for."""
        self.buckets = [list(b) for b in _buckets(ordered, self.bucket_bytes)]
        self.slot = {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self.slot[id(p)] = (bi, off)
                off += p.numel()
        self.sizes = [sum(p.numel() for p in b) for b in self.buckets]
        self.flat = [None] * len(self.buckets)

    def _reset(self):
        self.pending = [sum(1 for p in b if id(p) not in self.unused) for b in self.buckets]
        self.work = [None] * len(self.buckets)
        self.seen = set()
        self.order = []            # parameter indices in arrival order (this step)
        self.stale = set()         # buckets holding a parameter whose .grad changed after the bucket's collective left
        self.late = set()          # indices of never-used parameters whose hook fired this step (on this rank)
        self.next_issue = 0
        self._early = 0

    def abort(self):
        """Abandon the current step (backward raised, or the step is skipped after some hooks fired): wait for the collectives
        already issued -- every rank issues the same ones, so this cannot hang as long as all ranks abort the same step -- and
        forget the partial state.  The .grad fields are left as backward left them."""
        for w in self.work:
            if w is not None:
                w[0].wait()
        for bi, b in enumerate(self.buckets):                  # slots of never-used parameters must read zero next step
            if self.flat[bi] is not None and any(id(p) in self.unused for p in b):
                self.flat[bi].zero_()
        self._reset()

    def _buffer(self, bi):
        if self.flat[bi] is None:
            dev = self.buckets[bi][0].device
            self.flat[bi] = torch.zeros(self.sizes[bi], dtype=self.bucket_dtype, device=dev)
        return self.flat[bi]

    def _hook(self, p):
        bi, off = self.slot[id(p)]
        if id(p) in self.seen:
            # A second accumulation into the same parameter before finish() (two backward() calls per step = gradient
            # accumulation, or a parameter reached twice by one backward): .grad now holds the SUM, the bucket the first
            # micro-batch only.  If the bucket is still here it is refreshed in place; if its collective has already left,
            # it is marked stale and finish() reduces it again from .grad (the first result is discarded).
            if not self.check_late and self.steps_done > 0:
                raise RuntimeError("GradReducer(check_late=False): a second accumulation into a parameter before finish()")
            if self.work[bi] is None:
                self._buffer(bi)[off:off + p.numel()].copy_(p.grad.reshape(-1))
            else:
                self.stale.add(bi)
            return
        self.seen.add(id(p))
        self.order.append(self.index[id(p)])
        if id(p) in self.unused:
            # a tensor the discovery step never saw a gradient for: no bucket counted on it.  It leaves the never-used set in
            # finish(), on every rank together (self.late is OR-ed over the ranks there), and is expected from the next step on.
            if not self.check_late:
                raise RuntimeError("GradReducer(check_late=False): a never-used parameter received a gradient after the discovery step")
            self.late.add(self.index[id(p)])
            if self.work[bi] is None:
                self._buffer(bi)[off:off + p.numel()].copy_(p.grad.reshape(-1))
            else:
                self.stale.add(bi)
            return
        self._buffer(bi)[off:off + p.numel()].copy_(p.grad.reshape(-1))
        self.pending[bi] -= 1
        self._issue_ready(True)

    def _issue_ready(self, early=False):
        while self.next_issue < len(self.buckets) and self.pending[self.next_issue] == 0:
            self._issue(self.next_issue)
            self.next_issue += 1
            self._early += 1 if early else 0

    def _issue(self, bi):
        flat = _staged(self._buffer(bi), self.group)
        self.work[bi] = (dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat)

    def _agree(self, flags):
        """Element-wise OR over ranks of a small list of 0/1 flags (one tiny collective; every rank calls it once per finish())."""
        t = torch.tensor(flags, dtype=torch.int32)
        if dist.get_backend(self.group) != "gloo":
            t = t.to(self.params[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [bool(v) for v in t.tolist()]

    def finish(self):
        """Call after the step's LAST backward: completes the all-reduce and leaves the rank-averaged gradient in every .grad.
        With gradient accumulation (several backward() calls before finish()) the result is the average of the accumulated
        gradients: buckets whose collective left before a later micro-batch added to one of their parameters are reduced
        again from .grad."""
        if self.world == 1:
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            return
        for bi, b in enumerate(self.buckets):                  # expected parameters whose hook never fired this step: zeros
            if self.work[bi] is None and self.pending[bi] > 0:
                buf = self._buffer(bi)
                for p in b:
                    if id(p) not in self.seen and id(p) not in self.unused:
                        _, off = self.slot[id(p)]
                        buf[off:off + p.numel()].zero_()
                self.pending[bi] = 0
        self.issued_early = self._early
        self._issue_ready()
        # the buckets to reduce again must be the same on every rank (a late parameter may have fired on one rank only), and so must
        # the never-used set: one collective carries the stale-bucket mask and a "fired late" flag per never-used parameter
        nb = len(self.buckets)
        unused_idx = sorted(self.index[i] for i in self.unused)            # same list on every rank (the set only changes below)
        if self.check_late or self.steps_done == 0:
            flags = self._agree([1 if bi in self.stale else 0 for bi in range(nb)] + [1 if i in self.late else 0 for i in unused_idx])
        else:
            flags = [False] * (nb + len(unused_idx))
        stale = flags[:nb]
        for i, fired in zip(unused_idx, flags[nb:]):
            if fired:
                p = self.params[i]
                self.unused.discard(id(p))                     # expected from the next step on, on every rank
                # (this step's sum is already right on every rank: while a parameter is in the never-used set EVERYWHERE its slot only
                #  ever receives zeros -- 0 / world = 0 -- so a rank where it did not fire contributes 0; from the next step on the slot
                #  is an expected one: overwritten by the hook or zero-filled at the top of finish())
        for bi in [i for i, v in enumerate(stale) if v]:       # same order on every rank
            self.work[bi][0].wait()                            # (its result is superseded)
            buf = self._buffer(bi)
            for p in self.buckets[bi]:
                _, off = self.slot[id(p)]
                if p.grad is None or (id(p) not in self.seen):
                    buf[off:off + p.numel()].zero_()
                else:
                    buf[off:off + p.numel()].copy_(p.grad.reshape(-1))
            self._issue(bi)
        for bi, b in enumerate(self.buckets):
            work, flat = self.work[bi]
            work.wait()
            own = self.flat[bi]
            if flat is not own:                                # staged through the host and / or widened (gloo)
                own.copy_(flat.to(own.device))
            res = own if own.dtype == torch.float32 else own.float()
            res.div_(self.world)                               # (slots of never-used parameters hold 0 / world = 0: nothing to restore)
            off = 0
            for p in b:
                n = p.numel()
                g = res[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        order = self.order
        self.steps_done += 1
        if self.rebuild and self.steps_done == 1:
            self._rebuild(order)
        self._reset()

    def _rebuild(self, order):
        """End of the discovery step: rank 0's arrival order becomes the bucket order on every rank (never-used tensors last)."""
        n = len(self.params)
        t = torch.full((n,), -1, dtype=torch.int64)
        t[:len(order)] = torch.tensor(order, dtype=torch.int64) if order else t[:0]
        if dist.get_backend(self.group) != "gloo":
            t = t.to(self.params[0].device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        fired = [int(v) for v in t.tolist() if v >= 0]
        fired_set = set(fired)
        rest = [i for i in range(n - 1, -1, -1) if i not in fired_set]          # reverse registration order, like the provisional layout
        self.unused = {id(self.params[i]) for i in rest}
        self._layout([self.params[i] for i in fired] + [self.params[i] for i in rest])

    def close(self):
        for h in self.handles:
            h.remove()
        self.handles = []


def all_ranks_agree_to_skip(bad, world=None, group=None):
    """True on EVERY rank when `bad` (a bool / 0-d tensor: this rank saw a NaN) is true on ANY rank.  The reference skips a
    step on its own NaN only (main_fabind.py:394-396) because accelerate's DDP would otherwise hang the other ranks in the
    gradient all-reduce; here the decision is made collectively, so all ranks skip the same steps."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    flag = bad if torch.is_tensor(bad) else torch.tensor(bool(bad))
    if world == 1:
        return bool(flag)
    flag = flag.to(torch.int32).reshape(1)
    if flag.is_cuda and dist.get_backend(group) == "gloo":
        flag = flag.cpu()
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item())


def train_step(model, data, optimizer, compute_loss, world=1, clip=1.0, stage=1, reducer=None):
    """One DP step with the reference's order: forward, NaN guard, 6-term loss, backward, all-reduce,
    clip, optimizer step (main_fabind.py:392-426).  Returns (loss, terms) or None when the batch is skipped -- on ALL
    ranks together (the NaN flag is all-reduced first, so no rank is left waiting in the gradient collective).
    reducer: a GradReducer over the model's parameters (overlaps the all-reduce with backward); without one the
    gradients are reduced after backward."""
    out = model(data, stage=stage, train=True)
    bad = torch.stack([torch.isnan(t).any() for t in (out[0], out[2], out[3], out[4], out[8])]).any()
    if all_ranks_agree_to_skip(bad, world):
        return None
    loss, terms = compute_loss(out, data)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    params = [p for p in model.parameters() if p.requires_grad]
    if reducer is not None:
        reducer.finish()
    else:
        allreduce_gradients(params, world)
    if clip:
        clip_grad_norm_(params, clip)
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in terms.items()}
