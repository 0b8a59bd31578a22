"""The kernel-side parameter pack as ONE launch each way.

The kernels want the nn.Module parameters in other layouts than the reference's `__init__` leaves them in (models/egnn.py:40-60,
186-208, models/cross_att.py:15-40, models/model_utils.py:83-131): the first edge Linear split per node, q | k | v with the interleaved
kv split undone, zero-padded 32-wide Hadamard projections, bf16 copies.  Built from torch ops that is ~200 launches of a few
microseconds per model call and ~250 more in its adjoint -- a quarter of all launches of a training step at the bench shape, half
at the pocket shape.  `ParamPack` collects the copy-type requests (concatenation of parameter views / zero blocks along one axis,
with a cast) and executes them as one `fabind_multi_copy` launch; under autograd it is one node whose backward is one more launch
writing every parameter gradient slice.  Products of parameters (composed weights) stay ordinary torch ops on the pack's outputs.

A parameter element may appear in at most ONE request (its gradient slice is written, not accumulated): the builders in engine.py
keep to that, and `FABIND_PARAM_PACK=check` verifies it on every call."""
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, dt_code, stream

_SEG = np.dtype([("src", np.uint64), ("dst", np.uint64), ("src_sr", np.int64), ("src_sc", np.int64), ("dst_sr", np.int64),
                 ("dst_sc", np.int64), ("rows", np.int32), ("cols", np.int32), ("src_dt", np.int32), ("dst_dt", np.int32),
                 ("vec4", np.int32), ("pad", np.int32)])
_ESZ = {torch.float32: 4, torch.bfloat16: 2}
_RING, _RING_POS, _RING_EVT = [], [0], []
_RING_SLOTS = 64          # (several small index tables per step share the ring with the two pack tables; a slot is rewritten only after its copy ran)


def _upload(table, dev):
    """Device copy of a segment table through a ring of pinned staging buffers (asynchronous, no stream drain).  Every slot
    carries the event recorded behind its last host-to-device copy; a slot is rewritten only after that copy has run, so a
    host that gets more than a ring's worth of pack launches ahead of the device (a training loop without a per-step sync)
    waits here instead of overwriting a table the device has not read yet."""
    raw = table.view(np.uint8)
    n = raw.shape[0]
    if not _RING:
        _RING.extend(torch.empty(64 * 1024, dtype=torch.uint8, pin_memory=True) for _ in range(_RING_SLOTS))
        _RING_EVT.extend([None] * _RING_SLOTS)
    k = _RING_POS[0] = (_RING_POS[0] + 1) % len(_RING)
    if n > _RING[k].numel():
        if _RING_EVT[k] is not None:
            _RING_EVT[k].synchronize()          # the old (smaller) buffer may still be the source of a queued copy
        _RING[k] = torch.empty(max(n, 2 * _RING[k].numel()), dtype=torch.uint8, pin_memory=True)
    elif _RING_EVT[k] is not None:
        _RING_EVT[k].synchronize()
    stage = _RING[k]
    stage[:n].numpy()[:] = raw
    out = stage[:n].to(dev, non_blocking=True)
    if out.is_cuda:
        _RING_EVT[k] = torch.cuda.Event()
        _RING_EVT[k].record(torch.cuda.current_stream(out.device))
    return out


def upload(arr, dev, dtype):
    """Device copy of a small contiguous numpy array as a tensor of `dtype`, ASYNCHRONOUS (pinned staging ring): a plain
    `torch.from_numpy(a).to(device)` / `torch.tensor(list, device=...)` is a pageable copy, which waits for everything the stream still
    has queued -- index tables built in the middle of a step (group descriptors, tile maps, pair offsets) used to drain the device."""
    if torch.device(dev).type != "cuda":                    # host tensors (CPU-side tests of index logic): nothing to stage
        return torch.from_numpy(np.ascontiguousarray(arr).copy()).view(dtype).view(arr.shape)
    flat = np.ascontiguousarray(arr).reshape(-1)
    return _upload(flat, dev).view(dtype).view(arr.shape)


def _launch(table, dev):
    if len(table) == 0:
        return
    tdev = _upload(table, dev)
    big = int((table["rows"].astype(np.int64) * table["cols"]).max())
    blocks = max(1, min(64, (big + 4095) // 4096))
    check(_lib.load().fabind_multi_copy(tdev.data_ptr(), len(table), blocks, stream()), "fabind_multi_copy")
    return tdev                                  # the caller keeps it alive until the launch has been queued


class _H:
    """Handle of a ParamPack request (replaced by its tensor in `ParamPack.resolve`)."""
    __slots__ = ("k",)

    def __init__(self, k):
        self.k = k


class EagerPack:
    """The same request interface executed immediately with torch ops (the reference behaviour: one launch or more per request;
    used by the single-module entry points and by FABIND_PARAM_PACK=0)."""
    def __init__(self, device):
        self.dev = device

    def cat(self, pieces, dim=0, dtype=None, with_T=False):
        first = next(p for p in pieces if not isinstance(p, tuple))
        ts = [torch.zeros(p[1:], dtype=first.dtype, device=self.dev) if isinstance(p, tuple) else p for p in pieces]
        t = ts[0] if len(ts) == 1 else torch.cat(ts, dim)
        return (t.to(dtype) if dtype is not None else t).contiguous()

    def copy(self, view, dtype=None, with_T=False):
        return self.cat([view], 0, dtype)

    def pad2d(self, view, rows, cols, dtype=None, with_T=False):
        out = torch.zeros((rows, cols), dtype=view.dtype, device=self.dev)
        out[:view.shape[0], :view.shape[1]] = view
        return out.to(dtype) if dtype is not None else out

    zeros = staticmethod(lambda *shape: ("zeros",) + tuple(shape))

    def resolve(self, tree):
        return tree


class ParamPack:
    def __init__(self, device):
        self.dev = device
        self.reqs = []           # (ndim, dtype, dim, [piece]) with piece = tensor view | ("zeros", *shape)
        self.t_of = {}           # request -> request holding its transpose

    def resolve(self, tree):
        """Run the collected requests and replace every handle in a nested dict / list structure by its tensor."""
        outs = self.run()
        for k, kt in self.t_of.items():                        # W._fab_T = W^T, written by the same launch (the adjoints' operand)
            outs[k]._fab_T = outs[kt]

        def walk(x):
            if isinstance(x, _H):
                return outs[x.k]
            if isinstance(x, dict):
                return {k: walk(v) for k, v in x.items()}
            if isinstance(x, list):
                return [walk(v) for v in x]
            return x
        return walk(tree)

    # ---- requests ------------------------------------------------------------------------------------------------
    def cat(self, pieces, dim=0, dtype=None, with_T=False):
        """Handle of cat(pieces, dim) cast to dtype; a piece is a 1-D / 2-D view of a parameter or ParamPack.zeros(...).
        with_T (2-D): the same launch also writes the transpose, attached to the result as `._fab_T` (the input-gradient
        contraction of a Linear wants W^T; nothing differentiates through it)."""
        first = next(p for p in pieces if not isinstance(p, tuple))
        nd = first.dim()
        assert nd in (1, 2) and (dim == 0 or nd == 2)
        dtype = dtype or first.dtype
        self.reqs.append((nd, dtype, dim, list(pieces)))
        k = len(self.reqs) - 1
        if with_T and nd == 2:
            tp = [("zeros", p[2], p[1]) if isinstance(p, tuple) else p.t() for p in pieces]
            self.reqs.append((2, dtype, 1 - dim, tp))
            self.t_of[k] = k + 1
        return _H(k)

    def copy(self, view, dtype=None, with_T=False):
        return self.cat([view], 0, dtype, with_T)

    def pad2d(self, view, rows, cols, dtype=None, with_T=False):
        """Handle of a [rows, cols] tensor holding the 2-D view in its top-left corner, zeros elsewhere (the zero-padded weights of
        the FABind+ LN-MLPs: contraction dims padded to the GEMM's K granularity).  with_T as in `cat`."""
        r, c = view.shape
        assert view.dim() == 2 and r <= rows and c <= cols
        dtype = dtype or view.dtype
        self.reqs.append((2, dtype, "pad", [view, ("zeros", r, cols - c), ("zeros", rows - r, cols)]))
        k = len(self.reqs) - 1
        if with_T:
            self.reqs.append((2, dtype, "pad", [view.t(), ("zeros", c, rows - r), ("zeros", cols - c, rows)]))
            self.t_of[k] = k + 1
        return _H(k)

    @staticmethod
    def zeros(*shape):
        return ("zeros",) + tuple(shape)

    # ---- execution -----------------------------------------------------------------------------------------------
    def run(self):
        """-> list of output tensors (one per request), differentiable w.r.t. the parameters behind the views."""
        if getattr(self, "_n_planned", -1) != len(self.reqs):       # (a pack that is run again keeps its base list)
            bases, seen = [], {}
            for _, _, _, pieces in self.reqs:
                for p in pieces:
                    if isinstance(p, tuple):
                        continue
                    b = p._base if p._base is not None else p
                    if id(b) not in seen:
                        seen[id(b)] = len(bases)
                        bases.append(b)
            self._bases, self._base_idx, self._n_planned = bases, seen, len(self.reqs)
            self._fwd = self._bwd = None
        bases = self._bases
        if torch.is_grad_enabled() and any(b.requires_grad for b in bases):
            return list(_PackFn.apply(self, *bases))
        return self._forward()

    def _layout(self):
        """Per request: output shape and per piece (r0, c0, rows, cols)."""
        lay = []
        for nd, dtype, dim, pieces in self.reqs:
            shapes = []
            for p in pieces:
                sh = tuple(p[1:]) if isinstance(p, tuple) else tuple(p.shape)
                shapes.append((1, sh[0]) if nd == 1 else sh)
            if dim == "pad":                 # [view | zeros right] over [zeros bottom]
                (r, c), (_, cr), (rb, cb) = shapes
                lay.append(((r + rb, c + cr), [(0, 0, r, c), (0, c, r, cr), (r, 0, rb, cb)]))
                continue
            if nd == 1 or dim == 1:
                R = shapes[0][0]
                assert all(s[0] == R for s in shapes)
                offs, c = [], 0
                for s in shapes:
                    offs.append((0, c))
                    c += s[1]
                out = (R, c)
            else:
                Cc = shapes[0][1]
                assert all(s[1] == Cc for s in shapes)
                offs, r = [], 0
                for s in shapes:
                    offs.append((r, 0))
                    r += s[0]
                out = (r, Cc)
            lay.append((out, [(o[0], o[1], s[0], s[1]) for o, s in zip(offs, shapes)]))
        return lay

    @staticmethod
    def _strides(v):
        return (0, v.stride(0)) if v.dim() == 1 else (v.stride(0), v.stride(1))

    # The table of a launch is (static part) + (addresses of this call's outputs / incoming gradients): the static part -- which rows
    # exist, their shapes, strides, dtypes, parameter addresses, offsets inside an output -- is computed ONCE per pack and kept
    # (`_plan_fwd` / `_plan_bwd`), so a pack that is run again (engine.py keeps the pack of a model across training steps) costs one
    # torch.empty per output, one vectorised address computation and one launch instead of ~1,000 Python-level tensor queries.
    def _plan_forward(self):
        lay = self._layout()
        specs, rows, req, rel = [], [], [], []
        for k, ((nd, dtype, dim, pieces), (oshape, offs)) in enumerate(zip(self.reqs, lay)):
            specs.append((oshape if nd == 2 else (oshape[1],), dtype))
            ld, esz_o = oshape[1], _ESZ[dtype]
            for p, (r0, c0, nr, nc) in zip(pieces, offs):
                if nr == 0 or nc == 0:                 # (an empty zero strip of a pad2d request)
                    continue
                d_rel = (r0 * ld + c0) * esz_o
                req.append(k)
                rel.append(d_rel)
                if isinstance(p, tuple):
                    rows.append((0, 0, 0, 0, ld, 1, nr, nc, 0, dt_code(dtype), 0))
                    continue
                sr, sc = self._strides(p)
                # (outputs are fresh allocations: 256-byte aligned, so the alignment of a destination is that of its offset)
                v4 = int(sc == 1 and nc % 4 == 0 and (sr % 4 == 0 or nr == 1) and ld % 4 == 0 and p.data_ptr() % (4 * _ESZ[p.dtype]) == 0
                         and d_rel % (4 * esz_o) == 0)
                rows.append((p.data_ptr(), 0, sr, sc, ld, 1, nr, nc, dt_code(p.dtype), dt_code(dtype), v4))
        self._lay = lay
        self._fwd = (specs, self._table(rows), np.asarray(req, dtype=np.int64), np.asarray(rel, dtype=np.uint64))
        self._fwd_key = tuple(b.data_ptr() for b in self._bases)

    def _forward(self):
        if getattr(self, "_fwd", None) is None or self._fwd_key != tuple(b.data_ptr() for b in self._bases):
            self._plan_forward()
            self._bwd = None
        specs, table, req, rel = self._fwd
        outs = [torch.empty(sh, dtype=dt, device=self.dev) for sh, dt in specs]
        if len(table):
            optr = np.fromiter((o.data_ptr() for o in outs), dtype=np.uint64, count=len(outs))
            t = table.copy()
            t["dst"] = optr[req] + rel
            self._keep = _launch(t, self.dev)
        return outs

    @staticmethod
    def _table(rows):
        t = np.zeros(len(rows), dtype=_SEG)
        for k, name in enumerate(("src", "dst", "src_sr", "src_sc", "dst_sr", "dst_sc", "rows", "cols", "src_dt", "dst_dt", "vec4")):
            t[name] = [r[k] for r in rows]
        return t

    def _plan_backward(self):
        bases = self._bases
        sizes = [b.numel() for b in bases]
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        rows, req, rel, esz = [], [], [], []
        t_reqs = set(self.t_of.values())
        check_overlap = os.environ.get("FABIND_PARAM_PACK", "") == "check"
        cover = [np.zeros(n, dtype=np.int8) for n in sizes] if check_overlap else None
        for k_req, ((nd, dtype, dim, pieces), (oshape, offs)) in enumerate(zip(self.reqs, self._lay)):
            if k_req in t_reqs:                          # (nothing differentiates through the transposed copies)
                continue
            ld = oshape[1]
            for p, (r0, c0, nr, nc) in zip(pieces, offs):
                if isinstance(p, tuple):
                    continue
                b = p._base if p._base is not None else p
                k = self._base_idx[id(b)]
                if not b.requires_grad:
                    continue
                assert b.is_contiguous() and b.dtype == torch.float32, "ParamPack: parameters are contiguous fp32 tensors"
                off = p.storage_offset() - b.storage_offset()
                sr, sc = self._strides(p)
                d_rel = (int(starts[k]) + off) * 4
                vstat = int(sc == 1 and nc % 4 == 0 and (sr % 4 == 0 or nr == 1) and ld % 4 == 0 and d_rel % 16 == 0)
                rows.append((0, d_rel, ld, 1, sr, sc, nr, nc, 0, dt_code(torch.float32), vstat))
                req.append(k_req)
                rel.append(r0 * ld + c0)                  # in ELEMENTS of the incoming gradient (its dtype is known per call)
                if check_overlap:
                    idx = (off + np.arange(nr)[:, None] * sr + np.arange(nc)[None, :] * sc).reshape(-1)
                    cover[k][idx] += 1
        if check_overlap:
            for k, c in enumerate(cover):
                assert c.max(initial=0) <= 1, "ParamPack: a parameter element is requested twice (its gradient would be overwritten)"
        self._bwd = (sizes, starts, self._table(rows), np.asarray(req, dtype=np.int64), np.asarray(rel, dtype=np.int64))

    def _backward(self, gouts):
        """Gradients of the base parameters: every requested view's slice of its parameter's gradient receives the matching block of
        the output gradient (fp32); parameter elements outside every request stay zero."""
        if getattr(self, "_bwd", None) is None:
            self._plan_backward()
        sizes, starts, table, req, rel = self._bwd
        bases = self._bases
        flat = torch.zeros(int(starts[-1]), dtype=torch.float32, device=self.dev)
        n_req = len(self.reqs)
        gptr = np.zeros(n_req, dtype=np.uint64)
        gesz = np.ones(n_req, dtype=np.int64)
        gdt = np.zeros(n_req, dtype=np.int32)
        have = np.zeros(n_req, dtype=bool)
        keep_g = []
        for k_req, g in enumerate(gouts):
            if g is None:
                continue
            g = g.contiguous()
            keep_g.append(g)                      # (alive until the launch below has been queued)
            gptr[k_req], gesz[k_req], gdt[k_req], have[k_req] = g.data_ptr(), _ESZ[g.dtype], dt_code(g.dtype), True
        if len(table):
            sel = have[req]
            t = table[sel].copy()
            r = req[sel]
            src = gptr[r] + (rel[sel] * gesz[r]).astype(np.uint64)
            t["src"] = src
            t["dst"] = t["dst"] + np.uint64(flat.data_ptr())
            t["src_dt"] = gdt[r]
            t["vec4"] = t["vec4"] & ((src % (4 * gesz[r]).astype(np.uint64)) == 0)
            self._keep_b = _launch(t, self.dev) if len(t) else None
        return [flat[int(starts[k]):int(starts[k + 1])].view(bases[k].shape) if bases[k].requires_grad else None for k in range(len(bases))]


class _PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack, *bases):
        ctx.pack = pack
        ctx.set_materialize_grads(False)          # an unused output's gradient stays None (a materialised zero block would be WRITTEN)
        return tuple(pack._forward())

    @staticmethod
    def backward(ctx, *gouts):
        from . import kernels as K
        K.tn_flush()                              # queued weight-gradient contractions write the tensors this adjoint is about to read
        return (None,) + tuple(ctx.pack._backward(gouts))
