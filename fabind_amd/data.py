"""Batch builder: raw per-complex arrays -> the collated `HeteroData` field contract, assembled ON THE DEVICE
(reference FABind/fabind/utils/utils.py:202-442 `construct_data_from_graph_gvp_mean` +
`get_keepNode` (121-144), applied per sample by data.py's Dataset and then collated by PyG's DataLoader).

The reference builds one python `HeteroData` per complex on the host (a python loop over residues for the pocket crop,
scipy `cdist`, a dozen `torch.cat`s) and collates them afterwards.  Here the raw arrays of a whole batch are packed into
two flat host buffers (one float, one integer), cross the bus once each, and every derived field -- protein centring,
pocket crop around the ligand centroid (+ the < 5 residues fallback), initial pose (RDKit conformer moved to the pocket
centre), the `[glb_c | ligand | glb_p | protein]` node layouts of both graphs with their segment / mask / is_global /
batch vectors, offset edge lists, the clamped pocket-ligand distance map, pocket labels -- is computed for all complexes
at once with batched index arithmetic on the GPU.  Production configuration of the reference only:
`compound_coords_init_mode='pocket_center_rdkit'`, `use_compound_com_as_pocket=True`, `use_whole_protein=False`,
no torsion noise / local_eval (those branches need RDKit)."""
import numpy as np
import torch

from .synthetic import HeteroBatch

FLOAT_FIELDS = ("protein_node_xyz", "protein_esm2_feat", "coords", "compound_node_features", "rdkit_coords")


def _offsets(cnt):
    out = torch.zeros(cnt.shape[0] + 1, dtype=torch.long, device=cnt.device)
    out[1:] = torch.cumsum(cnt, 0)
    return out


def _seg_mean(x, idx, n):
    """Per-complex mean of the rows of x; idx is a SORTED batch vector (complex-contiguous rows): the sums run in a fixed order
    (torch.segment_reduce: one thread walks a segment), not as float atomics in arrival order -- the batches this builder emits are
    bit-identical from run to run."""
    cnt = torch.bincount(idx, minlength=n)
    # (the builder reads sizes back anyway: the sortedness of idx and -- inside segment_reduce, unsafe=False -- the lengths are CHECKED;
    #  an unsorted or filtered batch vector raises instead of giving wrong centres)
    if idx.numel() > 1 and not bool((idx[1:] >= idx[:-1]).all()):
        raise ValueError("fabind_amd.data: the batch vector must be sorted (complex-contiguous rows)")
    return torch.segment_reduce(x, "sum", lengths=cnt, axis=0) / cnt.clamp(min=1).to(x.dtype)[:, None]


_POOL = {}


def _pinned(n, dtype, depth=3):
    """(page-locked host tensor of n elements, its pool slot) from a small rotating pool.  A slot is handed out again only
    after the device copy that last read it has completed (build_batch records an event into the slot)."""
    ring = _POOL.setdefault(dtype, {"slots": [{"buf": None, "ev": None} for _ in range(depth)], "i": 0})
    i = ring["i"] = (ring["i"] + 1) % depth
    slot = ring["slots"][i]
    if slot["ev"] is not None:
        slot["ev"].synchronize()
        slot["ev"] = None
    if slot["buf"] is None or slot["buf"].numel() < n:
        slot["buf"] = torch.empty(int(n * 1.1) + 16, dtype=dtype).pin_memory()
    return slot["buf"][:n], slot


def pack_samples(samples, pin=True):
    """Host side: flatten a list of raw complexes into (float32 buffer, int64 buffer, layout).  Each sample is a mapping
    with `protein_node_xyz [L,3]`, `protein_esm2_feat [L,Fp]`, `coords [Nc,3]` (native ligand), `compound_node_features
    [Nc,Fc]`, `rdkit_coords [Nc,3]`, `input_atom_edge_list [Eb,>=2]`, `LAS_edge_index [2,El]` (array-likes), `pdb`."""
    a = lambda v, dt: np.ascontiguousarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=dt)
    floats, lay = [], {"n": len(samples)}
    for f in FLOAT_FIELDS:
        parts = [a(s[f], np.float32) for s in samples]
        lay[f] = (sum(p.size for p in floats), [p.shape for p in parts])
        floats.extend(p.reshape(-1) for p in parts)
    bonds = [a(s["input_atom_edge_list"], np.int64)[:, :2] for s in samples]
    las = [a(s["LAS_edge_index"], np.int64).T for s in samples]
    counts = np.array([[s_["protein_node_xyz"].shape[0], s_["coords"].shape[0], b.shape[0], l.shape[0]]
                       for s_, b, l in zip(samples, bonds, las)], dtype=np.int64)
    ints = np.concatenate([counts.reshape(-1)] + [b.reshape(-1) for b in bonds] + [l.reshape(-1) for l in las])
    n_f = sum(p.size for p in floats)
    if pin and torch.cuda.is_available():
        # rotating pool of page-locked staging buffers (allocating + locking ~0.5 GB per batch would cost more than the copy)
        fbuf, s1 = _pinned(n_f, torch.float32)
        np.concatenate(floats, out=fbuf.numpy())
        ibuf, s2 = _pinned(ints.size, torch.int64)
        ibuf.numpy()[:] = ints
        lay["_slots"] = (s1, s2)
    else:
        fbuf, ibuf = torch.from_numpy(np.concatenate(floats)), torch.from_numpy(ints)
    lay["pdb"] = [s.get("pdb", "c%d" % i) for i, s in enumerate(samples)]
    return fbuf, ibuf, lay


@torch.no_grad()
def build_batch(samples, device, pocket_radius=20.0, interaction_threshold=10.0, add_noise_to_com=None,
                pocket_idx_no_noise=True, generator=None, packed=None):
    """-> HeteroBatch on `device` with every field of SURVEY.md A.10 (what `IaBNet.forward / inference`, the losses and the
    evaluation loop read).  `add_noise_to_com`: uniform +-noise on the crop centre (the reference's train-time option,
    utils/utils.py:129-130), drawn from `generator`."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("fabind_amd.data.build_batch assembles batches on a HIP device; got %s" % dev)
    fbuf, ibuf, lay = packed if packed is not None else pack_samples(samples)
    B = lay["n"]
    fd, idv = fbuf.to(dev, non_blocking=True), ibuf.to(dev, non_blocking=True)
    if "_slots" in lay:                                        # the staging buffers may be refilled once these copies are done
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        for slot in lay["_slots"]:
            slot["ev"] = ev

    def fl(name, width):
        start, shapes = lay[name]
        rows = sum(s[0] for s in shapes)
        return fd[start:start + rows * width].view(rows, width)
    fp, fc = lay["protein_esm2_feat"][1][0][1], lay["compound_node_features"][1][0][1]
    prot, esm = fl("protein_node_xyz", 3), fl("protein_esm2_feat", fp)
    lig, cfeat, rdk = fl("coords", 3), fl("compound_node_features", fc), fl("rdkit_coords", 3)
    cnt = idv[:4 * B].view(B, 4)
    L, Nc, Eb, El = cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3]
    nb, nl = int(Eb.sum()), int(El.sum())                      # the one host read-back (edge-list extents)
    bonds = idv[4 * B:4 * B + 2 * nb].view(nb, 2)
    las = idv[4 * B + 2 * nb:4 * B + 2 * nb + 2 * nl].view(nl, 2)
    ar = lambda n: torch.arange(n, device=dev)
    pb = torch.repeat_interleave(ar(B), L)
    cb = torch.repeat_interleave(ar(B), Nc)
    bb = torch.repeat_interleave(ar(B), Eb)
    lb = torch.repeat_interleave(ar(B), El)
    # ---- centre on the protein, crop the pocket around the ligand centroid (utils/utils.py:208-229, 121-144)
    bias = _seg_mean(prot, pb, B)
    prot, lig = prot - bias[pb], lig - bias[cb]
    com = _seg_mean(lig, cb, B)

    def crop(centre):
        k = (prot - centre[pb]).pow(2).sum(-1).sqrt() < pocket_radius
        few = torch.bincount(pb[k], minlength=B) < 5
        loc = ar(pb.shape[0]) - _offsets(L)[pb]
        return k | (few[pb] & (loc < 100))
    keep_plain = (prot - com[pb]).pow(2).sum(-1).sqrt() < pocket_radius
    if add_noise_to_com:
        noise = add_noise_to_com * (2 * torch.rand(B, 3, generator=generator).to(dev) - 1)     # host generator, like np.random
        keep = crop(com + noise)
    else:
        keep = crop(com)
    pocket_idx = (keep_plain if pocket_idx_no_noise else keep).to(torch.int32)
    pocket, pocket_b = prot[keep], pb[keep]
    Pk = torch.bincount(pocket_b, minlength=B)
    pocket_centre = _seg_mean(pocket, pocket_b, B)
    init = rdk - _seg_mean(rdk, cb, B)[cb] + pocket_centre[cb]          # 'pocket_center_rdkit' (utils/utils.py:315-321)
    init_w = init - _seg_mean(init, cb, B)[cb]

    data = HeteroBatch()
    data["compound"].node_feats, data["compound"].node_coords = cfeat, init
    data["compound"].rdkit_coords, data["compound"].batch = rdk, cb
    data["protein_whole"].node_feats, data["protein_whole"].batch = esm, pb
    data["pocket"].node_feats, data["pocket"].keepNode, data["pocket"].batch = esm[keep], keep, pocket_b
    cloc = ar(cb.shape[0]) - _offsets(Nc)[cb]
    for name, pr, prb, Pn, li in (("complex", pocket, pocket_b, Pk, init), ("complex_whole_protein", prot, pb, L, init_w)):
        n = Nc + Pn + 2
        off = _offsets(n)
        total = li.shape[0] + pr.shape[0] + 2 * B
        lig_pos = off[cb] + 1 + cloc
        prot_pos = off[prb] + Nc[prb] + 2 + (ar(prb.shape[0]) - _offsets(Pn)[prb])
        X = torch.zeros(total, 3, device=dev)
        X[lig_pos], X[prot_pos] = li, pr
        XL = torch.zeros(total, 3, device=dev)
        XL[lig_pos] = rdk
        seg = torch.zeros(total, device=dev)
        seg[prot_pos] = 1
        seg[off[:-1] + Nc + 1] = 1
        glb = torch.zeros(total, dtype=torch.bool, device=dev)
        glb[off[:-1]] = True
        glb[off[:-1] + Nc + 1] = True
        st = data[name]
        st.node_coords, st.node_coords_LAS, st.segment, st.is_global = X, XL, seg, glb
        st.mask = (seg == 0) | glb
        st.batch = torch.repeat_interleave(ar(B), n, output_size=total)
        data[name, "c2c", name].edge_index = (bonds + 1 + off[bb][:, None]).t().contiguous()
        data[name, "LAS", name].edge_index = (las + 1 + off[lb][:, None]).t().contiguous()
    data["compound", "LAS", "compound"].edge_index = (las + _offsets(Nc)[lb][:, None]).t().contiguous()
    data["compound_atom_edge_list"].x, data["compound_atom_edge_list"].batch = bonds + 1, bb
    data["LAS_edge_list"].x, data["LAS_edge_list"].batch = las + 1, lb
    # ---- pocket-ligand distance map, clamped (utils/utils.py:232-237): protein-major inside a complex
    npair = Pk * Nc
    pairb = torch.repeat_interleave(ar(B), npair)
    ploc = ar(pairb.shape[0]) - _offsets(npair)[pairb]
    pi = _offsets(Pk)[pairb] + torch.div(ploc, Nc[pairb], rounding_mode="floor")
    ci = _offsets(Nc)[pairb] + ploc % Nc[pairb]
    data.dis_map = (pocket[pi] - lig[ci]).pow(2).sum(-1).sqrt().clamp(max=interaction_threshold)
    data.node_xyz, data.node_xyz_whole, data.coords = pocket, prot, lig
    data.coords_center, data.coord_offset, data.pocket_idx = com, bias, pocket_idx
    data.pocket_residue_center = pocket_centre                 # FABind+ (model.py:176-183)
    data.pdb = lay["pdb"]
    return data


class DeviceFeeder:
    """Double-buffered feeder: while the model works on batch k, batch k+1 is packed on a host thread, copied on a side
    stream and assembled there; `__iter__` yields device-resident batches in order."""

    def __init__(self, sample_batches, device, plan_with=None, **build_kw):
        """plan_with: an IaBNet model -- every batch is yielded as (batch, plan) with plan = model.plan_stage1(batch) built on the side
        stream too (the forward's index tables, pair lists, layouts and input graphs: `model(batch, stage=1, train=..., plan=plan)`
        then makes no host round trip of its own)."""
        self.src, self.device, self.kw, self.plan_with = sample_batches, torch.device(device), build_kw, plan_with

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=2)

        def host():
            for samples in self.src:
                q.put(pack_samples(samples))
            q.put(None)
        threading.Thread(target=host, daemon=True).start()
        side = torch.cuda.Stream(self.device)
        nxt = q.get()
        while nxt is not None:
            with torch.cuda.stream(side):
                batch = build_batch(None, self.device, packed=nxt, **self.kw)
                plan = self.plan_with.plan_stage1(batch) if self.plan_with is not None else None
                ready = torch.cuda.Event()
                ready.record(side)
            nxt = q.get()                                       # overlap: the next pack is already in flight
            torch.cuda.current_stream(self.device).wait_event(ready)
            for st in list(batch._stores.values()) + [batch._glob]:
                for v in st.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield batch if plan is None else (batch, plan)
