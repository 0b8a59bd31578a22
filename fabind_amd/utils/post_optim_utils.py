"""Ligand post-optimisation on the GPU (reference FABind/fabind/utils/post_optim_utils.py:9-64, driven per complex by
fabind_inference.py:285-328).

`post_optimize_compound_coords` keeps the reference's name, arguments and return triple for ONE ligand;
`post_optimize_compound_coords_batched` runs a whole batch of ligands in one kernel launch (csrc/post_optim.hip: one
work-group per ligand, all Adam iterations inside the kernel).  HIP device tensors only -- there is no CPU fallback.

Numerics: the objective is non-smooth (|.| and relu) and Adam moves each coordinate by ~lr = 0.1 A per step, so the
iteration is chaotic: the reference's own result changes by ~0.1 A between its two `torch.cdist` code paths (direct
for <= 25 atoms, matmul-based above).  The kernel matches the reference step for step to float rounding (tests pin a short
horizon at 1e-5 A on the direct-cdist path) and reaches the same loss / RMSD level after the full 1000 steps."""
import torch

from .. import _lib
from .._lib import check, ptr, stream


def compute_RMSD(a, b):
    return torch.sqrt((((a - b) ** 2).sum(axis=-1)).mean())


def _neighbour_lists(LAS_edge_index, n_atoms, device):
    """CSR over atoms of the symmetrised, de-duplicated directed edge set: for every distinct (i, j) the list of i holds j
    and the list of j holds i (the dense boolean mask of the reference ignores duplicate edges)."""
    e = torch.unique(LAS_edge_index.to(device=device, dtype=torch.int64), dim=1)
    owner = torch.cat([e[0], e[1]])
    other = torch.cat([e[1], e[0]])
    order = torch.argsort(owner, stable=True)
    ptr_ = torch.zeros(n_atoms + 1, dtype=torch.int32, device=device)
    ptr_[1:] = torch.cumsum(torch.bincount(owner, minlength=n_atoms), 0).to(torch.int32)
    return ptr_, other[order].to(torch.int32).contiguous()


def post_optimize_compound_coords_batched(reference_compound_coords, predict_compound_coords, compound_batch,
                                          total_epoch=1000, LAS_edge_index=None, lr=0.1):
    """All ligands of a batch at once.  reference / predict: [sum Nc, 3]; compound_batch: sorted ligand id per atom;
    LAS_edge_index: [2, E] GLOBAL atom ids (edges never cross ligands) or None.
    -> (x [sum Nc, 3], loss [L] at the last epoch, rmsd [L] to the reference conformer after the last step)."""
    if not predict_compound_coords.is_cuda:
        raise RuntimeError("fabind_amd: post-optimisation runs on a HIP device only (no CPU fallback); got %s"
                           % predict_compound_coords.device)
    dev = predict_compound_coords.device
    x0 = predict_compound_coords.detach().to(torch.float32).contiguous()
    ref = reference_compound_coords.detach().to(device=dev, dtype=torch.float32).contiguous()
    n_atoms = x0.shape[0]
    cnt = torch.bincount(compound_batch.to(dev))
    L = cnt.shape[0]
    off = torch.zeros(L + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    max_atoms = int(cnt.max().item())
    if LAS_edge_index is not None:
        nptr, nidx = _neighbour_lists(LAS_edge_index, n_atoms, dev)
    else:
        nptr = nidx = None
    x = torch.empty_like(x0)
    loss = torch.empty(L, dtype=torch.float32, device=dev)
    rmsd = torch.empty(L, dtype=torch.float32, device=dev)
    check(_lib.load().fabind_post_optimize(ptr(x0), ptr(ref), ptr(off), ptr(nptr), ptr(nidx), L, max_atoms,
                                           0 if LAS_edge_index is not None else 1, int(total_epoch), float(lr), ptr(x),
                                           ptr(loss), ptr(rmsd), stream()), "fabind_post_optimize")
    return x, loss, rmsd


def post_optimize_compound_coords(reference_compound_coords, predict_compound_coords, total_epoch=1000, LAS_edge_index=None,
                                  mode=0):
    """The reference's per-ligand entry point: -> (x [Nc, 3], loss of the last epoch (float), RMSD (float)).  `mode` only
    selects the reference's unused interaction term (post_optim_utils.py:15-23: computed, never added to the loss)."""
    if mode not in (0, 1, 2):
        raise NotImplementedError()
    batch = torch.zeros(predict_compound_coords.shape[0], dtype=torch.int64, device=predict_compound_coords.device)
    x, loss, rmsd = post_optimize_compound_coords_batched(reference_compound_coords, predict_compound_coords, batch, total_epoch,
                                                          LAS_edge_index)
    return x, float(loss[0].item()), float(rmsd[0].item())
