"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's ligand post-optimisation
(FABind/fabind/utils/post_optim_utils.py:9-64): 1000 Adam steps (lr 0.1) on the predicted ligand coordinates against

    loss = sum_{(i,j) in LAS mask} | |x_i - x_j| - |r_i - r_j| |  +  2 * sum_{i,j} relu(1.22 - |x_i - x_j|)

(r = the RDKit reference conformer; the LAS mask is the dense adjacency of LAS_edge_index; without LAS edges the first
term runs over all pairs and the excluded-volume term is dropped).  The gradient is written out by hand (what the HIP
kernel evaluates) instead of going through autograd; pinned against the reference function itself in
tests/test_oracle_golden.py (fixtures produced by oracle/make_golden.py `post`).  Only tests / smoke / the bench's CPU
baseline may import this module."""
import numpy as np


def post_optimize_compound_coords(ref, pred, total_epoch=1000, LAS_edge_index=None, lr=0.1):
    """-> (x [N,3] float32, loss at the last epoch, RMSD to `ref` after the last step).  float32 arithmetic throughout."""
    f = np.float32
    ref, x = np.asarray(ref, f), np.array(pred, f)
    n = x.shape[0]
    dref = np.sqrt(((ref[:, None] - ref[None]) ** 2).sum(-1, dtype=f)).astype(f)
    if LAS_edge_index is not None:
        mask = np.zeros((n, n), bool)
        mask[np.asarray(LAS_edge_index[0]), np.asarray(LAS_edge_index[1])] = True
    else:
        mask = np.ones((n, n), bool)
    w = (mask.astype(f) + mask.T.astype(f))                 # d loss / d x_k collects the (k,j) and the (j,k) term
    m, v = np.zeros_like(x), np.zeros_like(x)
    b1, b2, eps = f(0.9), f(0.999), f(1e-8)
    loss = f(0)
    for t in range(1, total_epoch + 1):
        diff = x[:, None] - x[None]
        d = np.sqrt((diff * diff).sum(-1, dtype=f)).astype(f)
        safe = np.where(d > 0, d, f(1))
        u = diff / safe[..., None] * (d > 0)[..., None]      # cdist backward: zero where the distance is zero
        dev = d - dref
        g_pair = w * np.sign(dev).astype(f)
        loss = np.abs(dev)[mask].sum(dtype=f)
        if LAS_edge_index is not None:
            g_pair = g_pair - f(4) * ((d < f(1.22)) & (d > 0))
            loss = loss + f(2) * np.maximum(f(1.22) - d, 0).sum(dtype=f)
        g = (g_pair[..., None] * u).sum(1, dtype=f).astype(f)
        m = b1 * m + (f(1) - b1) * g
        v = b2 * v + (f(1) - b2) * g * g
        step = f(lr) / f(1 - 0.9 ** t)
        x = (x - step * m / (np.sqrt(v) / np.sqrt(f(1 - 0.999 ** t)) + eps)).astype(f)
    rmsd = float(np.sqrt(((ref - x) ** 2).sum(-1).mean()))
    return x, float(loss), rmsd
