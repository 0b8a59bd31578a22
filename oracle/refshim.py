"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-ins that let the *reference's own* `models/*.py` (under /root/reference, read-only,
never copied) import and run on CPU in THIS container so that golden vectors can be captured
from it (`oracle/make_golden.py`).  The reference depends on third-party wheels that are not
installed here:

  * torch_scatter 2.1.0  -> scatter_sum / scatter_add / scatter_mean / scatter_softmax / scatter_max
  * torch_geometric 2.4.0 -> utils.to_dense_batch (+ name-only to_dense_adj, data.Data/HeteroData)
  * torchmetrics, rdkit  -> name-only stubs (only needed so `utils/utils.py` imports)

The arithmetic of those libraries is restated from their documented behaviour; call sites in the
reference: egnn.py:221,444,777  att_model.py:43  model.py:138-144,344-346.

Nothing here travels to the GPU box in a way that matters: /root/reference does not exist there and
`load_reference()` raises if it is missing.
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------------------------
# torch_scatter stand-ins (documented semantics: reduce `src` into `out[index]` along `dim`)
# ----------------------------------------------------------------------------------------------
def _broadcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode="floor")
    return out


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    index_b = _broadcast(index, src, dim)
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    else:
        size[dim] = int(index.max()) + 1
    res = torch.full(size, float("-inf"), dtype=src.dtype, device=src.device)
    res = res.scatter_reduce(dim, index_b, src, reduce="amax", include_self=True)
    return res, None


def scatter_softmax(src, index, dim=-1, dim_size=None):
    index_b = _broadcast(index, src, dim)
    mx, _ = scatter_max(src, index, dim, dim_size=dim_size)
    rec = src - mx.gather(dim, index_b)
    e = rec.exp()
    s = scatter_sum(e, index, dim, dim_size=dim_size)
    return e / s.gather(dim, index_b)


# ----------------------------------------------------------------------------------------------
# torch_geometric.utils.to_dense_batch (batch vector sorted; pads ragged rows to the max count)
# ----------------------------------------------------------------------------------------------
def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    if batch is None:
        mask = torch.ones(1, x.size(0), dtype=torch.bool, device=x.device)
        return x.unsqueeze(0), mask
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() > 0 else 1
    num_nodes = torch.zeros(batch_size, dtype=torch.long, device=x.device).scatter_add_(
        0, batch, torch.ones_like(batch))
    cum = torch.cat([num_nodes.new_zeros(1), num_nodes.cumsum(0)])
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    tmp = torch.arange(batch.size(0), device=x.device) - cum[batch]
    idx = tmp + batch * max_num_nodes
    size = [batch_size * max_num_nodes] + list(x.size())[1:]
    out = x.new_full(size, fill_value)
    out[idx] = x
    out = out.view([batch_size, max_num_nodes] + list(x.size())[1:])
    mask = torch.zeros(batch_size * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[idx] = 1
    mask = mask.view(batch_size, max_num_nodes)
    return out, mask


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    """torch_geometric.utils.to_dense_adj for ONE graph (batch=None): [1, N, N] edge counts, N = max index + 1.  Imported
    by egnn.py:14 (never called on the model path) and called by utils/post_optim_utils.py:40."""
    if batch is not None or edge_attr is not None:
        raise NotImplementedError
    n = int(edge_index.max()) + 1 if max_num_nodes is None else max_num_nodes
    adj = torch.zeros(1, n, n)
    adj[0].index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.shape[1]), accumulate=True)
    return adj


# ----------------------------------------------------------------------------------------------
# A dict-like stand-in for a collated PyG HeteroData batch (field contract: SURVEY.md A.10)
# ----------------------------------------------------------------------------------------------
class Store(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class FakeHeteroData:
    """data['compound'].batch, data['complex','c2c','complex'].edge_index, data.coords ..."""

    def __init__(self):
        object.__setattr__(self, "_stores", {})
        object.__setattr__(self, "_glob", Store())

    def __getitem__(self, key):
        st = self._stores
        if key not in st:
            st[key] = Store()
        return st[key]

    def __getattr__(self, k):
        try:
            return self._glob[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self._glob[k] = v

    def to(self, device):
        return self


def _pearson_corrcoef(a, b):
    a, b = a - a.mean(), b - b.mean()
    return (a * b).sum() / (a.pow(2).sum().sqrt() * b.pow(2).sum().sqrt())


def _mean_squared_error(a, b, squared=True):
    m = (a - b).pow(2).mean()
    return m if squared else m.sqrt()


def _install_stubs():
    ts = types.ModuleType("torch_scatter")
    for n, f in dict(scatter_sum=scatter_sum, scatter_add=scatter_add, scatter_mean=scatter_mean,
                     scatter_softmax=scatter_softmax, scatter_max=scatter_max).items():
        setattr(ts, n, f)
    sys.modules["torch_scatter"] = ts

    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.to_dense_batch = to_dense_batch
    tgu.to_dense_adj = to_dense_adj
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = object
    tgd.HeteroData = FakeHeteroData
    tgd.Dataset = object
    tg.utils, tg.data = tgu, tgd
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tgu, "torch_geometric.data": tgd})

    for name in ("torchmetrics", "rdkit", "rdkit.Chem", "rdkit.Chem.rdMolTransforms", "rdkit.Geometry", "mlflow",
                 "mlflow.sklearn"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.modules["rdkit.Chem"].rdMolTransforms = sys.modules["rdkit.Chem.rdMolTransforms"]
    sys.modules["rdkit"].Geometry = sys.modules["rdkit.Geometry"]
    sys.modules["rdkit.Geometry"].Point3D = object
    sys.modules["mlflow"].sklearn = sys.modules["mlflow.sklearn"]
    sys.modules["mlflow.sklearn"].autolog = lambda *a, **k: None

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    tm = sys.modules["torchmetrics"]
    tm.__getattr__ = lambda name: _Any  # type: ignore
    tm.functional = types.ModuleType("torchmetrics.functional")
    tm.functional.__getattr__ = lambda name: _Any  # type: ignore
    # the three torchmetrics 0.x functionals the evaluation path calls (utils/metrics.py:62-77), restated from their
    # published definitions: Pearson r = cov / (std std), RMSE (squared=False) / MSE, MAE
    tm.functional.pearson_corrcoef = _pearson_corrcoef
    tm.functional.mean_squared_error = _mean_squared_error
    tm.functional.mean_absolute_error = lambda a, b: (a - b).abs().mean()
    sys.modules["torchmetrics.functional"] = tm.functional


def load_reference(variant="FABind"):
    """Import the reference's `models.model` (FABind or FABind_plus) with the stand-ins above.

    Returns the dict of imported reference modules.  Only for this container."""
    root = os.path.join(REFERENCE_ROOT, variant, "fabind")
    if not os.path.isdir(root):
        raise RuntimeError("reference tree not present (expected on the build container only): " + root)
    _install_stubs()
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "utils"
              or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, root)
    try:
        mods = {}
        for name in ("models.model_utils", "models.cross_att", "models.egnn", "models.att_model",
                     "models.model", "utils.utils"):
            mods[name] = importlib.import_module(name)
    finally:
        sys.path.remove(root)
    return mods


def load_post_optim(variant="FABind"):
    """Import the reference's utils/post_optim_utils.py (torch-only arithmetic; rdkit is name-stubbed, file I/O unused)."""
    root = os.path.join(REFERENCE_ROOT, variant, "fabind")
    import torch._dynamo  # noqa: F401  (torch.optim pulls it in lazily; import it before the name-only stubs exist)
    _install_stubs()
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_post_optim_utils", os.path.join(root, "utils", "post_optim_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def production_args(**over):
    """The Namespace the reference's eval scripts rebuild (test_fabind.py:182)."""
    from argparse import Namespace
    a = dict(
        mode=5, n_iter=8, mean_layers=4, hidden_size=512, pocket_pred_hidden_size=128,
        pocket_pred_layers=1, pocket_pred_n_iter=1, refine="refine_coord", coordinate_scale=5.0,
        geometry_reg_step_size=0.001, rm_layernorm=True, add_attn_pair_bias=True,
        explicit_pair_embed=True, add_cross_attn_layer=True, norm_type="per_sample",
        random_n_iter=True, center_dist_threshold=4.0, stage_prob=0.25, distmap_pred="mlp",
        use_esm2_feat=True, esm2_concat_raw=False, inter_cutoff=10.0, intra_cutoff=8.0,
        pocket_radius=20.0, gs_tau=1.0, gs_hard=False, local_eval=False, train_pred_pocket_noise=0.0,
        compound_coords_init_mode="pocket_center_rdkit", ablation_no_attention=False,
        ablation_no_attention_with_cross_attn=False, keep_trig_attn=False, opm=False,
        rm_F_norm=False, fix_pocket=False, rm_LAS_constrained_optim=False,
    )
    a.update(over)
    return Namespace(**a)


def production_args_plus(**over):
    """FABind+ (FABind_plus/fabind/test_regression_fabind.py:42) on top of the v1 production flags;
    every field FABind_plus/fabind/utils/parsing.py defines that the model constructors read."""
    a = vars(production_args(mean_layers=5))
    a.update(dict(
        use_ln_mlp=True, mlp_hidden_scale=1, dropout=0.1, mha_heads=4, rel_dis_pair_bias="no",
        inter_additional_mlp=False, only_last_LAS=False, geom_reg_steps=1, permutation_invariant=True,
        use_for_radius_pred="ligand", dis_map_thres=15.0, pocket_radius_buffer=5.0, min_pocket_radius=20.0,
        force_fix_radius=False, use_clustering=False, dbscan_eps=9.0, dbscan_min_samples=2, choose_cluster_prob=0.5,
        stack_mlp=False, confidence_dropout=0.1, confidence_use_ln_mlp=False, confidence_mlp_hidden_scale=2,
        train_ligand_torsion_noise=False, train_pred_pocket_noise=0.0, infer_dropout=False, confidence_training=False,
        ranking_loss="logsigmoid", wandb=False, num_copies=1, keep_cls_2A=False, symmetric_rmsd=None))
    a.update(over)
    from argparse import Namespace
    return Namespace(**a)
