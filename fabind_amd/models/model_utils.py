"""Parameter containers with the reference's names (FABind/fabind/models/model_utils.py:41-223).

The arithmetic of these blocks is executed by fused HIP kernels orchestrated in fabind_amd/engine.py;
the classes here own the parameters (so `state_dict` keys and initialisation match the reference)
and expose the stand-alone forwards that are cheap to provide on the kernels."""
import torch
import torch.nn as nn
from torch.nn import LayerNorm, Linear

from .. import kernels as K
from .. import ops


def _require_rm_layernorm(flag, who):
    if not flag:
        raise NotImplementedError(
            "%s: the HIP path implements the production configuration --rm-layernorm (reference test_fabind.py:182); "
            "LayerNorm'ed pair/node blocks are not built" % who)


class Attention(nn.Module):
    """Gated multi-head attention parameters (reference model_utils.py:41-159): q/k/v without bias,
    output projection and sigmoid gate with bias; c_hidden is the per-head width."""

    def __init__(self, c_q, c_k, c_v, c_hidden, no_heads, gating=True):
        super().__init__()
        self.c_q, self.c_k, self.c_v, self.c_hidden, self.no_heads, self.gating = c_q, c_k, c_v, c_hidden, no_heads, gating
        self.linear_q = Linear(c_q, c_hidden * no_heads, bias=False)
        self.linear_k = Linear(c_k, c_hidden * no_heads, bias=False)
        self.linear_v = Linear(c_v, c_hidden * no_heads, bias=False)
        self.linear_o = Linear(c_hidden * no_heads, c_q)
        self.linear_g = Linear(c_q, c_hidden * no_heads) if gating else None
        self.sigmoid = nn.Sigmoid()

    def forward(self, q_x, kv_x, biases=None):
        """Reference signature (model_utils.py:136-159): [*, Q, C_q], [*, K, C_k], biases broadcastable to [*, H, Q, K]."""
        from .. import dense
        return dense.attention(self, q_x, kv_x, biases)


class Transition(nn.Module):
    """x -> W2 relu(W1 x) (reference model_utils.py:162-175)."""

    def __init__(self, hidden_dim=128, n=4, rm_layernorm=False):
        super().__init__()
        self.rm_layernorm = rm_layernorm
        if not rm_layernorm:
            self.layernorm = LayerNorm(hidden_dim)
        self.linear_1 = Linear(hidden_dim, n * hidden_dim)
        self.linear_2 = Linear(n * hidden_dim, hidden_dim)

    def forward(self, x):
        _require_rm_layernorm(self.rm_layernorm, "Transition")
        shp = x.shape
        wd = ops.mm_dtype()
        t = ops.linear(x.reshape(-1, shp[-1]).float().contiguous(), self.linear_1.weight.to(wd), self.linear_1.bias,
                       act_epi=K.ACT_RELU)
        return ops.linear(t, self.linear_2.weight.to(wd), self.linear_2.bias).reshape(shp)


class InteractionModule(nn.Module):
    """Hadamard pair embedding W_o((W_p p_i) * (W_c c_j)) (reference model_utils.py:177-223, opm=False)."""

    def __init__(self, node_hidden_dim, pair_hidden_dim, hidden_dim, opm=False, rm_layernorm=False):
        super().__init__()
        if opm:
            raise NotImplementedError("opm=True is off in every shipped configuration and is not built")
        self.hidden_dim, self.pair_hidden_dim, self.node_hidden_dim, self.opm = hidden_dim, pair_hidden_dim, node_hidden_dim, opm
        self.rm_layernorm = rm_layernorm
        if not rm_layernorm:
            self.layer_norm_p = nn.LayerNorm(node_hidden_dim)
            self.layer_norm_c = nn.LayerNorm(node_hidden_dim)
        self.linear_p = nn.Linear(node_hidden_dim, hidden_dim)
        self.linear_c = nn.Linear(node_hidden_dim, hidden_dim)
        self.linear_out = nn.Linear(hidden_dim, pair_hidden_dim)

    def forward(self, p_embed, c_embed, p_mask=None, c_mask=None):
        """Reference signature (model_utils.py:200-223) -> (inter_embed [*, P, C, pair_hidden], inter_mask [*, P, C]).  The stack
        never materialises this tensor (fabind_amd/engine.py); this stand-alone form builds it because the API returns it."""
        from .. import dense
        return dense.interaction(self, p_embed, c_embed, p_mask, c_mask)
