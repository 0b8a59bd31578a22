"""The FABind+ LN-MLPs and attention primitives (reference FABind_plus/fabind/models/model_utils.py:10-98, 150-320):
parameters under the reference's names.  The stack runs their arithmetic through fabind_amd.plus.engine; the stand-alone
forwards with the reference's (dense) signatures go through fabind_amd/plus/dense.py and fabind_amd/dense.py, on the
same kernels."""
import torch.nn as nn
from torch.nn import Linear


class _LnMlp(nn.Module):
    def __init__(self, args, embedding_channels, out_channels, n, bias2):
        super().__init__()
        if not args.use_ln_mlp:
            raise NotImplementedError("only --use-ln-mlp (FABind+ production) is built")
        self.args = args
        self.layernorm = nn.LayerNorm(embedding_channels)
        self.linear1 = Linear(embedding_channels, int(n * embedding_channels))
        self.linear2 = Linear(int(n * embedding_channels), out_channels, bias=bias2)


class MLP(_LnMlp):
    """LN -> linear1 -> relu -> [dropout] -> linear2 (model_utils.py:10-30)."""

    def __init__(self, args, embedding_channels=256, out_channels=256, n=4):
        super().__init__(args, embedding_channels, out_channels, n, True)
        if args.dropout > 0:
            self.dropout = nn.Dropout(args.dropout)

    def forward(self, z):
        from .. import dense
        return dense.mlp(self, z, False)


class MLPwithLastAct(_LnMlp):
    """LN -> linear1 -> relu -> [dropout] -> linear2 -> relu -> [dropout] (model_utils.py:32-53)."""

    def __init__(self, args, embedding_channels=256, out_channels=256, n=4):
        super().__init__(args, embedding_channels, out_channels, n, True)
        if args.dropout > 0:
            self.dropout1 = nn.Dropout(args.dropout)
            self.dropout2 = nn.Dropout(args.dropout)

    def forward(self, z):
        from .. import dense
        return dense.mlp(self, z, True)


class MLPwoBias(_LnMlp):
    """LN -> linear1 -> relu -> [dropout] -> linear2 without bias (model_utils.py:55-74)."""

    def __init__(self, args, embedding_channels=256, out_channels=256, n=4):
        super().__init__(args, embedding_channels, out_channels, n, False)
        if args.dropout > 0:
            self.dropout = nn.Dropout(args.dropout)

    def forward(self, z):
        from .. import dense
        return dense.mlp(self, z, False)


class MLP4Confidence(nn.Module):
    """Confidence / ranking head MLP: [LN] -> linear1 -> relu -> [dropout] -> linear2 (model_utils.py:77-98)."""

    def __init__(self, args, embedding_channels=256, out_channels=256, n=4):
        super().__init__()
        self.args = args
        if args.confidence_use_ln_mlp:
            self.layernorm = nn.LayerNorm(embedding_channels)
        if args.confidence_dropout > 0:
            self.dropout = nn.Dropout(args.confidence_dropout)
        self.linear1 = Linear(embedding_channels, n * embedding_channels)
        self.linear2 = Linear(n * embedding_channels, out_channels)

    def forward(self, z):
        from .. import dense
        return dense.mlp(self, z, False)


class Attention(nn.Module):
    """Gated multi-head attention parameters (model_utils.py:150-270)."""

    def __init__(self, args, c_q, c_k, c_v, c_hidden, no_heads, gating=True, mha_permu=False):
        super().__init__()
        if args.rel_dis_pair_bias != "no":
            raise NotImplementedError("rel_dis_pair_bias add/mul is off in the FABind+ production flags; not built")
        self.args, self.c_hidden, self.no_heads, self.mha_permu = args, c_hidden, no_heads, mha_permu
        self.linear_q = Linear(c_q, c_hidden * no_heads, bias=False)
        self.linear_k = Linear(c_k, c_hidden * no_heads, bias=False)
        self.linear_v = Linear(c_v, c_hidden * no_heads, bias=False)
        self.linear_o = Linear(c_hidden * no_heads, c_q)
        self.linear_g = Linear(c_q, c_hidden * no_heads) if gating else None
        self.sigmoid = nn.Sigmoid()

    def forward(self, q_x, kv_x, biases=None, distance=None):
        """Reference signature (model_utils.py:109-147): [*, Q, Cq], [*, K, Ck], biases broadcastable to [*, heads, Q, K]."""
        from ... import dense
        return dense.attention(self, q_x, kv_x, biases)


class InteractionModule(nn.Module):
    """Hadamard pair embedding parameters (model_utils.py:273-320); opm=False, rm_layernorm only."""

    def __init__(self, node_hidden_dim, pair_hidden_dim, hidden_dim, opm=False, rm_layernorm=False):
        super().__init__()
        if opm or not rm_layernorm:
            raise NotImplementedError("only opm=False with --rm-layernorm (production) is built")
        self.hidden_dim, self.pair_hidden_dim, self.node_hidden_dim, self.opm = hidden_dim, pair_hidden_dim, node_hidden_dim, opm
        self.rm_layernorm = rm_layernorm
        self.linear_p = nn.Linear(node_hidden_dim, hidden_dim)
        self.linear_c = nn.Linear(node_hidden_dim, hidden_dim)
        self.linear_out = nn.Linear(hidden_dim, pair_hidden_dim)

    def forward(self, p_embed, c_embed, p_mask=None, c_mask=None):
        """Reference signature (model_utils.py:296-320) -> (pair embedding [*, P, C, pair_hidden], pair mask)."""
        from ... import dense
        return dense.interaction(self, p_embed, c_embed, p_mask, c_mask)
