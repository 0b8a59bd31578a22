"""CrossAttentionModule / RowAttentionBlock (reference cross_att.py:7-134): parameters with the reference's names; the stack runs
their arithmetic through fabind_amd.engine, the stand-alone forwards with the reference's dense signatures go through the
adapter fabind_amd/dense.py (same kernels)."""
import torch.nn as nn
from torch.nn import LayerNorm, Linear

from .model_utils import Attention, InteractionModule, Transition


class RowAttentionBlock(nn.Module):
    inf = 1e9

    def __init__(self, node_hidden_dim, pair_hidden_dim, attention_hidden_dim=32, no_heads=4, dropout=0.1,
                 rm_layernorm=False):
        super().__init__()
        self.no_heads, self.attention_hidden_dim = no_heads, attention_hidden_dim
        self.pair_hidden_dim, self.node_hidden_dim = pair_hidden_dim, node_hidden_dim
        self.rm_layernorm = rm_layernorm
        if not rm_layernorm:
            self.layernorm_node_i = LayerNorm(node_hidden_dim)
            self.layernorm_node_j = LayerNorm(node_hidden_dim)
            self.layernorm_pair = LayerNorm(pair_hidden_dim)
        self.linear = Linear(pair_hidden_dim, no_heads)
        self.linear_g = Linear(pair_hidden_dim, no_heads)
        self.dropout = nn.Dropout(dropout)
        self.mha = Attention(node_hidden_dim, node_hidden_dim, node_hidden_dim, attention_hidden_dim, no_heads)

    def forward(self, node_embed_i, node_embed_j, pair_embed, pair_mask, node_mask_i):
        """Reference signature (cross_att.py:118-134): [*, I, C], [*, J, C], [*, I, J, C_pair], masks -> [*, I, C]."""
        from .. import dense
        return dense.row_attention(self, node_embed_i, node_embed_j, pair_embed, pair_mask, node_mask_i)


class CrossAttentionModule(nn.Module):
    def __init__(self, node_hidden_dim, pair_hidden_dim, rm_layernorm=False, keep_trig_attn=False, dist_hidden_dim=32,
                 normalize_coord=None):
        super().__init__()
        if keep_trig_attn:
            raise NotImplementedError("--keep-trig-attn (triangle attention) is off in every shipped command; not built")
        self.pair_hidden_dim, self.keep_trig_attn = pair_hidden_dim, keep_trig_attn
        self.p_attention_block = RowAttentionBlock(node_hidden_dim, pair_hidden_dim, rm_layernorm=rm_layernorm)
        self.c_attention_block = RowAttentionBlock(node_hidden_dim, pair_hidden_dim, rm_layernorm=rm_layernorm)
        self.p_transition = Transition(node_hidden_dim, 2, rm_layernorm=rm_layernorm)
        self.c_transition = Transition(node_hidden_dim, 2, rm_layernorm=rm_layernorm)
        self.pair_transition = Transition(pair_hidden_dim, 2, rm_layernorm=rm_layernorm)
        self.inter_layer = InteractionModule(node_hidden_dim, pair_hidden_dim, 32, opm=False, rm_layernorm=rm_layernorm)

    def forward(self, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask, c_c_dist_embed=None,
                p_p_dist_embed=None):
        """Reference signature (cross_att.py:24-54) -> (p', c', pair') as zero-padded dense tensors."""
        from .. import dense
        return dense.cross_attention(self, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask)
