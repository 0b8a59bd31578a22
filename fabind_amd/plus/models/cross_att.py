"""CrossAttentionModule / RowAttentionBlock of FABind+ (reference FABind_plus/fabind/models/cross_att.py:7-89): parameters
under the reference's names; the stack runs them through fabind_amd.plus.engine, the stand-alone forwards with the
reference's dense signatures through the adapters fabind_amd/plus/dense.py / fabind_amd/dense.py (same kernels)."""
import torch.nn as nn
from torch.nn import Linear

from .model_utils import Attention, InteractionModule, MLPwithLastAct


class RowAttentionBlock(nn.Module):
    inf = 1e9

    def __init__(self, args, node_hidden_dim, pair_hidden_dim, attention_hidden_dim=32, no_heads=4, dropout=0.1,
                 rm_layernorm=False, mha_permu=False):
        super().__init__()
        if not rm_layernorm:
            raise NotImplementedError("only --rm-layernorm (production) is built")
        if no_heads != 4 or attention_hidden_dim != 32:
            raise NotImplementedError("the attention kernels are built for 4 heads x 32 channels (production)")
        self.no_heads, self.attention_hidden_dim = no_heads, attention_hidden_dim
        self.pair_hidden_dim, self.node_hidden_dim, self.rm_layernorm = pair_hidden_dim, node_hidden_dim, rm_layernorm
        self.linear = Linear(pair_hidden_dim, no_heads)
        self.linear_g = Linear(pair_hidden_dim, no_heads)
        self.dropout = nn.Dropout(args.dropout)
        self.mha = Attention(args, node_hidden_dim, node_hidden_dim, node_hidden_dim, attention_hidden_dim, no_heads,
                             mha_permu=mha_permu)

    def forward(self, node_embed_i, node_embed_j, pair_embed, pair_mask, node_mask_i, distance=None):
        """Reference signature (cross_att.py:72-89): [*, I, C], [*, J, C], [*, I, J, C_pair], masks -> [*, I, C]."""
        from ... import dense
        return dense.row_attention(self, node_embed_i, node_embed_j, pair_embed, pair_mask, node_mask_i)


class CrossAttentionModule(nn.Module):
    def __init__(self, args, node_hidden_dim, pair_hidden_dim, rm_layernorm=False, keep_trig_attn=False, dist_hidden_dim=32,
                 normalize_coord=None):
        super().__init__()
        if keep_trig_attn:
            raise NotImplementedError("--keep-trig-attn is off in every shipped command; not built")
        self.pair_hidden_dim, self.keep_trig_attn = pair_hidden_dim, keep_trig_attn
        self.p_attention_block = RowAttentionBlock(args, node_hidden_dim, pair_hidden_dim, no_heads=args.mha_heads,
                                                   rm_layernorm=rm_layernorm, mha_permu=True)
        self.c_attention_block = RowAttentionBlock(args, node_hidden_dim, pair_hidden_dim, no_heads=args.mha_heads,
                                                   rm_layernorm=rm_layernorm, mha_permu=False)
        n = args.mlp_hidden_scale
        self.p_transition = MLPwithLastAct(args, embedding_channels=node_hidden_dim, n=n, out_channels=node_hidden_dim)
        self.c_transition = MLPwithLastAct(args, embedding_channels=node_hidden_dim, n=n, out_channels=node_hidden_dim)
        self.pair_transition = MLPwithLastAct(args, embedding_channels=pair_hidden_dim, n=n, out_channels=pair_hidden_dim)
        self.inter_layer = InteractionModule(node_hidden_dim, pair_hidden_dim, 32, opm=False, rm_layernorm=rm_layernorm)

    def forward(self, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask, c_c_dist_embed=None,
                p_p_dist_embed=None, distance=None):
        """Reference signature (cross_att.py:20-47) -> (p', c', updated pair embedding)."""
        from .. import dense
        return dense.cross_attention(self, p_embed_batched, p_mask, c_embed_batched, c_mask, pair_embed, pair_mask)
