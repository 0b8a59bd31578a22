"""EfficientMCAttModel of FABind+ (reference FABind_plus/fabind/models/att_model.py:131-223): the refinement loop over
the layer stack, returning (X, H, pair_embed [B, P, C, H])."""
import random

import torch.nn as nn

from ...models.att_model import ComplexGraph
from .. import engine
from .egnn import MCAttEGNN
from .model_utils import InteractionModule


class EfficientMCAttModel(nn.Module):
    def __init__(self, args, embed_size, hidden_size, n_channel, n_edge_feats=0, n_layers=5, dropout=0.1, n_iter=5,
                 dense=False, inter_cutoff=10, intra_cutoff=8, normalize_coord=None, unnormalize_coord=None):
        super().__init__()
        self.n_iter, self.args, self.random_n_iter = n_iter, args, args.random_n_iter
        if args.ablation_no_attention or args.ablation_no_attention_with_cross_attn:
            raise NotImplementedError("ablation stacks are not built")
        if args.refine != 'refine_coord' or not args.explicit_pair_embed:
            raise NotImplementedError("only refine='refine_coord' with --explicit-pair-embed (production) is built")
        self.gnn = MCAttEGNN(args, embed_size, hidden_size, hidden_size, n_channel, n_edge_feats, n_layers=n_layers,
                             residual=True, dropout=dropout, dense=dense, normalize_coord=normalize_coord,
                             unnormalize_coord=unnormalize_coord, geometry_reg_step_size=args.geometry_reg_step_size)
        self.extract_edges = ComplexGraph(args, inter_cutoff=inter_cutoff, intra_cutoff=intra_cutoff,
                                          normalize_coord=normalize_coord, unnormalize_coord=unnormalize_coord)
        self.inter_layer = InteractionModule(hidden_size, hidden_size, hidden_size, rm_layernorm=args.rm_layernorm)

    def forward(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                batched_complex_coord_LAS, LAS_mask=None, pair="dense"):
        """Same contract as the reference: mutates X in place, returns (X, H_out, pair_embed_batched).
        pair="ragged" / "none" skip the padded [B, P, C, H] copy (see engine.stack_forward)."""
        iter_i = random.randint(1, self.n_iter) if (self.training and self.random_n_iter) else self.n_iter
        return engine.stack_forward(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                                    batched_complex_coord_LAS, iter_i, pair=pair)
