"""EfficientMCAttModel / ComplexGraph (reference att_model.py:29-246) on the HIP engine."""
import random

import torch
import torch.nn as nn

from .. import engine
from .egnn import MCAttEGNN
from .model_utils import InteractionModule


class ComplexGraph(nn.Module):
    """Holds the (normalised) cut-offs; edges are built by the CSR kernels (fabind_amd.engine.Graph)."""

    def __init__(self, args, inter_cutoff=10, intra_cutoff=8, normalize_coord=None, unnormalize_coord=None):
        super().__init__()
        self.args = args
        self.inter_cutoff = normalize_coord(inter_cutoff)
        self.intra_cutoff = normalize_coord(intra_cutoff)

    @torch.no_grad()
    def construct_edges(self, X, batch_id, segment_ids, is_global):
        """Reference signature (att_model.py:38).  Returns (ctx_edges [2,E_c'], inter_edges [2,E_i],
        (reduced_batchid, reduced_offsets)); ctx edges exclude the covalent bonds, like the reference,
        and are row-sorted (the reference concatenates three row-sorted runs instead)."""
        lay = engine.Layout(batch_id, segment_ids)
        x = X.reshape(-1, 3).float().contiguous()
        empty = torch.zeros(0, dtype=torch.int32, device=x.device)
        g = engine.Graph(lay, x, empty, empty, torch.zeros(lay.B + 1, dtype=torch.int32, device=x.device),
                         float(self.intra_cutoff), float(self.inter_cutoff))
        ctx = torch.stack([g.row_ctx.long(), g.col_ctx.long()])
        inter = torch.stack([g.row_int.long(), g.col_int.long()])
        half = inter[0] < inter[1]
        rb = batch_id[inter[0][half]]
        return ctx, inter, (rb, lay.node_off.long()[rb])

    def forward(self, X, batch_id, segment_id, is_global):
        return self.construct_edges(X, batch_id, segment_id, is_global)


class EfficientMCAttModel(nn.Module):
    def __init__(self, args, embed_size, hidden_size, n_channel, n_edge_feats=0, n_layers=5, dropout=0.1, n_iter=5,
                 dense=False, inter_cutoff=10, intra_cutoff=8, normalize_coord=None, unnormalize_coord=None):
        super().__init__()
        self.n_iter, self.args, self.random_n_iter = n_iter, args, args.random_n_iter
        if args.ablation_no_attention or args.ablation_no_attention_with_cross_attn:
            raise NotImplementedError("ablation stacks are dead under the production flags and are not built")
        if args.refine != 'refine_coord':
            raise NotImplementedError("only refine='refine_coord' (production) is built")
        self.gnn = MCAttEGNN(args, embed_size, hidden_size, hidden_size, n_channel, n_edge_feats, n_layers=n_layers,
                             residual=True, dropout=dropout, dense=dense, normalize_coord=normalize_coord,
                             unnormalize_coord=unnormalize_coord, geometry_reg_step_size=args.geometry_reg_step_size)
        self.extract_edges = ComplexGraph(args, inter_cutoff=inter_cutoff, intra_cutoff=intra_cutoff,
                                          normalize_coord=normalize_coord, unnormalize_coord=unnormalize_coord)
        if not args.explicit_pair_embed:
            raise NotImplementedError("only --explicit-pair-embed (production) is built")
        self.inter_layer = InteractionModule(hidden_size, hidden_size, hidden_size, rm_layernorm=args.rm_layernorm)
        self.dropout_p = dropout

    def context(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                batched_complex_coord_LAS, LAS_mask=None):
        """Per-batch state for `FABindLayer.forward` / `MCAttEGNN.forward` (same arguments as `forward`): batch
        layout, pair-embedding factors of the input H (att_model.py:198-206), and the ctx / inter graphs of X
        (att_model.py:209-214; call `ctx.rebuild_graph(x)` again after moving the coordinates)."""
        ctx = engine.StackContext(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                                  batched_complex_coord_LAS)
        ctx.rebuild_graph(X.reshape(-1, 3))
        return ctx

    def forward(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index, LAS_edge_index,
                batched_complex_coord_LAS, LAS_mask=None):
        """Same contract as the reference (att_model.py:170): X [N,1,3] is updated in place for the
        `mask`ed nodes and returned together with the last iteration's H."""
        if self.training and self.random_n_iter:
            iter_i = random.randint(1, self.n_iter)
        else:
            iter_i = self.n_iter
        return engine.stack_forward(self, X, H, batch_id, segment_id, mask, is_global, compound_edge_index,
                                    LAS_edge_index, batched_complex_coord_LAS, iter_i)
