"""MC_E_GCL / MC_Att_L / MCAttEGNN of FABind+ (reference FABind_plus/fabind/models/egnn.py:20-433): parameters under the
reference's names; the stack runs them through fabind_amd.plus.engine.stack_forward, the stand-alone forwards with the
reference's signatures through fabind_amd/plus/dense.py (same layer functions, same kernels)."""
import torch
import torch.nn as nn

from .cross_att import CrossAttentionModule
from .model_utils import InteractionModule, MLPwithLastAct, MLPwoBias


class MC_E_GCL(nn.Module):
    def __init__(self, args, input_nf, output_nf, hidden_nf, n_channel, edges_in_d=0, act_fn=nn.SiLU(), residual=True,
                 attention=False, normalize=False, coords_agg='mean', tanh=False, dropout=0.1, coord_change_maximum=10):
        super().__init__()
        assert n_channel == 1 and edges_in_d == 0 and residual and not attention and not tanh and coords_agg == 'mean'
        self.args, self.residual, self.coords_agg = args, residual, coords_agg
        n = args.mlp_hidden_scale
        self.edge_mlp = MLPwithLastAct(args, embedding_channels=input_nf * 2 + n_channel ** 2 + edges_in_d, n=n,
                                       out_channels=hidden_nf)
        self.node_mlp = MLPwithLastAct(args, embedding_channels=hidden_nf + input_nf, n=n, out_channels=output_nf)
        self.coord_mlp = MLPwoBias(args, embedding_channels=hidden_nf, n=n, out_channels=n_channel)
        torch.nn.init.xavier_uniform_(self.coord_mlp.linear2.weight, gain=0.001)
        self.coord_change_maximum = coord_change_maximum

    def forward(self, h, edge_index, coord, edge_attr=None, node_attr=None, batch_id=None):
        """Reference signature (egnn.py:100-118) -> (h', coord')."""
        assert edge_attr is None and node_attr is None and batch_id is not None
        from .. import dense
        return dense.gcl_forward(self, h, edge_index, coord, batch_id)


class MC_Att_L(nn.Module):
    def __init__(self, args, input_nf, output_nf, hidden_nf, n_channel, edges_in_d=0, act_fn=nn.SiLU(), dropout=0.1,
                 coord_change_maximum=10, opm=False, normalize_coord=None):
        super().__init__()
        assert n_channel == 1 and edges_in_d == 0
        if args.inter_additional_mlp:
            raise NotImplementedError("--inter-additional-mlp is off in the FABind+ production flags; not built")
        if not (args.add_cross_attn_layer and args.explicit_pair_embed and args.add_attn_pair_bias):
            raise NotImplementedError("only --add-cross-attn-layer --explicit-pair-embed --add-attn-pair-bias are built")
        self.args, self.hidden_nf = args, hidden_nf
        self.dropout = nn.Dropout(args.dropout)
        self.linear_q = nn.Linear(input_nf, hidden_nf)
        self.linear_kv = nn.Linear(input_nf + n_channel ** 2 + edges_in_d, hidden_nf * 2)
        self.coord_mlp = MLPwoBias(args, embedding_channels=hidden_nf, n=args.mlp_hidden_scale, out_channels=n_channel)
        torch.nn.init.xavier_uniform_(self.coord_mlp.linear2.weight, gain=0.001)
        self.coord_change_maximum = coord_change_maximum
        self.cross_attn_module = CrossAttentionModule(args, node_hidden_dim=input_nf, pair_hidden_dim=input_nf,
                                                      rm_layernorm=args.rm_layernorm, keep_trig_attn=args.keep_trig_attn,
                                                      dist_hidden_dim=input_nf, normalize_coord=normalize_coord)
        # constructed by the reference and bypassed when add_cross_attn_layer is on; kept for state_dict parity
        self.inter_layer = InteractionModule(input_nf, output_nf, hidden_nf, opm=opm, rm_layernorm=args.rm_layernorm)
        self.attn_bias_proj = nn.Linear(hidden_nf, 1)

    def forward(self, h, edge_index, coord, edge_attr=None, segment_id=None, batch_id=None, reduced_tuple=None,
                pair_embed_batched=None, pair_mask=None, LAS_mask=None, p_p_dist_embed=None, c_c_dist_embed=None):
        """Reference signature (egnn.py:277-300) -> (h', coord', attention weights [E], updated dense pair embedding);
        `reduced_tuple` is recomputed from the edge list."""
        assert edge_attr is None
        from .. import dense
        return dense.att_layer(self, h, edge_index, coord, segment_id, batch_id, pair_embed_batched, pair_mask)


class MCAttEGNN(nn.Module):
    def __init__(self, args, in_node_nf, hidden_nf, out_node_nf, n_channel, in_edge_nf=0, act_fn=nn.SiLU(), n_layers=4,
                 residual=True, dropout=0.1, dense=False, normalize_coord=None, unnormalize_coord=None,
                 geometry_reg_step_size=0.001):
        super().__init__()
        assert not dense and in_edge_nf == 0
        if not args.rm_layernorm or args.fix_pocket or args.rm_LAS_constrained_optim or args.only_last_LAS \
                or args.rm_F_norm or args.norm_type != 'per_sample':
            raise NotImplementedError("only the FABind+ production flags are built")
        self.args = args
        self.geometry_reg_step_size, self.geom_reg_steps = geometry_reg_step_size, 1
        self.hidden_nf, self.n_layers = hidden_nf, n_layers
        self.dropout = nn.Dropout(args.dropout)
        self.linear_in = nn.Linear(in_node_nf, hidden_nf)
        self.dense, self.normalize_coord, self.unnormalize_coord = dense, normalize_coord, unnormalize_coord
        self.linear_out = nn.Linear(hidden_nf, out_node_nf)
        for i in range(n_layers):
            self.add_module(f'gcl_{i}', MC_E_GCL(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=in_edge_nf,
                                                 act_fn=act_fn, residual=residual, dropout=dropout,
                                                 coord_change_maximum=normalize_coord(10)))
            self.add_module(f'att_{i}', MC_Att_L(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=0,
                                                 act_fn=act_fn, dropout=dropout, coord_change_maximum=normalize_coord(10),
                                                 opm=args.opm, normalize_coord=normalize_coord))
        self.out_layer = MC_E_GCL(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=in_edge_nf, act_fn=act_fn,
                                  residual=residual, coord_change_maximum=normalize_coord(10))

    def forward(self, h, x, *reference_args, **reference_kwargs):
        """The reference's positional signature (egnn.py:358-433): forward(h, x, ctx_edges, att_edges, LAS_edge_list,
        batched_complex_coord_LAS, segment_id=, batch_id=, reduced_tuple=, pair_embed_batched=, pair_mask=, ...) ->
        (h_out, x_out[, attention weights], pair embedding).  The model's own route is engine.stack_forward."""
        from .. import dense
        return dense.egnn_forward(self, h, x, *reference_args, **reference_kwargs)
