"""MC_E_GCL / MC_Att_L / MCAttEGNN parameter containers (reference egnn.py:20-466).

`FABindLayer` is the thin name BASELINE.json's north_star uses for one loop body of
MCAttEGNN.forward (gcl_i -> att_i -> LAS step, reference egnn.py:402-449)."""
import torch
import torch.nn as nn

from .. import engine
from .cross_att import CrossAttentionModule
from .model_utils import InteractionModule


class MC_E_GCL(nn.Module):
    def __init__(self, args, input_nf, output_nf, hidden_nf, n_channel, edges_in_d=0, act_fn=nn.SiLU(), residual=True,
                 attention=False, normalize=False, coords_agg='mean', tanh=False, dropout=0.1, coord_change_maximum=10):
        super().__init__()
        assert n_channel == 1 and edges_in_d == 0 and residual and not attention and not tanh and coords_agg == 'mean'
        self.args, self.residual, self.coords_agg = args, residual, coords_agg
        self.dropout = nn.Dropout(dropout)
        self.edge_mlp = nn.Sequential(nn.Linear(input_nf * 2 + n_channel ** 2 + edges_in_d, hidden_nf), act_fn,
                                      nn.Linear(hidden_nf, hidden_nf), act_fn)
        self.node_mlp = nn.Sequential(nn.Linear(hidden_nf + input_nf, hidden_nf), act_fn, nn.Linear(hidden_nf, output_nf))
        layer = nn.Linear(hidden_nf, n_channel, bias=False)
        torch.nn.init.xavier_uniform_(layer.weight, gain=0.001)
        self.coord_mlp = nn.Sequential(nn.Linear(hidden_nf, hidden_nf), act_fn, layer)
        self.coord_change_maximum = coord_change_maximum

    def forward(self, h, edge_index, coord, edge_attr=None, node_attr=None, batch_id=None):
        """Reference signature (egnn.py:130-144): h [N,H], edge_index [2,E] (any order), coord [N,1,3], batch_id [N]
        (complex-contiguous) -> (h', coord').  Runs the same kernels as the stack (engine.gcl_layer) on a CSR built
        from the given edge list."""
        assert edge_attr is None and node_attr is None and batch_id is not None
        if not h.is_cuda:
            raise RuntimeError("fabind_amd: MC_E_GCL runs on a HIP device only (no CPU fallback)")
        n = h.shape[0]
        g = engine.EdgeListGraph(edge_index, n)
        lay = engine.BatchOnlyLayout(batch_id)
        pd = self.dropout.p if self.training else 0.0
        h2, x2 = engine.gcl_layer(engine.gcl_params(self), h.float().contiguous(), coord.reshape(n, 3).float().contiguous(),
                                  lay, g, float(self.coord_change_maximum), pd)
        return h2, x2.reshape(coord.shape)


class MC_Att_L(nn.Module):
    def __init__(self, args, input_nf, output_nf, hidden_nf, n_channel, edges_in_d=0, act_fn=nn.SiLU(), dropout=0.1,
                 coord_change_maximum=10, opm=False, normalize_coord=None):
        super().__init__()
        assert n_channel == 1 and edges_in_d == 0
        self.args, self.hidden_nf = args, hidden_nf
        self.dropout = nn.Dropout(dropout)
        self.linear_q = nn.Linear(input_nf, hidden_nf)
        self.linear_kv = nn.Linear(input_nf + n_channel ** 2 + edges_in_d, hidden_nf * 2)
        layer = nn.Linear(hidden_nf, n_channel, bias=False)
        torch.nn.init.xavier_uniform_(layer.weight, gain=0.001)
        self.coord_mlp = nn.Sequential(nn.Linear(hidden_nf, hidden_nf), act_fn, layer)
        self.coord_change_maximum = coord_change_maximum
        if not (args.add_cross_attn_layer and args.explicit_pair_embed and args.add_attn_pair_bias):
            raise NotImplementedError("only the production flags --add-cross-attn-layer --explicit-pair-embed "
                                      "--add-attn-pair-bias are built")
        self.cross_attn_module = CrossAttentionModule(node_hidden_dim=input_nf, pair_hidden_dim=input_nf,
                                                      rm_layernorm=args.rm_layernorm, keep_trig_attn=args.keep_trig_attn,
                                                      dist_hidden_dim=input_nf, normalize_coord=normalize_coord)
        # constructed by the reference but bypassed when add_cross_attn_layer is on (egnn.py:180-182, 266-284):
        # kept so that state_dict keys match; never receives a gradient.
        self.inter_layer = InteractionModule(input_nf, output_nf, hidden_nf, opm=opm, rm_layernorm=args.rm_layernorm)
        self.attn_bias_proj = nn.Linear(hidden_nf, 1)

    def forward(self, h, edge_index, coord, edge_attr=None, segment_id=None, batch_id=None, reduced_tuple=None,
                pair_embed_batched=None, pair_mask=None, LAS_mask=None, p_p_dist_embed=None, c_c_dist_embed=None):
        """Reference signature (egnn.py:308-333): h [N,H], inter edges [2,E] (both directions), coord [N,1,3], the dense
        pair embedding [B,Pmax,Cmax,H] + mask -> (h', coord', attention weights [E]).  `reduced_tuple` is recomputed from the
        edge list.  The stack itself never takes this route (fabind_amd.engine.att_layer works on the factored pair embedding);
        this adapter packs the dense tensors into ragged lists and runs the same kernels (fabind_amd/dense.py)."""
        assert edge_attr is None
        from .. import dense
        return dense.att_layer(self, h, edge_index, coord, segment_id, batch_id, pair_embed_batched, pair_mask)


class FABindLayer(nn.Module):
    """One FABind layer = MC_E_GCL -> MC_Att_L -> LAS step (reference egnn.py:402-449); a view over the
    parameters of layer `i` of an MCAttEGNN."""

    def __init__(self, gnn, i):
        super().__init__()
        self.gcl, self.att, self.index = getattr(gnn, "gcl_%d" % i), getattr(gnn, "att_%d" % i), i

    def forward(self, h, x, ctx, return_attention=False):
        """(h [N,H], x [N,3] or [N,1,3]) -> (h', x') through layer `index`: intra-graph message passing (MC_E_GCL),
        cross attention + inter-graph attention (MC_Att_L), LAS geometry step -- reference egnn.py:402-449.
        `ctx` = `EfficientMCAttModel.context(...)` (batch layout, current graph, pair-embedding factors)."""
        shp = x.shape
        h2, x2, alpha = ctx.layer(self.index, h, x.reshape(-1, 3))
        out = (h2, x2.reshape(shp))
        return out + (alpha,) if return_attention else out


class MCAttEGNN(nn.Module):
    def __init__(self, args, in_node_nf, hidden_nf, out_node_nf, n_channel, in_edge_nf=0, act_fn=nn.SiLU(), n_layers=4,
                 residual=True, dropout=0.1, dense=False, normalize_coord=None, unnormalize_coord=None,
                 geometry_reg_step_size=0.001):
        super().__init__()
        assert not dense and in_edge_nf == 0
        if not args.rm_layernorm:
            raise NotImplementedError("only --rm-layernorm (production) is built")
        if args.fix_pocket or args.rm_LAS_constrained_optim or args.rm_F_norm or args.norm_type != 'per_sample':
            raise NotImplementedError("only the production flags (no fix_pocket, LAS on, per_sample radial norm) are built")
        self.args = args
        self.geometry_reg_step_size, self.geom_reg_steps = geometry_reg_step_size, 1
        self.hidden_nf, self.n_layers = hidden_nf, n_layers
        self.dropout = nn.Dropout(dropout)
        self.linear_in = nn.Linear(in_node_nf, hidden_nf)
        self.dense, self.normalize_coord, self.unnormalize_coord = dense, normalize_coord, unnormalize_coord
        self.linear_out = nn.Linear(hidden_nf, out_node_nf)
        for i in range(n_layers):
            self.add_module(f'gcl_{i}', MC_E_GCL(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=in_edge_nf,
                                                 act_fn=act_fn, residual=residual, dropout=dropout,
                                                 coord_change_maximum=normalize_coord(10)))
            self.add_module(f'att_{i}', MC_Att_L(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=0,
                                                 act_fn=act_fn, dropout=dropout,
                                                 coord_change_maximum=normalize_coord(10), opm=args.opm,
                                                 normalize_coord=normalize_coord))
        self.out_layer = MC_E_GCL(args, hidden_nf, hidden_nf, hidden_nf, n_channel, edges_in_d=in_edge_nf, act_fn=act_fn,
                                  residual=residual, coord_change_maximum=normalize_coord(10))

    def layers(self):
        return [FABindLayer(self, i) for i in range(self.n_layers)]

    def forward(self, h, x, *reference_args, ctx=None, **reference_kwargs):
        """linear_in -> n_layers x FABindLayer -> out_layer -> linear_out (egnn.py:392-466) -> (h_out, x_out).

        Two call forms.  `forward(h, x, ctx=EfficientMCAttModel.context(...))`: the stack's own route (CSR graphs and the
        factored pair embedding live in `ctx`; the dense [B,P,C,H] tensor is never built).  `forward(h, x, ctx_edges,
        att_edges, LAS_edge_list, batched_complex_coord_LAS, segment_id=, batch_id=, reduced_tuple=, pair_embed_batched=,
        pair_mask=, ...)`: the reference's positional signature, served by the dense adapter (fabind_amd/dense.py)."""
        if ctx is None:
            from .. import dense
            return dense.egnn_forward(self, h, x, *reference_args, **reference_kwargs)
        shp = x.shape
        h2, x2 = ctx.gnn(h, x.reshape(-1, 3))
        return h2, x2.reshape(shp)
