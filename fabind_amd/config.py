"""Process-wide numeric mode of the HIP path."""
import os

_PRECISION = {"mode": "bf16", "x3_backward": "bf16" if (os.environ.get("FABIND_X3_WGRAD", "bf16") == "bf16" and
                                                          os.environ.get("FABIND_X3_PAIRBIAS_BWD", "bf16") == "bf16") else "exact"}
MODES = ("fp32", "bf16", "bf16x3")
_PRECISION["split_sites"] = int(os.environ.get("FABIND_SPLIT_SITES", "3"))


def set_precision(mode):
    """'fp32': exact-fp32 MFMA everywhere (parity mode, 1e-4 A gate; 1/16 of the bf16 matrix rate).
    'bf16': bf16 MFMA operands with fp32 accumulation; coordinates, radial terms, softmax and all
    reductions stay fp32; edge-level intermediates are stored as bf16.
    'bf16x3': fp32 storage everywhere like 'fp32'; the FORWARD pass and the activation-gradient chain of the backward pass run every
    contraction as SPLIT bf16 on the bf16 matrix cores: an operand element x is hi + lo with hi = bf16(x), lo = bf16(x - hi) (16
    significand bits), a product term is three MFMAs (lo*hi + hi*lo + hi*hi) with fp32 accumulation -- fp32-grade outputs (the mode
    that meets the 1e-4 A gate) at 3/16 of the fp32 matrix time.  Three parts of the BACKWARD pass default to single bf16 roundings of
    their fp32 operands (`set_x3_backward("bf16")`, the default): the weight-gradient contractions, the fused edge backward (the bf16
    recompute kernel on bf16 copies of AB and the weights) and the pair-bias adjoint -- parameter gradients are then bf16-grade
    (whole-gradient l2 error ~2e-3, tests/test_gpu_headline.py), the outputs and losses are not affected.
    `set_x3_backward("exact")` runs the weight gradients as split contractions, the pair-bias adjoint in fp32 and -- round 5 -- a
    differentiable pass's intra-graph edge pipeline UNFUSED (engine.gcl_layer: fp32 edge tensors, split contractions in the forward, the
    input gradients and the weight gradients; there is no split form of the fused edge backward kernel): no bf16 rounding anywhere in the
    adjoint, at the cost of ~3 GB of edge tensors per layer and the unfused launches; bench.py reports both (`gate_mode`,
    `gate_mode_exact_bwd`)."""
    assert mode in MODES
    _PRECISION["mode"] = mode


def set_split_sites(level):
    """'bf16' mode only: which node-level GEMM sites run their FORWARD contraction in split precision (three bf16 MFMAs per product term on
    the fp32 activation and the fp32 master weight, like 'bf16x3') instead of on bf16-rounded operands.  Measured at the headline shape
    (tools/probes/precision_sites.py, precision_mixed.py; profiles/r05_precision_sites.txt): of the bf16 mode's 1.0e-4 A ligand-RMSD gap to
    the fp32 reference, the bf16 edge pipeline carries 3.5e-6 and the bf16 attention tiles 1.3e-6; the node-level GEMM operand roundings
    carry the rest, and three sites most of that -- the inter-edge attention's q | k | v projection (5.4e-5 alone), its coordinate-MLP
    projection cv = Wc v (5.7e-5) and the stack's input Linear (3.2e-5).
      0: none (rounds 1-4: every GEMM on bf16 operands): 1.00e-4 / 1.76e-4 / 6.26e-4 A at n_iter 1 / 2 / 8;
      1: those three sites (9 of the ~50 node-level GEMMs of a pass): 3.84e-5 / 6.71e-5 / 2.45e-4 A (q | k | v fully split);
      2 (round 5's default): + the protein-query attention block's output projection and both blocks' k | v projections (21 GEMMs of a pass), and of
         q | k | v only the V columns (the q and k columns do not carry the gap: tools/probes/precision_qkv_parts.py):
         3.07e-5 / 4.66e-5 / 1.90e-4 A -- the 1e-4 A gate met with margin for one and two passes; headline -2.4 %, forward only -5 %.
      (Measured on MI355X: profiles/r05_precision_sites.txt.  The remaining gap sits in the GEMMs whose activation operand is a HIDDEN
      layer -- node MLP, Transition --, which the bf16 mode stores as bf16: emulated with those in split precision too the loop reads
      5.8e-5 A at n_iter 8; 'bf16x3' is the mode that meets the gate there.)
      3 (round 6, default): + both Linears of every node MLP (egnn.py:89-109) and of both Transitions (model_utils.py:162-175), with the
         hidden layer kept in fp32 between them (no-grad passes: two split-precision launches instead of the one-kernel bf16 node chain;
         under autograd the bf16 roundings of the hidden activation / its derivative are saved for the unchanged backward), the
         ligand-query block's output projection, and the full model's embedding Linears around the two stacks (models/model.py `_lin`):
         1.37e-5 / 2.73e-5 / 7.76e-5 A at n_iter 1 / 2 / 8 -- the 1e-4 A gate met in the production loop at the headline shape; config 3 read
         literally 1.6e-5 A (level 2: 5.6e-5); full IaBNet at production size, n_iter 8: stage 1 9.1e-5 A (met), stage 2 -- the ligand moves
         17 A -- 1.3e-4 A (missed; 'bf16x3' is the mode for that loop).  Cost, same box (profiles/r06_split_sites.txt): stack step -6.5 %,
         config-3 step -4 %, n_iter 8 -12 % (the four MLP contractions of a layer cost three MFMAs per product term and fp32 operands).
    Backward passes are unchanged (bf16 operands; the gradients of a bf16-mode step are bf16-grade either way)."""
    assert level in (0, 1, 2, 3)
    _PRECISION["split_sites"] = int(level)


def split_sites():
    return _PRECISION["split_sites"] if _PRECISION["mode"] == "bf16" else 0


_PRECISION["x3_edge"] = os.environ.get("FABIND_X3_EDGE", "split")


def set_x3_edge(kind):
    """'bf16x3' mode: the fused intra-graph edge pipeline (MC_E_GCL edge / coord model, egnn.py:68-128) as
      'split' (default): the split-bf16 forward kernel on fp32 AB rows (fp32-grade node features: H to 4e-6 of its largest entry);
      'bf16'  (round 5, an option): the bf16 kernels of the 'bf16' mode on a bf16 copy of the per-node projections AB: forward 2.2 instead
               of 4.9 ms per launch -- gate mode 531 -> 592 complexes/s, the n_iter 8 loop 156 -> 202 -- and the COORDINATE / loss gates still
               met, with less margin: ligand RMSD 3.5e-6 / 7.0e-6 / 3.0e-5 A at n_iter 1 / 2 / 8 at the headline shape (split: 8e-7 / 9e-7 /
               2.9e-6), full IaBNet at production size 5.1e-5 / 2.0e-5 A (2.3e-6 / 2.6e-6); node features H to 3.4e-4 only (the messages
               are summed as bf16) -- profiles/r05_precision_sites.txt, tests/test_gpu_headline.py."""
    assert kind in ("bf16", "split")
    _PRECISION["x3_edge"] = kind


def x3_edge_bf16():
    return _PRECISION["x3_edge"] == "bf16"


# (round 6: `set_x3_attn("bf16")` -- the bf16 fused attention kernels inside the 'bf16x3' mode, +3.5 % -- is RETIRED: the full IaBNet's stage 2
#  read 9.35e-5 A with it, inside the 1e-4 A gate without margin; a switch whose on-state cannot be asserted at the gate does not ship.)


def set_x3_backward(kind):
    """'bf16' (default) / 'exact': see set_precision('bf16x3')."""
    assert kind in ("bf16", "exact")
    _PRECISION["x3_backward"] = kind


def x3_backward_bf16():
    return _PRECISION["x3_backward"] == "bf16"


def get_precision():
    return _PRECISION["mode"]


def fp32_storage():
    """True in the modes that keep every activation / weight in fp32 ('fp32' and 'bf16x3')."""
    return _PRECISION["mode"] != "bf16"
