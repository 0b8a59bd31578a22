"""ctypes binding of libfabind_hip.so (the C ABI declared in include/fabind_hip.h).

The library is the product: there is NO fallback.  If it cannot be loaded, or a tensor handed to a
kernel wrapper is not on a HIP device, the call raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FABIND_LIB") or os.path.join(_HERE, "libfabind_hip.so")      # FABIND_LIB: an A/B build (tools/probes)

ABI_VERSION = 18         # FABIND_ABI_VERSION of include/fabind_hip.h this binding mirrors
DT_F32, DT_BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_SIGMOID, ACT_STORED_DERIV = 0, 1, 2, 3, 4

_vp, _i, _f, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long


class GemmArgs(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("A", "A2", "W", "C", "bias", "R", "r_index", "dotvec", "dot_out", "aux", "groups", "C2")] + \
               [(n, _i) for n in ("M", "N", "K", "K1", "lda", "lda2", "ldw", "ldc", "ldr", "ldaux", "dot_ld",
                                  "a_dtype", "w_dtype", "c_dtype", "aux_dtype", "act_pro", "act_epi", "dact_epi",
                                  "accumulate", "store_preact", "n_groups", "max_m", "max_n", "groups_ext", "epi_fast", "k_splits")] + \
               [("alpha", _f), ("p_drop", _f), ("drop_seed", ctypes.c_uint)] + \
               [(n, _vp) for n in ("row_mu", "row_rs", "col_c", "C16")] + [("ldc16", _i), ("split3", _i), ("r_dtype", _i), ("c2_bf16", _i)]


class EdgeBwdArgs(ctypes.Structure):
    """Mirror of FabindEdgeBwdArgs (include/fabind_hip.h)."""
    _fields_ = [(n, _vp) for n in ("AB", "row", "col", "rhohat", "w_r", "W2p", "Wcp", "W2Tp", "WcTp", "b2", "bc", "w3", "ds",
                                  "dagg", "S1", "Mm", "dT", "dP2", "dP1", "drh", "dABrow", "part", "dbg", "bnd", "d2scratch")] + \
               [(n, _i) for n in ("ldab", "lddagg", "lddab", "E")] + [("p_drop", _f), ("seed", ctypes.c_uint), ("xcd_aware", _i), ("lddab16", _i), ("dAB16", _vp),
                                                                  ("d2f", _vp), ("z3f", _vp)]


class PairUpdateArgs(ctypes.Structure):
    """Mirror of FabindPairUpdateArgs (include/fabind_hip.h)."""
    _fields_ = [(n, _vp) for n in ("T", "p_node", "c_node", "z_in", "z_out", "Wop", "bo", "ln_w", "ln_b", "W1p", "b1", "W2p", "b2",
                                  "Wbp", "bb", "bias_out")] + \
               [(n, _i) for n in ("ldt", "b_off", "n_pairs")] + [("eps", _f), ("p_drop", _f), ("seed", ctypes.c_uint)]


class TnJob(ctypes.Structure):
    """Mirror of FabindTnJob (include/fabind_hip.h): one queued weight-gradient contraction of fabind_gemm_tn_multi."""
    _fields_ = [(n, _vp) for n in ("Y", "X", "C_part", "out", "out_tail")] + \
               [(n, _i) for n in ("ldy", "ldx", "M", "N", "E", "splits", "e_per", "n_tiles", "with_colsum", "out_dt", "ldo",
                                  "wg0", "n_wg", "blk0", "n_blk", "pad_")]


class AttnFusedBwdArgs(ctypes.Structure):
    """Mirror of FabindAttnFusedBwdArgs (include/fabind_hip.h)."""
    _fields_ = [(n, _vp) for n in ("qg", "kv", "a0", "bo", "boT", "toff", "koff", "bconst", "desc", "desc_p", "out", "lse", "dout",
                                  "dqg", "dkv", "dO", "Dv", "da0", "acat", "colpart", "part")] + \
               [(n, _i) for n in ("ldq", "ldkv", "lda0", "ldda0", "ldacat", "kcol0", "ldcolpart", "kp")] + [("scale", _f)] + \
               [(n, _i) for n in ("nsplit", "B", "part_rows")]


# name -> argtypes (every function returns int and takes the stream last)
SIGNATURES = {
    "fabind_gemm": [ctypes.POINTER(GemmArgs), _vp],
    "fabind_gemm_tn_multi": [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp],
    "fabind_gemm_tn": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "fabind_transpose_act": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fabind_colsum": [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "fabind_edges_count": [_vp, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp],
    "fabind_edges_fill": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_exclusive_scan": [_vp, _vp, _i, _vp],
    "fabind_inter_meta": [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_edge_geom": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "fabind_gcl_pre": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "fabind_gcl_edge_fused": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, ctypes.c_uint, _vp, _vp, _vp],
    "fabind_gcl_edge_fused_train": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, ctypes.c_uint, _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_gcl_edge_fused_x3": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, ctypes.c_uint, _vp, _vp],
    "fabind_gcl_edge_fused_x3_train": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, ctypes.c_uint, _vp, _vp, _vp, _vp, _vp],
    "fabind_gcl_edge_fused_bwd": [ctypes.POINTER(EdgeBwdArgs), _i, _i, _vp],
    "fabind_pair_update_fused": [ctypes.POINTER(PairUpdateArgs), _i, _vp],
    "fabind_gcl_edge_fused_bwd_set_tile": [_i],
    "fabind_gcl_edge_fused_bwd_tile": [],
    "fabind_gcl_edge_fused_bwd_set_variant": [_i],
    "fabind_gcl_edge_fused_set_variant": [_i],
    "fabind_gcl_edge_fused_variant": [],
    "fabind_gcl_edge_fused_bwd_variant": [],
    "fabind_gcl_edge_fused_bwd_variant_for": [_i],
    "fabind_row_stats": [_vp, _i, _i, _f, _i, _i, _vp, _vp, _vp],
    "fabind_layernorm_rows": [_vp, _i, _i, _vp, _vp, _f, _i, _i, _vp, _i, _i, _i, _vp],
    "fabind_edge_ln_concat": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _i, _vp],
    "fabind_edge_lnfold": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp, _f, ctypes.c_uint, _vp],
    "fabind_edge_lnfold_bwd_blocks": [_i],
    "fabind_edge_lnfold_bwd": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _vp],
    "fabind_inter_coord_fold": [_vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _i, _vp, _f, ctypes.c_uint, _vp],
    "fabind_post_optimize": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "fabind_multi_copy": [_vp, _i, _i, _vp],
    "fabind_zero_empty_rows": [_vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp],
    "fabind_split_sum": [_vp, _i, ctypes.c_long, _vp, _i, ctypes.c_long, _vp, _vp],
    "fabind_segment_sum": [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    "fabind_coord_update": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _vp],
    "fabind_cross_attn_fwd": [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i,
                              _vp],
    "fabind_cross_attn_mfma_fwd": [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i,
                                   _vp],
    "fabind_cross_attn_fused_fwd": [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp, _vp],
    "fabind_cross_attn_fused_bwd": [ctypes.POINTER(AttnFusedBwdArgs), _i, _i, _i, _i, _i, _i, _vp],
    "fabind_pair_bot_pack": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _l, _vp],
    "fabind_pair_bo_pack": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _l, _vp],
    "fabind_cross_attn_mfma_bwd": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp],
    "fabind_pair_bmat": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp],
    "fabind_pair_hadamard": [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp],
    "fabind_inter_attn_fwd": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                              _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_inter_attn_fwd_rows": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                   _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fabind_layernorm_rows_bwd": [_vp, _i, _i, _vp, _vp, _i, _i, _f, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp],
    "fabind_edge_concat": [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "fabind_las_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp],
    "fabind_select_rows": [_vp, _vp, _vp, _i, _i, _vp, _vp],
    "fabind_add": [_vp, _vp, _vp, _l, _vp],
    "fabind_mul_dact": [_vp, _i, _vp, _i, _i, _vp, _i, _l, _f, _vp],
    "fabind_mul_dact_colsum": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp],
    "fabind_drop_mix": [_vp, _vp, _vp, _l, _f, ctypes.c_uint, _vp],
    "fabind_drop_mix_bwd": [_vp, _vp, _vp, _l, _f, ctypes.c_uint, _vp],
    "fabind_mul_dropmask_colsum": [_vp, _i, _vp, _i, _i, _i, _f, ctypes.c_uint, _vp, _vp, _i, _vp],
    "fabind_rowdot_bwd": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "fabind_edge_geom_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "fabind_gcl_pre_bwd": [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "fabind_gather_dact": [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "fabind_coord_update_bwd": [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "fabind_cross_attn_bwd": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp],
    "fabind_pair_bias_cat": [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "fabind_batched_transpose_pad": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp],
    "fabind_node_chain_fwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp],
    "fabind_node_chain_x3_fwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i,
                                 _i, _i, _vp],
    "fabind_rows_hadamard_bwd": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "fabind_pair_hadamard_bwd_grid": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "fabind_pair_hadamard_bwd_rows": [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp],
    "fabind_pair_hadamard_bwd": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i,
                                 _vp],
    "fabind_inter_attn_bwd": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "fabind_inter_attn_bwd_rows": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp],
    "fabind_las_step_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp],
    "fabind_pair_bias_bwd": [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fabind_pair_bias_btcat": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp],
    "fabind_pair_bias_finish": [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _l, _vp],
    "fabind_pack_frag_multi": [_vp, _vp, _i, _vp],
    "fabind_lower_bound": [_vp, _i, _vp, _i, _vp, _vp],
    "fabind_pocket_center_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp],
    "fabind_pocket_center_bwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "fabind_pair_dist_fwd": [_vp, _i, _i, _vp, _vp, _f, _f, _f, _vp, _vp],
    "fabind_pair_dist_bwd": [_vp, _i, _i, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp],
    "fabind_block_hadamard_fwd": [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp],
    "fabind_block_hadamard_bwd": [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp],
    "fabind_loss_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _vp, _l, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_loss_bwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _l, _vp, _vp, _l, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp,
                        _vp, _vp, _vp, _vp, _vp, _vp],
    "fabind_layernorm_fwd": [_vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp],
    "fabind_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fabind_amd: %s is missing -- build it with `python -m fabind_amd.build` (hipcc, gfx950). "
            "There is no CPU/eager fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.fabind_last_error.restype = ctypes.c_char_p
    lib.fabind_abi_version.restype = ctypes.c_int
    got = lib.fabind_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("fabind_amd: %s has ABI version %d, this binding needs %d -- rebuild with `python -m fabind_amd.build`"
                           % (LIB_PATH, got, ABI_VERSION))
    lib.fabind_sizeof_args.argtypes, lib.fabind_sizeof_args.restype = [ctypes.c_int], ctypes.c_int
    lib.fabind_cross_attn_bwd_scratch.argtypes, lib.fabind_cross_attn_bwd_scratch.restype = [_i, _i, _i], ctypes.c_long
    lib.fabind_pair_bias_cat_parts.argtypes, lib.fabind_pair_bias_cat_parts.restype = [_i, _i], ctypes.c_long
    lib.fabind_cross_attn_fused_bwd_scratch.argtypes, lib.fabind_cross_attn_fused_bwd_scratch.restype = [_i, _i, _i, _i, _i], ctypes.c_long
    lib.fabind_cross_attn_fused_bwd_parts.argtypes, lib.fabind_cross_attn_fused_bwd_parts.restype = [_i, _i], ctypes.c_int
    lib.fabind_pair_bias_finish_parts.argtypes, lib.fabind_pair_bias_finish_parts.restype = [_i], ctypes.c_int
    lib.fabind_loss_blocks.argtypes, lib.fabind_loss_blocks.restype = [_l, _l, _l], ctypes.c_int
    for nm in ("fabind_pair_block_tile", "fabind_pair_block_chunk"):
        getattr(lib, nm).argtypes, getattr(lib, nm).restype = [], ctypes.c_int
    for which, mirror in enumerate((GemmArgs, EdgeBwdArgs, PairUpdateArgs, TnJob, AttnFusedBwdArgs)):
        if lib.fabind_sizeof_args(which) != ctypes.sizeof(mirror):
            raise RuntimeError("fabind_amd: ctypes mirror %s is %d bytes, the library's struct is %d -- _lib.py and "
                               "include/fabind_hip.h disagree" % (mirror.__name__, ctypes.sizeof(mirror),
                                                                  lib.fabind_sizeof_args(which)))
    lib.fabind_gemm_set_config.argtypes = [ctypes.c_int]
    lib.fabind_gemm_set_config.restype = None
    lib.fabind_gemm_set_persistent.argtypes = [ctypes.c_int]
    lib.fabind_gemm_set_persistent.restype = None
    lib.fabind_gemm_set_small_m.argtypes = [ctypes.c_int]
    lib.fabind_gemm_set_small_m.restype = None
    lib.fabind_gemm_set_x3_tile.argtypes = [ctypes.c_int]
    lib.fabind_gemm_set_x3_tile.restype = None
    if os.environ.get("FABIND_GEMM_SMALL_M"):
        lib.fabind_gemm_set_small_m(int(os.environ["FABIND_GEMM_SMALL_M"]))
    lib.fabind_gemm_tn_set_waves.argtypes = [ctypes.c_int]
    lib.fabind_gemm_tn_set_waves.restype = None
    lib.fabind_gemm_tn_set_exp.argtypes = [ctypes.c_int]
    lib.fabind_gemm_tn_set_exp.restype = None
    lib.fabind_gemm_tn_tile_n.argtypes = []
    lib.fabind_gemm_tn_tile_n.restype = ctypes.c_int
    lib.fabind_gcl_edge_fused_set_xcd_aware.argtypes = [ctypes.c_int]
    lib.fabind_gcl_edge_fused_set_xcd_aware.restype = None
    for nm in ("fabind_gcl_edge_fused_bwd3_set_exp", "fabind_gcl_edge_fused_bwd4_set_exp"):      # development knobs (void)
        getattr(lib, nm).argtypes = [ctypes.c_int]
        getattr(lib, nm).restype = None
    if os.environ.get("FABIND_EDGE_BWD3_EXP"):           # development knob: experiment mask of the store-wave backward (32 = nt operand stores)
        lib.fabind_gcl_edge_fused_bwd3_set_exp(int(os.environ["FABIND_EDGE_BWD3_EXP"]))
    if os.environ.get("FABIND_EDGE_BWD4_EXP"):           # sensitivity mask of the saved-forward backward (results wrong: timing only)
        lib.fabind_gcl_edge_fused_bwd4_set_exp(int(os.environ["FABIND_EDGE_BWD4_EXP"]))
    if "FABIND_EDGE_BWD_VARIANT" in os.environ:          # development knobs for same-box A/B runs (tools/probes)
        lib.fabind_gcl_edge_fused_bwd_set_variant.argtypes = [ctypes.c_int]
        lib.fabind_gcl_edge_fused_bwd_set_variant(int(os.environ["FABIND_EDGE_BWD_VARIANT"]))
    if os.environ.get("FABIND_EDGE_FWD_VARIANT"):
        lib.fabind_gcl_edge_fused_set_variant.argtypes = [ctypes.c_int]
        lib.fabind_gcl_edge_fused_set_variant(int(os.environ["FABIND_EDGE_FWD_VARIANT"]))
    if os.environ.get("FABIND_TN_WAVES"):                        # development knob: work-group layout of the TN contraction
        lib.fabind_gemm_tn_set_waves(int(os.environ["FABIND_TN_WAVES"]))
    if "FABIND_EDGE_XCD" in os.environ:
        lib.fabind_gcl_edge_fused_set_xcd_aware(int(os.environ["FABIND_EDGE_XCD"]))
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = argt
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


_DEBUG_SYNC = os.environ.get("FABIND_DEBUG_SYNC", "0") == "1"     # development aid: surface a device fault at its launch


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, load().fabind_last_error().decode()))
    if _DEBUG_SYNC:
        import torch
        print("[fabind] " + what, flush=True)
        torch.cuda.synchronize()


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: no silent fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("fabind_amd kernels need HIP device tensors; got a %s tensor" % t.device)
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """HIP stream handle of torch's current stream on the current device.  Every kernel launch asks for it: the raw accessors
    (what torch's own extensions use) cost ~0.5 us; torch.cuda.current_stream() builds a Stream object through three Python layers,
    ~9 us -- 1.3 ms per step of the pocket-sized shape's forward alone (tools/probes/stack_hostprof.py)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def dt_code(dtype):
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError("unsupported dtype %s" % dtype)
