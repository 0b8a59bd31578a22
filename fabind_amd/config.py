"""Process-wide numeric mode of the HIP path."""
_PRECISION = {"mode": "bf16"}
MODES = ("fp32", "bf16", "bf16x3")


def set_precision(mode):
    """'fp32': exact-fp32 MFMA everywhere (parity mode, 1e-4 A gate; 1/16 of the bf16 matrix rate).
    'bf16': bf16 MFMA operands with fp32 accumulation; coordinates, radial terms, softmax and all
    reductions stay fp32; edge-level intermediates are stored as bf16.
    'bf16x3': fp32 storage everywhere like 'fp32', but every contraction runs as SPLIT bf16 on the bf16 matrix cores: an operand
    element x is hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significand bits), a product term is three MFMAs
    (lo*hi + hi*lo + hi*hi) with fp32 accumulation -- fp32-grade results (meets the 1e-4 A gate) at 3/16 of the fp32 matrix time."""
    assert mode in MODES
    _PRECISION["mode"] = mode


def get_precision():
    return _PRECISION["mode"]


def fp32_storage():
    """True in the modes that keep every activation / weight in fp32 ('fp32' and 'bf16x3')."""
    return _PRECISION["mode"] != "bf16"
