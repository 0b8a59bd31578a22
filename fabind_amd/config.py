"""Process-wide numeric mode of the HIP path."""
_PRECISION = {"mode": "bf16"}


def set_precision(mode):
    """'fp32': exact-fp32 MFMA everywhere (parity mode, 1e-4 A gate).
    'bf16': bf16 MFMA operands with fp32 accumulation; coordinates, radial terms, softmax and all
    reductions stay fp32; edge-level intermediates are stored as bf16."""
    assert mode in ("fp32", "bf16")
    _PRECISION["mode"] = mode


def get_precision():
    return _PRECISION["mode"]
