"""PROBE: resident work-groups per CU the runtime grants the two round-3 kernels whose LDS footprint is exactly 80 KiB (half of a CU's 160 KiB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabind_amd import _lib
torch.zeros(1, device="cuda:0")
lib = _lib.load()
for f in ("fabind_gemm_x3_occupancy", "fabind_cross_attn_fused_occupancy"):
    getattr(lib, f).argtypes, getattr(lib, f).restype = [ctypes.c_int], ctypes.c_int
print("OCC gemm_x3<128 rows> (81,920 B LDS):", lib.fabind_gemm_x3_occupancy(2), " <256 rows> (122,880 B):", lib.fabind_gemm_x3_occupancy(4))
for b in (81920, 80896, 79872, 73728, 65536):
    print("OCC cross_attn_fused_fwd<512,0> at %d B of LDS:" % b, lib.fabind_cross_attn_fused_occupancy(b))
