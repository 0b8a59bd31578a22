"""Probe: what would the two adjoint contractions of the pair bias cost on the pipelined NT GEMM with PLAIN groups, if the
gradient operand D were laid out uniformly padded per complex?  (a) dA[P x 512] += D_b[P x 352] . BT_b[512 x 352]^T (K = 8 C_b = 328
padded to 352), accumulating into a strided fp32 output; (b) T_b[352 x 512] = Dt_b[352 x 1504] . At_b[512 x 1504]^T (K = P_b = 1500
padded to 1504).  Random data, timing only; today the two run at ~437 us (fallback NT kernel) and ~540 us (grouped TN kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabind_amd import kernels as K
dev = torch.device("cuda:0")
B, P, C8, H = 64, 1500, 328, 512
Pp, Kp = 1504, 352
def time(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lo = lambda v: v - (1 << 32) if (v & 0xFFFFFFFF) >= (1 << 31) else (v & 0xFFFFFFFF)
# (a)
D = torch.randn(B * Pp, Kp, device=dev).bfloat16()
BT = torch.randn(B * H, Kp, device=dev).bfloat16()
ld_out = 1024
out = torch.zeros(B * 1542, ld_out, device=dev)
ga = torch.tensor([[b * Pp, P, b * H, H, lo((b * 1542 + 42) * ld_out), ((b * 1542 + 42) * ld_out) >> 32, ld_out, 0] for b in range(B)],
                  dtype=torch.int32, device=dev)
ta = time(lambda: K.gemm(D, BT, out=out, accumulate=True, groups=ga, n_groups=B, max_m=P, max_n=H, M=B * Pp, N=B * H, ldc=ld_out))
# (b)
Dt = torch.randn(B * Kp, Pp, device=dev).bfloat16()
At = torch.randn(B * H, Pp, device=dev).bfloat16()
T = torch.zeros(B * C8, H, device=dev)
gb = torch.tensor([[b * Kp, C8, b * H, H, lo(b * C8 * H), (b * C8 * H) >> 32, H, 0] for b in range(B)], dtype=torch.int32, device=dev)
tb = time(lambda: K.gemm(Dt, At, out=T, groups=gb, n_groups=B, max_m=C8, max_n=H, M=B * Kp, N=B * H, ldc=H))
fl = 2.0 * B * P * C8 * H
print("(a) dA += D . BT^T   grouped x%d  M=%d N=%d K=%d : %.1f us  (%.0f TFLOP/s)" % (B, P, H, Kp, ta, fl / ta / 1e6))
print("(b) T  = Dt . At^T   grouped x%d  M=%d N=%d K=%d : %.1f us  (%.0f TFLOP/s)" % (B, C8, H, Pp, tb, fl / tb / 1e6))
print("sum %.1f us vs ~437 + ~540 us today" % (ta + tb))
# (c) the eight blocks' (a) contractions as ONE launch: K concatenated (8 x 352 = 2816), one accumulating epilogue pass
Kc = 8 * Kp
Dc = torch.randn(B * P, Kc, device=dev).bfloat16()
BTc = torch.randn(B * H, Kc, device=dev).bfloat16()
gc = torch.tensor([[b * P, P, b * H, H, lo((b * 1542 + 42) * ld_out), ((b * 1542 + 42) * ld_out) >> 32, ld_out, 0] for b in range(B)],
                  dtype=torch.int32, device=dev)
tc = time(lambda: K.gemm(Dc, BTc, out=out, accumulate=True, groups=gc, n_groups=B, max_m=P, max_n=H, M=B * P, N=B * H, ldc=ld_out), n=10)
print("(c) dA += [D_0|..|D_7] . [BT_0|..|BT_7]^T  grouped x%d  M=%d N=%d K=%d : %.1f us  (%.0f TFLOP/s)  vs 8 x ~437 = ~3500 us today" % (
    B, P, H, Kc, tc, 8 * fl / tc / 1e6))
