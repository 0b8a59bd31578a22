"""PROBE: cycle counters of one work-group of cross_attn_fused_fwd_kernel at the headline block shapes: a0 staging / bias contraction /
bias epilogue / attention, both modes."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabind_amd import _lib, config, engine, ops, synthetic
dev = torch.device("cuda:0")
config.set_precision("bf16")
lib = _lib.load()
lib.fabind_cross_attn_fused_set_dbg.argtypes, lib.fabind_cross_attn_fused_set_dbg.restype = [ctypes.c_void_p], None
H = 512
inp = synthetic.make_stack_batch([(1500, 40)] * 64, 8, seed=0, snap=False)
lay = engine.Layout(inp["batch_id"].to(dev), inp["segment_id"].to(dev))
g = torch.Generator().manual_seed(0)
a0b0 = (torch.randn(lay.N, 2 * H, generator=g) * 0.5).to(dev)
wcomp = (torch.randn(2, 8, H, generator=g) / H ** 0.5).to(dev)
bconst = torch.randn(2, 8, generator=g).to(dev)
pb = ops.PairBias(a0b0, H, wcomp, bconst, lay)
qg_p, kv_p = torch.randn(lay.N, 256, device=dev), torch.randn(lay.sumC, 256, device=dev)
qg_c, kv_c = torch.randn(lay.sumC, 256, device=dev), torch.randn(lay.N, 256, device=dev)
dbg = torch.zeros(5, dtype=torch.int64, device=dev)
scale = 1.0 / math.sqrt(32.0)
names = ["a0 tile staged", "bias contraction", "bias epilogue -> LDS", "attention"]
for mode, (qg, kv) in enumerate(((qg_p, kv_p), (qg_c, kv_c))):
    for rep in range(3):
        lib.fabind_cross_attn_fused_set_dbg(dbg.data_ptr())
        with torch.no_grad():
            ops.cross_attn_fused(qg, kv, pb, mode, mode, lay, scale)
        torch.cuda.synchronize()
        t = dbg.cpu().tolist()
        d = [t[i + 1] - t[i] for i in range(4)]
        print("PHASES mode %d rep %d: " % (mode, rep) + ", ".join("%s %d" % (n, c) for n, c in zip(names, d)) + "  (total %d cycles of the 100 MHz counter x 24 = shader clocks?)" % (t[4] - t[0]))
lib.fabind_cross_attn_fused_set_dbg(None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode, (qg, kv) in enumerate(((qg_p, kv_p), (qg_c, kv_c))):
    with torch.no_grad():
        ops.cross_attn_fused(qg, kv, pb, mode, mode, lay, scale)
        e0.record()
        for _ in range(20):
            ops.cross_attn_fused(qg, kv, pb, mode, mode, lay, scale)
        e1.record()
    torch.cuda.synchronize()
    print("TIME mode %d: %.1f us per launch" % (mode, e0.elapsed_time(e1) * 50))
