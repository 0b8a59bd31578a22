"""Debug: per-call difference between the folded first edge Linear and LayerNorm -> Linear in the real model (eval)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, random
from fabind_amd import engine, ops, kernels as K
from fabind_amd.plus import engine as pe
from helpers import load_npz
from test_gpu_plus import _sampling_model, hetero_from_npz
dev = torch.device("cuda:0")
engine.set_precision("bf16")
g = load_npz("plus_model_sampling_tiny")
m = _sampling_model(g, dev)
m.eval()
orig = pe.gcl_layer
calls = [0]
def dbg(p, h, x, lay, gr, clampv, pd=0.0):
    e = p["edge"]; H = h.shape[1]
    d, rhohat = ops.edge_geom(x, gr.row_ctx, gr.col_ctx, gr.rp_ctx, lay, gr.ctx_by_col)
    y = K.edge_ln_concat(h, gr.row_ctx, gr.col_ctx, rhohat, e["ln_w"], e["ln_b"], torch.bfloat16, e["k_pad"])
    t_ref = ops.linear(y, e["W1"], e["b1"], act_epi=K.ACT_RELU, out_dtype=torch.bfloat16).float()
    f = p["fold"]
    hf = h.float(); mean = hf.mean(1); hcen = hf - mean[:, None]
    stat = torch.stack([mean, (hcen * hcen).sum(1)], 1).contiguous()
    AB = ops.linear(hcen, f["W_ab"], None, out_dtype=torch.bfloat16)
    t = K.edge_lnfold(AB, e["k_pad"], H, gr.row_ctx, gr.col_ctx, rhohat, stat, f["eps"], f["w_r"], f["c_r"], f["c_c"], f["dvec"]).float()
    if calls[0] < 12:
        print("call %d  E=%d H=%d  |h| max %.2f mean-of-means %.3f  t_ref max %.3f  max|dt| %.4f mean|dt| %.5f" % (
            calls[0], t.shape[0], H, h.abs().max().item(), mean.abs().mean().item(), t_ref.abs().max().item(),
            (t - t_ref).abs().max().item(), (t - t_ref).abs().mean().item()))
    calls[0] += 1
    return orig(p, h, x, lay, gr, clampv, pd)
pe.gcl_layer = dbg
random.seed(1)
with torch.no_grad():
    m.inference(hetero_from_npz(g).to(dev))
