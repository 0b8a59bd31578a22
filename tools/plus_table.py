"""Per-shape table of the profiled launches in one FABind+ sampling pose batch (B=64, 1500/40, n_iter=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess, torch
import bench
from fabind_amd import engine, kernels as K, synthetic
from fabind_amd.plus.models import get_model
dev = torch.device("cuda:0")
engine.set_precision("bf16")
a = bench.stack_args(512, 5, 8)
for k_, v_ in dict(use_ln_mlp=True, mlp_hidden_scale=1, dropout=0.1, mha_heads=4, rel_dis_pair_bias="no", inter_additional_mlp=False,
                   only_last_LAS=False, geom_reg_steps=1, use_for_radius_pred="ligand", dis_map_thres=15.0, pocket_radius_buffer=5.0,
                   min_pocket_radius=20.0, force_fix_radius=False, use_clustering=False, confidence_training=False).items():
    setattr(a, k_, v_)
class L:
    def log_message(self, m): pass
torch.manual_seed(0)
m = get_model(a, L()).to(dev); m.train()
hb = synthetic.make_hetero_batch([(1500, 40)] * 4 * 16, seed=0).to(dev)
m.inference(hb.clone()); torch.cuda.synchronize()
K.PROFILE = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); m.inference(hb.clone()); e1.record(); torch.cuda.synchronize()
print("one pose batch (64 complexes): %.1f ms" % e0.elapsed_time(e1))
rows = []
for label, evs in K.PROFILE.items():
    ms = sum(x.elapsed_time(y) for x, y, _ in evs)
    rows.append((ms, len(evs), evs[0][2], label))
rows.sort(reverse=True)
print("profiled GEMM-class launches: %.1f ms" % sum(r[0] for r in rows))
for ms, n, fl, label in rows[:30]:
    print("%8.3f ms %4d x  %7.1f TF/s  %s" % (ms, n, fl * n / ms / 1e9 if ms > 0 else 0, label[:100]))
