"""cProfile of the host side of one FABind+ sampling pose batch (B=64, 1500/40, n_iter=8)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fabind_amd import engine, synthetic
from fabind_amd.plus.models import get_model
dev = torch.device("cuda:0")
engine.set_precision("bf16")
a = bench.stack_args(512, 5, 8)
for k_, v_ in dict(use_ln_mlp=True, mlp_hidden_scale=1, dropout=0.1, mha_heads=4, rel_dis_pair_bias="no", inter_additional_mlp=False,
                   only_last_LAS=False, geom_reg_steps=1, use_for_radius_pred="ligand", dis_map_thres=15.0, pocket_radius_buffer=5.0,
                   min_pocket_radius=20.0, force_fix_radius=False, use_clustering=True, dbscan_eps=9.0, dbscan_min_samples=2, choose_cluster_prob=0.5, confidence_training=False).items():
    setattr(a, k_, v_)
class L:
    def log_message(self, m): pass
torch.manual_seed(0)
m = get_model(a, L()).to(dev); m.train()
hb = synthetic.make_hetero_batch([(1500, 40)] * 64, seed=0).to(dev)
for _ in range(2):
    m.inference(hb.clone())
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.time()
pr.enable()
m.inference(hb.clone())
pr.disable()
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print("host %.1f ms, +drain %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
