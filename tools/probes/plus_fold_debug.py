"""Debug: FABind+ GCL layer with / without the LN-fold path, same dropout seeds -> outputs should agree to bf16 noise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, random
from fabind_amd import engine
from fabind_amd.plus import engine as pe
from helpers import load_npz
from test_gpu_plus import _sampling_model, hetero_from_npz
dev = torch.device("cuda:0")
engine.set_precision("bf16")
g = load_npz("plus_model_sampling_tiny")
m = _sampling_model(g, dev)
for mode in ("eval", "train"):
    m.train(mode == "train")
    outs = []
    for fold in (True, True, False, False, True):
        pe.FOLD_EDGE_LN = fold
        torch.manual_seed(0); random.seed(1)
        with torch.no_grad():
            c = m.inference(hetero_from_npz(g).to(dev))[0]
        outs.append(c)
    for k in range(1, 5):
        print(mode, "run 0 vs run %d: max |d| %.4f A, mean %.4f" % (k, (outs[0] - outs[k]).abs().max().item(), (outs[0] - outs[k]).abs().mean().item()))
