"""Probe: does the v1 stack in bf16 show the same coordinate gap vs its oracle when the ligands are moved far away
(checks the attribution of the FABind+ finding to the shared bf16 arithmetic)?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
    sys.path.insert(0, p)
import torch
import fabind_oracle as orc
from fabind_amd import engine, synthetic
from helpers import rmsd
from test_gpu_stack import _random_stack, _run
dev = torch.device("cuda:0")
for shift in (0.0, 5.0, 50.0):
    inp = synthetic.make_stack_batch([(40, 6), (35, 5)], 32, seed=9)
    lig = (inp["segment_id"] == 0) & ~inp["is_global"]
    inp["X"][lig] += shift
    m = _random_stack(32, 2, 2, 21)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    Xr, Hr = orc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                               inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"], 2, 2)
    for mode in ("fp32", "bf16"):
        engine.set_precision(mode)
        X, H = _run(m.to(dev), inp, dev)
        engine.set_precision("fp32")
        mask = inp["mask"].numpy()
        print("v1 shift %5.1f %s  E_int %5d  ligand RMSD %.2e A  max|dH| %.4f" % (
            shift, mode, m.last_graph.E_int, rmsd(X.cpu().numpy()[mask] * 5, Xr.numpy()[mask] * 5), float((H.cpu() - Hr).abs().max())))
