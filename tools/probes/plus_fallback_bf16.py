"""Probe: bf16 gap of the FABind+ stack vs the oracle as the ligands are moved away from their proteins (E_int drops to
the 2-edge reference fallback), with the inference-only paths (LayerNorm folds, fused pair update) on and off."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import fabind_plus_oracle as porc
from fabind_amd import engine
from fabind_amd.plus import engine as pe
from helpers import load_npz, rmsd, stack_inputs, weights
from test_gpu_plus import _build, _run
dev = torch.device("cuda:0")
g = load_npz("plus_stack_tiny_it2")
hidden, layers, n_iter, _ = [int(v) for v in g["cfg"]]
for shift in (0.0, 5.0, 12.5, 25.0, 50.0):
    inp = stack_inputs(g)
    lig = (inp["segment_id"] == 0) & ~inp["is_global"]
    inp["X"] = inp["X"].clone(); inp["X"][lig] += shift
    sd = {k: v for k, v in weights(g).items()}
    with torch.no_grad():
        Xr, Hr, Zr = porc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"], inp["is_global"],
                                        inp["compound_edge_index"], inp["LAS_edge_index"], inp["coord_LAS"], layers, n_iter)
    mask = inp["mask"].numpy()
    for fold, fuse in ((True, True), (False, False)):
        pe.FOLD_EDGE_LN, pe.FUSE_PAIR = fold, fuse
        engine.set_precision("bf16")
        m = _build(g, dev)
        X, H, Z = _run(m, inp, dev)
        engine.set_precision("fp32")
        print("shift %5.1f fold %-5s fuse_pair %-5s  E_int %6d  ligand RMSD %.4f A  max|dH| %.4f  max|dZ| %.4f" % (
            shift, fold, fuse, m.last_graph.E_int, rmsd(X.cpu().numpy()[mask] * 5, Xr.numpy()[mask] * 5),
            float((H.cpu() - Hr).abs().max()), float((Z.cpu() - Zr).abs().max())))
