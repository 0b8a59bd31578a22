"""A minimal caller in the style of the reference's scripts (FABind/fabind/test_fabind.py:233-240: `from models.model import *`,
`get_model`, strict `load_state_dict`, eval, forward) that runs the binding INTEGRATION.md section 1 prescribes -- the
python block is taken from the text of INTEGRATION.md and executed as is.  Not a copy of any reference script: it loads a
reference-captured state_dict + batch from tests/golden/model_tiny.npz and checks `model.inference` against the reference's
output stored there.  Run by tests/test_gpu_dense_api.py in a fresh interpreter."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))          # tests/test_gpu_stack.py (the args helper) imports the checker

text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
section = text[text.index("## 1."):]
block = re.search(r"```python\n(.*?)```", section, re.S).group(1)
assert 'sys.modules["models." + _name]' in block
exec(compile(block, "INTEGRATION.md#1", "exec"))                # the alias recipe, verbatim (ends with set_precision("bf16"))

import numpy as np                                              # noqa: E402
import torch                                                    # noqa: E402
from models.model import *                                      # noqa: E402,F401,F403  -- what the reference's scripts do
import models.att_model, models.egnn, models.cross_att, models.model_utils   # noqa: E402,E401
from fabind_amd import engine                                   # noqa: E402
from helpers import hetero_from_npz, load_npz, rmsd, weights    # noqa: E402
from test_gpu_stack import _args                                # noqa: E402

assert models.egnn.MC_Att_L.__module__.startswith("fabind_amd.models")


class Logger:
    def log_message(self, s):
        print("[logger]", s)


g = load_npz("model_tiny")
hidden, pocket_hidden, layers, n_iter, _ = [int(v) for v in g["cfg"]]
args = _args(hidden, layers, n_iter)
args.pocket_pred_hidden_size, args.random_n_iter = pocket_hidden, False
device = torch.device("cuda:0")
model = get_model(args, Logger(), device)                       # noqa: F405
model.load_state_dict(weights(g), strict=True)                  # 'ckpt/best_model.bin' in the reference's scripts
model.to(device)
model.eval()
res = {}
for prec in ("bf16", "fp32"):
    engine.set_precision(prec)
    data = hetero_from_npz(g).to(device)
    with torch.no_grad():
        coords, batch = model.inference(data)
    res[prec] = rmsd(coords.cpu().numpy(), g["inf_coords"])
    assert "batch" in data["complex"]                           # side effect of the reference's inference (model.py:560)
print("ligand RMSD vs the reference's inference output: bf16 %.3e A, fp32 %.3e A" % (res["bf16"], res["fp32"]))
assert res["fp32"] < 1e-4 and res["bf16"] < 5e-2
print("integration recipe ok")
