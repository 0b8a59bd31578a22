"""GPU: bench.py's N>1 control flow (torch.distributed.run, per-rank batches, gradient all-reduce, barrier + MAX timing,
one JSON line from rank 0) exercised with two ranks on ONE device through the gloo test hook (RCCL needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--steps", "2", "--warmup", "1", "--batch", "3", "--n-prot", "70", "--n-lig", "9", "--hidden", "64", "--layers", "2",
         "--no-cpu-baseline", "--no-extras"]


def test_bench_json_contract_single_rank():
    r = _run([sys.executable, "bench.py"] + SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["unit"] == "complexes/s" and r["higher_is_better"] is True and r["data"] == "synthetic"
    assert abs(r["value"] - 3 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # the fraction is reproducible from the line itself -- executed flops (or compulsory bytes) per launch / average launch time / peak
    per = rf["flop_per_launch"] / 1e12 if rf["bound"] == "mfma" else rf["algorithmic_bytes_per_launch"] / 1e9
    assert abs(rf["achieved"] - per / (rf["avg_us"] * 1e-6)) <= 1e-6 * max(rf["achieved"], 1e-9)
    assert rf["design_bytes_per_launch"] >= 0 and rf["other_roofline"]["bound"] != rf["bound"]


def test_bench_two_ranks_control_flow():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2"] + SMALL,
             env={"FABIND_BENCH_DEVICE": "0", "FABIND_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 6
    assert abs(r["value"] - 6 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]      # whole-job aggregate over both ranks


def test_bench_eight_ranks_control_flow():
    """`--gpus 8` at a tiny shape, eight ranks on ONE device over gloo (the driver's 8-GPU run must not be the first
    time this control flow executes): per-rank batches, GradReducer through the full model's 33 never-used tensors, barrier + MAX timing,
    whole-job aggregate."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
              "--master-port", "29537", "bench.py", "--gpus", "8"] + SMALL,
             env={"FABIND_BENCH_DEVICE": "0", "FABIND_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 8 and r["config"]["global_batch"] == 24 and r["scaling"] == "weak"
    assert abs(r["value"] - 24 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]


def test_bench_self_launches_without_a_launcher():
    """`python bench.py --gpus 2` with no torch.distributed.run around it: the parent spawns the ranks before touching the GPU,
    relays rank 0's line and exit code (it used to die on an assert)."""
    env = {"FABIND_BENCH_DEVICE": "0", "FABIND_BENCH_BACKEND": "gloo"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert k not in os.environ
    r = _run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, env=env)
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 6
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, cwd=ROOT,       # no such device in the children
                         env=dict(os.environ, FABIND_BENCH_DEVICE="99", FABIND_BENCH_BACKEND="gloo"), capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0                                   # a failing child is reported through the exit code


SUB_OBJECTS = ("pocket", "gate_mode", "gate_mode_exact_bwd", "fp32", "train_mode", "n_iter8", "n_iter8_gate", "fwd", "model_fwdbwd",
               "model_gate", "gate_mode_bf16_edge", "n_iter8_gate_bf16_edge", "stack_fwdbwd", "config3_gate", "model_fwdbwd_train_n_iter8", "model_fwdbwd_train_n_iter8_bf16", "plus_train", "plus_train_gate", "plus_sampling")


def test_bench_line_carries_the_neighbouring_configurations():
    """Default mode at N=1: EVERY sub-object of the driver line rides in the same JSON line and none of them is an `{"error": ...}`
    (bench.py swallows a failing sub-object so that the headline survives; this test is what fails instead -- VERDICT r3 weak 19)."""
    r = _run([sys.executable, "bench.py", "--poses", "2"] + SMALL[:-1])
    for k in SUB_OBJECTS:
        assert k in r and "error" not in r[k], (k, r.get(k))
        assert r[k]["value"] > 0 and r[k]["unit"] == ("poses/s" if k == "plus_sampling" else "complexes/s")
        assert r[k]["steps"] >= 1 and r[k]["ms_per_step"] > 0
    assert r["fp32"]["dtype"] == "fp32" and r["train_mode"]["train_mode"] is True and r["n_iter8"]["n_iter"] == 8
    for k in ("gate_mode", "gate_mode_exact_bwd", "n_iter8_gate", "model_gate", "plus_train_gate", "model_fwdbwd_train_n_iter8"):
        assert r[k]["dtype"] == "bf16x3", k
    assert r["pocket"]["nodes"].startswith("100 protein") and r["pocket"]["pass"] == "fwdbwd"
    assert r["stack_fwdbwd"]["pass"] == "fwdbwd" and r["config3_gate"]["pass"] == "model" and r["config3_gate"]["dtype"] == "bf16x3"
    # the headline is BASELINE configs[2] read literally and says so: the full model on the whole graph, the six losses named
    assert r["config"]["pass"] == "model" and all(w in r["metric"] for w in ("pocket-cls", "pocket-centre", "coord", "distmap", "distill"))
    # north_star's targets ride in the line with where this build stands, said plainly (the 30 % matrix-core bar of the cross attention is NOT met)
    ns = r["north_star_targets"]
    assert ns["mfma_util_cross_attention"]["met"] is False and ns["mfma_util_cross_attention"]["target"] == 0.30
    assert ns["fwd_bwd_complexes_per_s_8gpu"]["target"] == 2000.0 and ns["fwd_complexes_per_s_8gpu"]["this_run_stack_forward_one_gpu"] == r["fwd"]["value"]
    # nothing else in the line is an unreported failure either
    assert not [k for k, v in r.items() if isinstance(v, dict) and "error" in v]


def test_rccl_backend_paths_with_one_rank():
    """The device-tensor collective branch of fabind_amd.parallel.allreduce_gradients and bench.py's "nccl" (= RCCL) process
    group run on a real GPU with one rank (the two-rank tests above need gloo and stage through the host): a one-rank SUM is the
    identity, so allreduce_gradients(..., world=2) must leave exactly grad / 2 over several buckets (checked inside the probe),
    and bench.py under torch.distributed.run must initialise, barrier, time and print its line with the RCCL group."""
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    out = subprocess.run(launch + ["--master-port", "29541", os.path.join("tools", "probes", "nccl_single_rank.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL single-rank path ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
    r = _run(launch + ["--master-port", "29542", "bench.py", "--gpus", "1"] + SMALL)
    assert r["n_gpus"] == 1 and r["value"] > 0
