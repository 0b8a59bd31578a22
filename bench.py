"""bench.py -- throughput of the FABind docking hot path on N MI355X GPUs (one process per GPU).

python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher (WORLD_SIZE unset): this process never touches the GPU; it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`
    as a child, relays rank 0's JSON line and exits with the child's code.  Under a launcher (the driver's way) the
    ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A "step" (default mode `config3` = BASELINE.json configs[2] read literally) = forward + backward of the full IaBNet -- pocket model,
4 FABind layers + out layer at hidden 512 on ALL 1500 protein / 40 ligand nodes of every complex (unbounded pocket radius), distance-map
head over all 60,000 pairs per complex -- with the reference's six-term loss (pocket-cls + pocket-centre + coord + two distance-map terms
+ distill, main_fabind.py:398-417), over one synthetic batch of --batch complexes per GPU resident in HBM, a fresh batch object per
step.  `--mode fwdbwd` / `fwd`: the layer stack alone (rounds 1-5's headline, now the `stack_fwdbwd` sub-object).  Prints ONE JSON line
on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# dense matrix-core peaks (MI355X_MICROARCH.md); "bf16x3": three bf16 MFMAs per product term, so a third of the bf16 peak in USEFUL flops
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3.0}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


# which roofline a kernel is REPORTED against is a property of the operation, not of whichever fraction happens to be larger
# (rounds 3-5 picked max(frac) over this design's own tile traffic, and the label flipped mfma -> hbm without the kernel becoming
# byte-minimal).  Dense contractions -- the fused GCL edge pipeline both ways (8(d): "intensity ~ 15 kFLOP/B => MFMA-bound; report that
# kernel against the MFMA roofline"), the pair path, the attention blocks, every GEMM -- are priced on EXECUTED flops against the dense
# matrix-core peak of the dtype; stand-alone gather / scatter / softmax / edge-build kernels on their 8(d) bytes against HBM.
MFMA_FAMILIES = ("gcl_edge_fused", "cross_attn", "pair_update_fused", "fabind_gemm", "fabind_node_chain")


def bound_of(label):
    return "mfma" if label.startswith(MFMA_FAMILIES) else "hbm"


def price_against_rooflines(flops, algo_bytes, ms, precision, bound, design_bytes=None):
    """Roofline object of a set of launches that ran `ms` milliseconds: `flops` executed, `algo_bytes` = SURVEY 8(d)'s compulsory bytes of
    the operation, `design_bytes` = what this design's launches read + write once (>= algo_bytes when tiles are materialised).
    `bound` ("mfma" | "hbm") comes from 8(d)'s classification (bound_of), never from comparing the two fractions; the other view rides along."""
    tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    gb = algo_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    mfma = {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS[precision], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS[precision]}
    hbm = {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS}
    out = dict(mfma if bound == "mfma" else hbm)
    out["other_roofline"] = hbm if bound == "mfma" else mfma
    if design_bytes is not None:
        dg = design_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["design_traffic"] = {"achieved": dg, "unit": "GB/s", "frac_of_hbm_peak": dg / HBM_PEAK_GBS,
                                 "note": "this design's own materialised tiles per launch over the launch time: a cost of the design, not the kernel's roofline"}
    return out


def stack_args(hidden, layers, n_iter):
    from argparse import Namespace
    return Namespace(
        mode=5, n_iter=n_iter, mean_layers=layers, hidden_size=hidden, pocket_pred_hidden_size=128,
        pocket_pred_layers=1, pocket_pred_n_iter=1, refine="refine_coord", coordinate_scale=5.0,
        geometry_reg_step_size=0.001, rm_layernorm=True, add_attn_pair_bias=True, explicit_pair_embed=True,
        add_cross_attn_layer=True, norm_type="per_sample", random_n_iter=False, center_dist_threshold=4.0,
        stage_prob=0.25, distmap_pred="mlp", use_esm2_feat=True, esm2_concat_raw=False, inter_cutoff=10.0,
        intra_cutoff=8.0, pocket_radius=20.0, gs_tau=1.0, gs_hard=False, local_eval=False,
        train_pred_pocket_noise=0.0, compound_coords_init_mode="pocket_center_rdkit", ablation_no_attention=False,
        ablation_no_attention_with_cross_attn=False, keep_trig_attn=False, opm=False, rm_F_norm=False,
        fix_pocket=False, rm_LAS_constrained_optim=False)


def build_model(hidden, layers, n_iter, seed=0, dropout=0.0):
    from fabind_amd.models.att_model import EfficientMCAttModel
    torch.manual_seed(seed)
    m = EfficientMCAttModel(stack_args(hidden, layers, n_iter), hidden, hidden, 1, n_layers=layers, n_iter=n_iter,
                            dropout=dropout, normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0)
    from fabind_amd import synthetic
    if LEGACY_BATCH:
        return m
    return synthetic.condition_for_large_graphs(m)      # random init kept out of the |h| ~ 1e6 regime (see its docstring)


PREFETCH = {"1": True, "0": False}.get(os.environ.get("FABIND_BENCH_PREFETCH", ""))   # next batch's layout + input graph on a feeder stream
prefetching = [False]                                                     # (engine.prefetch); None = the per-mode default below
REUSE_BATCH = os.environ.get("FABIND_BENCH_REUSE_BATCH", "0") == "1"    # round 2's protocol (one resident batch, same index tensor objects every
                                                                       # step: its layout is built once): same-box comparisons with that tree only
LEGACY_BATCH = os.environ.get("FABIND_BENCH_LEGACY_BATCH", "0") == "1"    # round-1 workload (4 geometries tiled 16x, plain init): A/B only


def make_batch(batch, n_prot, n_lig, hidden, seed):
    """`batch` DISTINCT seeded synthetic complexes, host generation ~1 s at B=64."""
    from fabind_amd import synthetic
    if LEGACY_BATCH and batch > 4:          # what round 1's bench.py timed: same-box comparisons against that tree (tools/probes)
        base = synthetic.make_stack_batch([(n_prot, n_lig)] * 4, hidden, seed=seed, snap=False)
        reps, n = (batch + 3) // 4, base["X"].shape[0]
        out = {k: torch.cat([base[k]] * reps)[: n // 4 * batch] for k in ("X", "H", "segment_id", "mask", "is_global", "coord_LAS")}
        per = n // 4
        out["batch_id"] = torch.repeat_interleave(torch.arange(batch), per)
        shift = lambda e: torch.cat([e + i * n for i in range(reps)], 1)
        ce, le = shift(base["compound_edge_index"]), shift(base["LAS_edge_index"])
        out["compound_edge_index"], out["LAS_edge_index"] = ce[:, ce[0] < per * batch], le[:, le[0] < per * batch]
        return out
    return synthetic.make_stack_batch([(n_prot, n_lig)] * batch, hidden, seed=seed, snap=False)


def _cpu_run_fn(hidden, layers, n_iter, n_prot, n_lig, batch, backward):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fabind_oracle as orc
    m = build_model(hidden, layers, n_iter)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if backward:
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
    inp = make_batch(batch, n_prot, n_lig, hidden, seed=0)

    def run():
        with torch.set_grad_enabled(backward):
            X, Hh = orc.stack_forward(sd, "", inp["X"].clone(), inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"],
                                      inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"],
                                      inp["coord_LAS"], layers, n_iter)[:2]
            if backward:
                for v in sd.values():
                    v.grad = None
                ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
    return run


def cpu_baseline(hidden, layers, n_iter, n_prot, n_lig, budget_s=25.0, backward=False, batch=2):
    """The oracle (CPU restatement of the reference algorithm, `kind: port`) timed on the host cores -- the BOUNDED sample of the
    default run (~25 s: one warm-up, then as many timed runs as fit, at most 5).  backward=True: the same pass as the GPU step of the
    default mode -- forward, the same scalar loss, backward to every parameter (autograd through the oracle) -- so that
    `cpu_baseline.value` and `value` measure the same work.  The full SURVEY 8(d) protocol (B = 2, >= 3 warm-ups + 5 timed runs, all
    cores and one, per stack pass and per 8-iteration forward) takes tens of minutes of host time: `bench.py --cpu-baseline-full`
    runs it, its committed result (profiles/r04_cpu_baseline.json) is attached under `cpu_baseline.protocol_8d`."""
    # torch's default (128 threads on the 256-CPU host of the GPU box) is the WORST setting for this workload: forward
    # 0.12 complexes/s at 128 threads, 0.26 at 64, 0.39 at 32, 0.38 at 16, 0.31 at 8 (tools/probes/cpu_threads.py)
    default_threads = torch.get_num_threads()
    torch.set_num_threads(min(default_threads, 32))
    run = _cpu_run_fn(hidden, layers, n_iter, n_prot, n_lig, batch, backward)
    t0 = time.time()
    run()
    first = time.time() - t0
    reps = max(1, min(5, int(budget_s / max(first, 1e-3)) - 1))
    t0 = time.time()
    for _ in range(reps):
        run()
    dt = (time.time() - t0) / reps
    cores = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    out = dict(value=batch / dt, unit="complexes/s", cores=cores, kind="port",
               sample="oracle stack %s, B=%d, %d/%d nodes, hidden %d, %d layers, n_iter=%d, fp32, %d timed run(s) after one warm-up"
                      % ("forward + backward" if backward else "forward", batch, n_prot, n_lig, hidden, layers, n_iter, reps))
    full = os.path.join(ROOT, "profiles", "r04_cpu_baseline.json")
    if os.path.exists(full):
        try:
            out["protocol_8d"] = dict(json.load(open(full)), source="profiles/r04_cpu_baseline.json (committed run of `bench.py --cpu-baseline-full` on a GPU box's host, not this run)")
        except Exception:
            pass
    return out


def cpu_baseline_config3_full(hidden, layers, n_lig, warmups=3, runs=5):
    """SURVEY 8(d)'s CPU-baseline protocol for the HEADLINE workload of round 6 (the full IaBNet + six-term loss, forward + backward through the
    oracle's autograd): B = 1, fp32, eval; 32 threads (>= 3 warm-ups + 5 timed runs) and 1 thread (1 warm-up + 2 runs) at 1500 / 40 with the
    whole-protein pocket (config 3 read literally, n_iter 1); the production configuration (20 A pocket crop, n_iter 8) at 32 threads."""
    rows = []
    for threads, n_prot, n_iter, radius, w, r in ((32, 1500, 1, 1e9, warmups, runs), (1, 1500, 1, 1e9, 1, 2), (32, 1500, 8, 20.0, warmups, runs)):
        run, _ = _config3_run_fn(hidden, layers, n_iter, n_prot, n_lig, 1, threads, radius)
        for _ in range(w):
            run()
        ts = []
        for _ in range(r):
            t0 = time.time()
            run()
            ts.append(time.time() - t0)
        ts.sort()
        rows.append(dict(threads=threads, n_iter=n_iter, nodes="%d/%d" % (n_prot, n_lig), pocket="whole protein" if radius > 1e6 else "20 A crop",
                         workload="oracle full IaBNet + six-term loss, forward + backward", batch=1,
                         complexes_per_s=1.0 / (sum(ts) / len(ts)), best_run_s=ts[0], worst_run_s=ts[-1],
                         protocol="%d warm-up(s) + %d timed runs" % (w, r)))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    return dict(kind="port", host_cpus=os.cpu_count(), rows=rows)


def _config3_run_fn(hidden, layers, n_iter, n_prot, n_lig, batch, threads, pocket_radius=1e9):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fabind_oracle as orc
    from fabind_amd import synthetic
    from fabind_amd.models import get_model

    class _Log:
        def log_message(self, m):
            pass
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = synthetic.condition_for_large_graphs(get_model(stack_args(hidden, layers, n_iter), _Log(), None))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    cfg = dict(orc.DEFAULT_CFG)
    cfg.update(mean_layers=layers, n_iter=n_iter)
    data = synthetic.make_hetero_batch([(n_prot, n_lig)] * batch, seed=0, pocket_radius=pocket_radius)

    def run():
        for v in sd.values():
            v.grad = None
        out = orc.model_forward(sd, cfg, data.clone(), stage=1)
        loss, _ = orc.compute_loss(out, data)
        loss.backward()
    return run, sd


def cpu_baseline_config3(hidden, layers, n_iter, n_prot, n_lig, budget_s=25.0, batch=1):
    """The headline's workload on the host: the oracle's full IaBNet (`oracle/fabind_oracle.py::model_forward`, pocket radius unbounded)
    + the six-term loss + backward to every parameter through autograd, `batch` complexes of n_prot / n_lig nodes, fp32, 32 threads
    (the best setting of tools/probes/cpu_threads.py).  Bounded sample: one warm-up, then as many timed runs as fit `budget_s`."""
    default_threads = torch.get_num_threads()
    run, _ = _config3_run_fn(hidden, layers, n_iter, n_prot, n_lig, batch, min(default_threads, 32))
    t0 = time.time()
    run()
    first = time.time() - t0
    reps = max(1, min(5, int(budget_s / max(first, 1e-3)) - 1))
    t0 = time.time()
    for _ in range(reps):
        run()
    dt = (time.time() - t0) / reps
    cores = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    out = dict(value=batch / dt, unit="complexes/s", cores=cores, kind="port",
               sample="oracle full IaBNet (whole-protein pocket) + six-term loss, forward + backward, B=%d, %d/%d nodes, hidden %d, %d layers, "
                      "n_iter=%d, fp32, %d timed run(s) after one warm-up" % (batch, n_prot, n_lig, hidden, layers, n_iter, reps))
    full = os.path.join(ROOT, "profiles", "r06_cpu_baseline.json")
    if os.path.exists(full):
        try:
            out["protocol_8d"] = dict(json.load(open(full)), source="profiles/r06_cpu_baseline.json (committed run of `bench.py --cpu-baseline-full` on a GPU box's host, not this run)")
        except Exception:
            pass
    return out


def cpu_baseline_full(hidden, layers, n_lig, warmups=3, runs=5, batch=2, slow_run_s=40.0, skip_run_s=150.0):
    """SURVEY 8(d)'s CPU-baseline protocol: the oracle at B = 2, fp32, eval: torch.set_num_threads(k) for k = the best many-thread
    setting (32; torch's default of half the host's CPUs is measured next to it on the stack pass) and k = 1; >= 3 warm-ups + 5 timed
    runs; complexes/s per stack pass (n_iter 1) and per full forward (n_iter 8), at the headline shape (1500 / 40) and at the
    pocket-sized one (100 / 40); plus the forward + backward pass the headline times.  A configuration whose first run exceeds
    `slow_run_s` seconds is timed with 1 warm-up + 3 runs, one whose first run exceeds `skip_run_s` is recorded from that single run
    (both said in its `protocol`): the whole protocol has to fit a GPU box's time limit."""
    default_threads = torch.get_num_threads()
    rows = []
    for n_prot in (1500, 100):
        for threads in (32, default_threads, 1):
            for n_iter, backward in ((1, False), (8, False), (1, True)):
                if threads == default_threads and (n_iter != 1 or backward or n_prot != 1500):
                    continue                               # (the default thread count is a calibration point: stack pass, headline shape)
                if threads == 1 and backward:
                    continue
                torch.set_num_threads(threads)
                run = _cpu_run_fn(hidden, layers, n_iter, n_prot, n_lig, batch, backward)
                t0 = time.time()
                run()
                first = time.time() - t0
                if first > skip_run_s:
                    ts, proto = [first], "ONE run (%.0f s; no warm-up)" % first
                else:
                    w, r = (warmups, runs) if first <= slow_run_s else (1, 3)
                    for _ in range(w - 1):
                        run()
                    ts = []
                    for _ in range(r):
                        t0 = time.time()
                        run()
                        ts.append(time.time() - t0)
                    ts.sort()
                    proto = "%d warm-up(s) + %d timed runs" % (w, r)
                rows.append(dict(threads=threads, n_iter=n_iter, backward=backward, batch=batch, nodes="%d/%d" % (n_prot, n_lig),
                                 complexes_per_s=batch / (sum(ts) / len(ts)), best_run_s=ts[0], worst_run_s=ts[-1], protocol=proto))
                print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    torch.set_num_threads(default_threads)
    return dict(kind="port", host_cpus=os.cpu_count(), torch_default_threads=default_threads, rows=rows)


def north_star_targets(value, world, fwd_value=None, stack_value=None):
    """BASELINE.json's north_star targets next to where this build stands, said plainly.  The utilisation / bandwidth
    figures are NOT measured in this run: they are the committed rocprofv3 PMC / kernel-trace numbers of the same command
    (profiles/r06_pmc_util.txt, r06_pmc.json, r06_fwdbwd_kernel_stats.txt)."""
    return {
        "fwd_bwd_complexes_per_s_8gpu": {"target": 2000.0, "this_run": value, "n_gpus": world,
                                         "note": "target quoted for 8 x MI355X; %s" % ("this run is one GPU: 8 x this value = %.0f if scaling were perfect (no 8-GPU "
                                                                                      "node was available to measure it)" % (8.0 * value) if world == 1 else "measured here")},
        "fwd_complexes_per_s_8gpu": {"target": 10000.0, "this_run_stack_forward_one_gpu": fwd_value,
                                     "note": "north_star's forward target is only reachable for one stack pass per complex (SURVEY 8(d)); the `fwd` sub-object is that "
                                             "pass on one GPU; the stack-only fwd+bwd step (`stack_fwdbwd`) reads %s on one GPU" % (("%.0f /s" % stack_value) if stack_value else "n/a")},
        "mfma_util_cross_attention": {"target": 0.30, "measured_fwd": 0.165, "measured_bwd": 0.08, "met": False,
                                      "why": "the fused kernels' matrix-core busy time is bounded by the block's node-level I/O: per 64-row protein tile 22 MFLOP "
                                             "against 160 KiB of fp32 q|gate rows, bf16 a0 tile and output -- 25 % at a perfect HBM floor, 17 % with the "
                                             "projections inside (DESIGN 6.3); the backward recomputes the bias contraction twice",
                                      "source": "profiles/r06_pmc_util.txt (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, cross_attn_fused_*_kernel<512>)"},
        "hbm_frac_message_passing": {"target": 0.40,
                                     "measured": {"segment_sum (sending-side aggregation, 1.60 GB / 483 us)": 0.41,
                                                  "block_hadamard_bwd (3.93 GB / 0.92 ms)": 0.53, "rowdot_bwd (7.86 GB / 1.74 ms)": 0.56,
                                                  "block_hadamard_fwd (3.93 GB written / 1.24 ms)": 0.40},
                                     "met": "for the reductions; the fused edge kernels are dense contractions (SURVEY 8(d): MFMA-bound class) at 0.20 (backward) / "
                                            "0.31 (forward) of the bf16 peak",
                                     "source": "profiles/r06_pmc.json, r06_fwdbwd_kernel_stats.txt"},
        "parity": {"target": "ligand RMSD <= 1e-4 A, losses <= 1e-5 relative vs the reference CPU path",
                   "this_dtype": "1.6e-5 A / 2.0e-7 on this workload (tests/test_gpu_production.py::test_config3_whole_graph_matches_oracle, asserted at the gates)"},
    }


def self_launch(n):
    """--gpus N with no launcher: spawn the N ranks as a child process group (one process per GPU over RCCL) BEFORE anything in
    this process initialises the GPU; relay the child's output and exit code.  (Never re-exec a GPU-initialised process.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: required for RCCL on this image
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="complexes per GPU")
    ap.add_argument("--n-prot", type=int, default=1500)
    ap.add_argument("--n-lig", type=int, default=40)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--n-iter", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="bf16: the dtype BASELINE configs[1..2] name; fp32: exact-fp32 MFMA; bf16x3: fp32 storage with split-bf16 "
                         "contractions (three bf16 MFMAs per product term) -- the fast mode that meets the 1e-4 A parity gate")
    ap.add_argument("--poses", type=int, default=20, help="plus_sampling: poses sampled per complex and step")
    ap.add_argument("--mode", default="config3", choices=["config3", "fwd", "fwdbwd", "model", "plus_sampling", "plus_train"],
                    help="config3 (default, the headline): BASELINE configs[2] read literally = `model --whole-pocket`: the full IaBNet with "
                         "the complex model and the distance-map head on ALL 1500 / 40 nodes, six-term loss, fwd+bwd; "
                         "fwd / fwdbwd: the layer stack alone on the whole graph (SURVEY 8(d); rounds 1-5's headline); model: the full "
                         "IaBNet (pocket model on the whole protein -> pocket crop -> complex model -> heads) with the "
                         "pocket-cls + coord + distmap losses, fwd+bwd (BASELINE configs[2] read literally); plus_sampling: "
                         "FABind+ sampling-mode inference (BASELINE configs[4]: dropout sampling, DBSCAN centre choice, "
                         "confidence head, --poses per complex); plus_train: one FABind+ training step (train mode, 7-term "
                         "loss incl. the permutation-invariant coordinate term, fwd+bwd)")
    ap.add_argument("--whole-pocket", action="store_true",
                    help="--mode model: unbounded pocket radius -- the complex model and the heads see the whole 1500 / 40 graph (BASELINE "
                         "configs[2] read literally; the `config3_whole_graph` sub-object of the default line)")
    ap.add_argument("--train-mode", action="store_true", help="fwdbwd / model: model.train() (dropout p = 0.1 at the reference's sites)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="run ONLY SURVEY 8(d)'s CPU-baseline protocol (oracle, B=2, >=3 warm-ups + 5 timed runs, 32 threads / torch default / 1 "
                         "thread, stack pass and 8-iteration forward) on the host and print its JSON; no GPU work (tens of minutes)")
    ap.add_argument("--no-extras", action="store_true",
                    help="default fwdbwd mode at N=1 also times, inside the same JSON line, the same step in fp32 (the mode that "
                         "meets the 1e-4 A gate), in train mode (dropout on), at n_iter=8, forward-only, and the full IaBNet with "
                         "the six-term loss; this flag skips those sub-objects")
    a = ap.parse_args()

    if a.cpu_baseline_full:
        res = cpu_baseline_config3_full(a.hidden, a.layers, a.n_lig)
        if os.environ.get("FABIND_CPU_BASELINE_STACK", "0") == "1":          # + the stack-only protocol of rounds 4-5 (tens of minutes)
            res["stack_protocol"] = cpu_baseline_full(a.hidden, a.layers, a.n_lig)
        print(json.dumps(res))
        return
    headline_config3 = a.mode == "config3"
    if headline_config3:
        a.mode, a.whole_pocket = "model", True
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                         "run `python bench.py --gpus %d` without a launcher)" % (a.gpus, world, a.gpus, a.gpus))
    # test hooks (one-GPU boxes): FABIND_BENCH_DEVICE pins every rank to one device, FABIND_BENCH_BACKEND=gloo swaps RCCL
    # for gloo (gradients are then staged through the host) so that the N>1 control flow can be exercised anywhere
    local = int(os.environ.get("FABIND_BENCH_DEVICE", local))
    backend = os.environ.get("FABIND_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from fabind_amd import engine
    from fabind_amd import kernels as K
    engine.set_precision(a.precision)
    # backward's Python (the adjoint kernels' launch code) runs in the CALLING thread: autograd's default hands every backward pass to a
    # per-device worker thread, and the two Python threads then trade the interpreter lock launch by launch.  Same-box pairs (round 4,
    # profiles/r04_ab_same_box.txt): pocket-sized step +8 ... +19 %, full model +1 %, headline +0.7 %.  A process-level torch setting any
    # training script can make; FABIND_BENCH_ST_BACKWARD=0 restores torch's default for A/B runs.
    if os.environ.get("FABIND_BENCH_ST_BACKWARD", "1") == "1":
        torch.autograd.set_multithreading_enabled(False)

    class _Log:
        def log_message(self, m):
            pass

    def make_step(mode, n_iter, train_mode=False, n_prot=None, whole_pocket=False):
        """-> (step function, complexes per step on this rank, parameter list).  Inputs are resident in HBM before the timed region.
        n_prot: protein nodes per complex (default --n-prot); whole_pocket (mode "model"): the pocket radius is unbounded, so the
        4-layer complex model and the distance-map head see the WHOLE 1500 / 40 graph -- BASELINE configs[2] read literally."""
        n_prot = a.n_prot if n_prot is None else n_prot
        if mode in ("plus_sampling", "plus_train"):
            from fabind_amd import synthetic
            from fabind_amd.plus.models import get_model as get_model_plus
            margs = stack_args(a.hidden, a.layers if a.layers != 4 else 5, n_iter)
            for k_, v_ in dict(use_ln_mlp=True, mlp_hidden_scale=1, dropout=0.1, mha_heads=4, rel_dis_pair_bias="no",
                               inter_additional_mlp=False, only_last_LAS=False, geom_reg_steps=1, use_for_radius_pred="ligand",
                               dis_map_thres=15.0, pocket_radius_buffer=5.0, min_pocket_radius=20.0, force_fix_radius=False,
                               use_clustering=True, dbscan_eps=9.0, dbscan_min_samples=2, choose_cluster_prob=0.5,
                               confidence_training=True, stack_mlp=True, confidence_use_ln_mlp=True, confidence_dropout=0.2,
                               confidence_mlp_hidden_scale=1).items():
                setattr(margs, k_, v_)
            if mode == "plus_train":
                from fabind_amd.plus.models import compute_loss as plus_loss
                margs.confidence_training, margs.use_clustering = False, False
                torch.manual_seed(0)
                model = get_model_plus(margs, _Log()).to(dev)        # LN-MLPs normalise every block: plain init is well conditioned
                model.train()
                hb = synthetic.make_hetero_batch([(n_prot, a.n_lig)] * a.batch, seed=rank).to(dev)
                radius = torch.full((a.batch,), 6.0, device=dev)
                num_atoms = [a.n_lig] * a.batch
                isos = [[list(range(a.n_lig)), list(reversed(range(a.n_lig)))] for _ in range(a.batch)]
                params = list(model.parameters())
                reducer = None
                if world > 1:
                    from fabind_amd import parallel
                    reducer = parallel.GradReducer(params, world)

                def step():
                    for p in params:
                        p.grad = None
                    data = hb.clone()
                    data.ligand_radius, data.num_atoms, data.isomorphisms = radius, num_atoms, isos
                    out = model(data, train=True)
                    loss, _ = plus_loss(out, data)
                    loss.backward()
                    if reducer is not None:
                        reducer.finish()
                return step, a.batch, params
            torch.manual_seed(0)
            model = get_model_plus(margs, _Log()).to(dev)
            model.train()                                              # --infer-dropout: dropout on, ranking head in eval
            for name_, sub_ in model.named_modules():
                if name_.startswith("confidence") or name_.startswith("ranking"):
                    sub_.eval()
            hb = synthetic.make_hetero_batch([(n_prot, a.n_lig)] * a.batch, seed=rank).to(dev)

            def step():
                for _ in range(a.poses):
                    model.inference(hb.clone())
            return step, a.batch, []
        if mode == "model":
            from fabind_amd import synthetic
            from fabind_amd.models import get_model
            from fabind_amd.models.model import compute_loss
            margs = stack_args(a.hidden, a.layers, n_iter)
            torch.manual_seed(0)
            model = get_model(margs, _Log(), dev).to(dev)
            synthetic.condition_for_large_graphs(model)
            model.train(train_mode)
            hb = synthetic.make_hetero_batch([(n_prot, a.n_lig)] * a.batch, seed=rank,
                                             **({"pocket_radius": 1e9} if whole_pocket else {})).to(dev)
            params = list(model.parameters())
            reducer = None
            if world > 1:
                # the overlapped reducer: this is the mode with the 33 tensors a step never reaches -- its discovery
                # step learns them and from the second step on every bucket but the trailing one leaves during backward
                from fabind_amd import parallel
                reducer = parallel.GradReducer(params, world)
            # The batch of step k + 1 "arrives" on a feeder stream while step k runs (what fabind_amd.data.DeviceFeeder does for a real
            # loader) and model.plan_stage1 builds, there, everything the forward would otherwise read back from the device in the middle
            # of the step: gather indices, pair lists, both stack models' layouts and input-coordinate graphs (~18 queue drains per
            # step).  All of that work still happens every step, inside the timed region.  FABIND_BENCH_PLAN=0: no plan (A/B).
            use_plan = os.environ.get("FABIND_BENCH_PLAN", "1") == "1"
            feeder = torch.cuda.Stream(dev) if use_plan else None
            pending = []

            def arrive():
                if feeder is None:
                    return hb.clone(), None
                with torch.cuda.stream(feeder):
                    data = hb.clone()
                    plan = model.plan_stage1(data)
                return data, plan

            def step():
                for p in params:
                    p.grad = None
                if not pending:
                    pending.append(arrive())
                data, plan = pending.pop()
                if plan is not None:
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(plan["event"])
                    for st_ in list(data._stores.values()) + [data._glob]:       # tensors made on the feeder stream, used on this one
                        for v_ in st_.values():
                            if torch.is_tensor(v_) and v_.is_cuda:
                                v_.record_stream(cur)
                out = model(data, stage=1, train=train_mode, plan=plan)
                loss, _ = compute_loss(out, data)
                loss.backward()
                if reducer is not None:
                    reducer.finish()
                if feeder is not None:
                    pending.append(arrive())                 # the next batch arrives while this step's kernels are still queued
            return step, a.batch, params
        model = build_model(a.hidden, a.layers, n_iter, dropout=0.1 if train_mode else 0.0).to(dev)
        model.train(train_mode)
        # two DISTINCT resident batches, served alternately, each time as fresh tensor objects for the index vectors: like a batch
        # arriving from a data loader, every step pays the per-batch layout construction (two host syncs + index assembly) --
        # the layout cache of engine.Layout.of only helps callers that re-run the very same batch object
        n_res = 1 if LEGACY_BATCH else 2
        batches = []
        for r_ in range(n_res):
            inp = make_batch(a.batch, n_prot, a.n_lig, a.hidden, seed=rank + 1000 * r_)
            batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()})
        counter = [0]
        params = list(model.parameters())
        reducer = None
        if world > 1 and mode == "fwdbwd":
            from fabind_amd import parallel
            reducer = parallel.GradReducer(params, world)           # all-reduce buckets overlap the rest of backward

        # The batch of step k + 1 "arrives" while step k runs, the way a data feeder delivers it (fabind_amd.data.DeviceFeeder works the
        # same way): fresh tensor objects made on a feeder stream, and engine.prefetch builds its layout and input-coordinate graph
        # there, so their host round trips do not drain the compute stream at the start of the step.  Every step still does all of
        # that work, inside the timed region.
        from fabind_amd import engine as _engine
        # Measured same-box.  Round 3 (profiles/r03_ab_same_box.txt): with the read-backs gone from the middle of the step the prefetch paid
        # at the headline shape (fwd+bwd 688.1 / 703.5 / 692.2 -> 715.5 / 711.2 / 718.1) and was a coin flip at the pocket-sized one, whose
        # step was host-bound.  Round 4 (profiles/r04_ab_same_box.txt): the pocket-sized step is device-bound once a batch's layout and
        # graph exist (15.97 ms = 4,007 complexes/s with ONE resident batch), and the step-start read-backs are what a fresh batch costs:
        # 3,042 / 3,129 -> 3,474 / 3,235 complexes/s with the prefetch.  Default: on for the stack modes; FABIND_BENCH_PREFETCH=1 / 0 forces it.
        want = (mode in ("fwd", "fwdbwd")) if PREFETCH is None else PREFETCH
        feeder = torch.cuda.Stream(dev) if (want and not LEGACY_BATCH and not REUSE_BATCH) else None
        prefetching[0] = feeder is not None
        pending = []

        def arrive():
            t = dict(batches[0 if REUSE_BATCH else counter[0] % n_res])
            counter[0] += 1
            if feeder is None:
                if not LEGACY_BATCH and not REUSE_BATCH:
                    t["batch_id"], t["segment_id"] = t["batch_id"].clone(), t["segment_id"].clone()
                return t, t["X"].clone(), None
            with torch.cuda.stream(feeder):
                t["batch_id"], t["segment_id"] = t["batch_id"].clone(), t["segment_id"].clone()
                X0 = t["X"].clone()
                with torch.no_grad():
                    _engine.prefetch(model, X0, t["batch_id"], t["segment_id"], t["compound_edge_index"])
                ev = torch.cuda.Event()
                ev.record(feeder)
            return t, X0, ev

        def step():
            if feeder is None:
                t, X0, _ = arrive()
            else:
                if not pending:
                    pending.append(arrive())
                t, X0, ev = pending.pop()
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for v in (X0, t["batch_id"], t["segment_id"]):
                    v.record_stream(cur)
            if mode == "fwd":
                with torch.no_grad():
                    model(X0, t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"], t["compound_edge_index"],
                          t["LAS_edge_index"], t["coord_LAS"])
            else:
                for p in params:
                    p.grad = None
                X, Hh = model(X0, t["H"], t["batch_id"], t["segment_id"], t["mask"], t["is_global"],
                              t["compound_edge_index"], t["LAS_edge_index"], t["coord_LAS"])
                loss = (X * X).mean() + (Hh * Hh).mean() * 1e-6
                loss.backward()
                if reducer is not None:
                    reducer.finish()
            if feeder is not None:
                pending.append(arrive())                     # the next batch arrives while this step's kernels are still queued
        return step, a.batch, params

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def aten_census(step, path):
        """Diagnostic (FABIND_BENCH_ATEN=<file>): one extra untimed step under torch.profiler; every torch (aten) op that launches something
        itself, by shapes and issuing site (the innermost fabind_amd frame, or the autograd node for backward ops), sorted by device time."""
        import collections
        from torch.profiler import ProfilerActivity, profile as tprofile
        torch.autograd.set_multithreading_enabled(False)
        with tprofile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            step()
            torch.cuda.synchronize()
        torch.autograd.set_multithreading_enabled(True)
        rows = collections.defaultdict(lambda: [0, 0.0])
        for e in prof.events():
            sdt = getattr(e, "self_device_time_total", 0) or 0
            if not e.name.startswith("aten::") or sdt <= 0:
                continue
            site = "?"
            for fr in (e.stack or []):
                if "fabind_amd" in fr or "bench.py" in fr:
                    site = fr.split("fabind_amd/")[-1][:80]
                    break
            key = (e.name, str(e.input_shapes)[:90], site)
            rows[key][0] += 1
            rows[key][1] += sdt
        tot = sum(v[1] for v in rows.values())
        with open(path, "w") as f:
            f.write("torch (aten) ops with device time of their own in one step: %d launches, %.2f ms\n" % (sum(v[0] for v in rows.values()), tot / 1e3))
            for (name, shapes, site), (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:120]:
                f.write("%9.1f us %4d x  %-28s %-90s %s\n" % (us, cnt, name, shapes, site))

    def timed(step, warmup, steps, profile, events_in_timed=True):
        """Timing protocol of the contract (warm-up, barrier + synchronize, exactly `steps` steps, barrier + synchronize).  Live HIP
        events for the roofline: the LAST warm-up step times every labelled launch and names the dominant kernel family; the timed
        region then times that family's launches only (all of them, every step) -- events around every launch of the step cost the
        host 2 % of the headline step and 6-11 % of the pocket-sized one (same-box pairs in profiles/r03_ab_same_box.txt).
        FABIND_BENCH_DUMP_PROFILE keeps the events of every launch (the launch-group listing needs them).
        events_in_timed=False (the sub-objects, several of which are host-bound): the timed region carries NO events at all; the returned
        profile is the fully timed warm-up step (every labelled launch of ONE step)."""
        probe = profile and not os.environ.get("FABIND_BENCH_DUMP_PROFILE")      # (the same on every rank: sync() holds a barrier)
        probe_prof, probe_bytes, probe_algo = None, {}, {}
        profile = profile and rank == 0
        only = None
        for w in range(warmup):
            if probe and w == warmup - 1:
                sync()
                K.PROFILE, K.PROFILE_ONLY = ({} if profile else None), None
                K.PROFILE_BYTES.clear()
                K.PROFILE_ALGO.clear()
                step()
                sync()
                fams = {}
                for label, evs in (K.PROFILE or {}).items():
                    fams[family(label)] = fams.get(family(label), 0.0) + sum(s_.elapsed_time(e_) for s_, e_, _ in evs)
                probe_prof, probe_bytes, probe_algo = K.PROFILE, dict(K.PROFILE_BYTES), dict(K.PROFILE_ALGO)
                K.PROFILE = None
                if fams:
                    only = max(fams.items(), key=lambda kv: kv[1])[0].split(" ")[0]
            else:
                step()
        sync()
        if os.environ.get("FABIND_BENCH_ATEN") and rank == 0:
            aten_census(step, os.environ["FABIND_BENCH_ATEN"])
            sync()
        K.PROFILE = {} if (profile and events_in_timed) else None
        K.PROFILE_BYTES.clear()
        K.PROFILE_ALGO.clear()
        K.PROFILE_ONLY = only
        t0 = time.time()
        for _ in range(steps):
            step()
        sync()
        dt = time.time() - t0
        prof = K.PROFILE
        K.PROFILE, K.PROFILE_ONLY = None, None
        if profile and not events_in_timed and probe_prof:
            prof = probe_prof
            K.PROFILE_BYTES.clear()
            K.PROFILE_BYTES.update(probe_bytes)
            K.PROFILE_ALGO.clear()
            K.PROFILE_ALGO.update(probe_algo)
        if world > 1:
            tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, prof

    def family(label):
        """Kernel family of a profiled launch label: the kernel's name; GEMMs are split by where their rows come from (node /
        ligand / pair / edge level differ by orders of magnitude in M) only through the name prefix, never by padded shape."""
        name = label.split(" ")[0]
        if name == "fabind_gemm" and "ragged" in label:
            return "fabind_gemm (ragged groups)"
        return name

    def roofline_of(prof, dt, precision):
        """Dominant kernel FAMILY of the timed region from live HIP-event timing of every launch (kernels._profiled), priced as SURVEY 8(d)
        prescribes: `bound` from the operation's classification (bound_of), `achieved` = (executed flops | 8(d) compulsory bytes of the
        family's launches) / (sum of their durations); a ragged launch carries the flops of its actual group sizes."""
        fams = {}
        for label, evs in prof.items():
            f = fams.setdefault(family(label), [0.0, 0, 0.0, label, 0.0, 0.0])
            for s_, e_, fl in evs:
                f[0] += s_.elapsed_time(e_)
                f[1] += 1
                f[2] += fl
            f[4] += K.PROFILE_BYTES.get(label, 0.0)
            f[5] += K.PROFILE_ALGO.get(label, 0.0)
        name, (ms, cnt, flops, label, nbytes, algo) = max(fams.items(), key=lambda kv: kv[1][0])
        out_ = price_against_rooflines(flops, algo, ms, precision, bound_of(name), design_bytes=nbytes)
        out_.update({"traffic": None,
                     "kernel": label if len([1 for l_ in prof if family(l_) == name]) == 1 else name + " (all shapes of the step)",
                     "flop_per_launch": flops / cnt, "launches": cnt, "avg_us": 1e3 * ms / cnt, "share_of_step": ms / (1e3 * dt),
                     "algorithmic_bytes_per_launch": algo / cnt, "design_bytes_per_launch": nbytes / cnt,
                     "classification": "SURVEY 8(d): %s" % ("dense contraction, reported against the matrix-core peak on executed flops"
                                                            if bound_of(name) == "mfma" else
                                                            "stand-alone gather / scatter / reduction, reported against HBM on its compulsory bytes")})
        return out_

    step, per_rank, _ = make_step(a.mode, a.n_iter, train_mode=a.train_mode, whole_pocket=a.whole_pocket)
    dt, prof = timed(step, a.warmup, a.steps, os.environ.get("FABIND_BENCH_NO_PROFILE", "0") != "1")   # (=1: no per-launch events, A/B of their cost)
    poses = a.poses if a.mode == "plus_sampling" else 1
    value = per_rank * world * a.steps / dt * poses
    out = None
    if rank == 0:
        out = {
            "metric": ("poses/sec, FABind+ sampling-mode inference (dropout sampling + DBSCAN centre + confidence head)"
                       if a.mode == "plus_sampling" else
                       "complexes/sec fwd+bwd (1500p/40l nodes), full IaBNet on the whole graph (pocket model + 4-layer complex model + "
                       "distance-map head on all nodes / pairs) with the pocket-cls + pocket-centre + coord + distmap + distmap-by-coords + "
                       "distill losses (BASELINE configs[2] read literally)" if (a.mode == "model" and a.whole_pocket) else
                       "complexes/sec fwd+bwd, full IaBNet (pocket model + pocket crop + complex model + heads) with "
                       "pocket-cls + coord + distmap losses" if a.mode == "model" else
                       "complexes/sec, one FABind+ training step (train mode, 7-term loss with the permutation-invariant "
                       "coordinate term, fwd+bwd)" if a.mode == "plus_train" else
                       "complexes/sec %s (1500p/40l nodes), one stack pass per refinement iteration" % (
                           "fwd+bwd" if a.mode == "fwdbwd" else "fwd")),
            "value": value, "unit": "poses/s" if a.mode == "plus_sampling" else "complexes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            **({"dtype_note": "bf16 MFMA operands, fp32 accumulation and residual stream; the edge pipeline, the attention tiles and every backward "
                              "contraction on bf16 operands; ~40 of the ~50 node-level GEMMs of a FORWARD pass (input Linear, inter-edge v and "
                              "coordinate projections, attention k|v and output projections, both Linears of every node MLP / Transition with an "
                              "fp32 hidden layer) contract in split bf16 (3 MFMAs per product term): config.set_split_sites(3).  Ligand RMSD from "
                              "the fp32 oracle: 1.6e-5 A on this workload (config 3), 1.4e-5 / 7.8e-5 A for the stack at n_iter 1 / 8"}
               if a.precision == "bf16" else {}),
            "config": {"workload": ("synthetic batch=%d/GPU, %d protein / %d ligand nodes, FABind+ model (5-layer LN-MLP stack, hidden "
                                    "%d, n_iter=%d), %d poses per complex and step" % (a.batch, a.n_prot, a.n_lig, a.hidden,
                                                                                        a.n_iter, a.poses))
                       if a.mode == "plus_sampling" else
                       "synthetic batch=%d/GPU, %d protein / %d ligand nodes, FABind+ model (5-layer LN-MLP stack, hidden %d, "
                       "n_iter=%d), training step" % (a.batch, a.n_prot, a.n_lig, a.hidden, a.n_iter)
                       if a.mode == "plus_train" else
                       "BASELINE configs[2]: synthetic batch=%d/GPU (%d distinct seeded geometries), %d protein / %d ligand nodes, full IaBNet "
                       "(hidden-128 pocket model + %d-layer hidden-%d FABind stack + out layer + distance-map head), %s, n_iter=%d, "
                       "six-term loss, fwd+bwd" % (a.batch, a.batch, a.n_prot, a.n_lig, a.layers, a.hidden,
                                                   "pocket = the whole protein (radius unbounded): every node and all %d pairs per complex"
                                                   % (a.n_prot * a.n_lig) if a.whole_pocket else "20 A pocket crop", a.n_iter)
                       if a.mode == "model" else
                       "synthetic batch=%d/GPU (%d distinct seeded geometries), %d protein / %d ligand nodes, %d-layer FABind stack "
                       "+ out layer, hidden %d, n_iter=%d, %s" % (a.batch, a.batch, a.n_prot, a.n_lig, a.layers, a.hidden,
                                                                   a.n_iter, a.mode),
                       "global_batch": a.batch * world, "n_iter": a.n_iter, "pass": a.mode,
                       **({"train_mode": True} if a.train_mode else {}),
                       **({"loss": "stack modes: a fixed quadratic of the stack's two outputs, (X*X).mean() + 1e-6 (H*H).mean() -- every parameter "
                                   "the training step reaches gets a gradient; the reference's pocket-cls + coord + distmap losses need the "
                                   "full IaBNet around the stack: that is the `config3_whole_graph` sub-object (BASELINE configs[2] read "
                                   "literally: complex model + heads on all 1500 / 40 nodes, six-term loss) and `model_fwdbwd` (the production "
                                   "pocket crop)",
                           "config3": "config3_whole_graph sub-object of this line (parity: tests/test_gpu_production.py::"
                                      "test_config3_whole_graph_matches_oracle)"} if a.mode in ("fwd", "fwdbwd") else {}),
                       **({"loss": "the reference's six-term training loss (main_fabind.py:398-417; fabind_amd.models.model.compute_loss)",
                           "parity": "tests/test_gpu_production.py::test_config3_whole_graph_matches_oracle (fp32 vs the CPU oracle: 11-tuple, six "
                                     "loss terms, 351 gradients; the bench dtype asserted at the 1e-4 A gate)",
                           "stack_only": "`stack_fwdbwd` sub-object = rounds 1-5's headline (the layer stack with a quadratic loss)"}
                          if (a.mode == "model" and a.whole_pocket) else {}),
                       "batch_arrival": ("a fresh batch object per step; its stage-1 plan (gather indices, pair lists, both stacks' layouts and "
                                         "input graphs) built on a feeder stream during the previous step (model.plan_stage1)") if a.mode == "model" else
                                        ("fresh index tensors per step; layout + input-coordinate graph of step k+1 built on a "
                                         "feeder stream during step k (engine.prefetch)") if prefetching[0]
                       else ("ONE resident batch re-served (round 2's protocol, FABIND_BENCH_REUSE_BATCH=1)" if REUSE_BATCH else
                             "fresh index tensors per step, layout built at the start of the step")},
        }
        if prof and os.environ.get("FABIND_BENCH_DUMP_PROFILE"):
            # development aid: live HIP-event time of every profiled launch group (GEMM shapes, fused kernels) in the timed region
            rows = sorted(((sum(s_.elapsed_time(e_) for s_, e_, _ in evs), len(evs), evs[0][2], label) for label, evs in prof.items()),
                          reverse=True)
            with open(os.environ["FABIND_BENCH_DUMP_PROFILE"], "w") as f:
                for ms, cnt, fl, label in rows:
                    f.write("%9.3f ms/step  %4d launches/step  %8.1f us avg  %7.1f TFLOP/s  %s\n" % (
                        ms / a.steps, cnt // a.steps, 1e3 * ms / cnt, fl / (ms / cnt * 1e-3) / 1e12 if ms > 0 else 0.0, label))
        if prof:
            out["roofline"] = roofline_of(prof, dt, a.precision)
            # HBM traffic of the dominant kernel: NOT measured in this run -- the per-launch figure of the committed PMC passes
            # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per MI355X_MICROARCH.md "HBM"); the source is named
            for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
                pmc = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc):
                    continue
                try:
                    tr = json.load(open(pmc))
                    for key, val in tr.get("per_kernel", {}).items():
                        if key in out["roofline"]["kernel"]:
                            out["roofline"]["traffic"] = val
                            out["roofline"]["traffic_source"] = "profiles/" + name + " (committed PMC pass, not this run)"
                except Exception:
                    pass
                if out["roofline"]["traffic"] is not None:
                    break

    # ---- the same JSON line also says what the neighbouring configurations cost (N=1, default workload only)
    if (a.mode == "fwdbwd" or headline_config3) and world == 1 and not a.no_extras:
        extras = {}
        del step
        torch.cuda.empty_cache()

        from fabind_amd import config as _config

        def sub(name, mode, n_iter, train_mode=False, precision=None, note="", steps=3, warmup=1, n_prot=None, whole_pocket=False,
                x3_backward="bf16", x3_edge="split"):
            only = os.environ.get("FABIND_BENCH_ONLY")             # development knob: comma-separated sub-object names to run
            if only and name not in only.split(","):
                return
            prec = precision or a.precision
            engine.set_precision(prec)
            _config.set_x3_backward(x3_backward)
            _config.set_x3_edge(x3_edge)
            try:
                st, per, _ = make_step(mode, n_iter, train_mode, n_prot=n_prot, whole_pocket=whole_pocket)
                d, pf = timed(st, warmup, steps, True, events_in_timed=False)
                mult = a.poses if mode == "plus_sampling" else 1
                o = {"value": per * steps * mult / d, "unit": "poses/s" if mode == "plus_sampling" else "complexes/s",
                     "ms_per_step": 1e3 * d / steps, "steps": steps, "warmup": warmup,
                     "dtype": prec, "pass": mode, "n_iter": n_iter, "train_mode": train_mode, "note": note}
                if n_prot is not None:
                    o["nodes"] = "%d protein / %d ligand" % (n_prot, a.n_lig)
                if pf:
                    rf = roofline_of(pf, d, prec)
                    o["roofline"] = {k_: rf[k_] for k_ in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_us")}
                    o["roofline"]["from"] = "the fully timed last warm-up step (no events inside this object's timed region)"
                extras[name] = o
            except Exception as e:                      # a sub-object must never cost the headline line
                extras[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            finally:
                engine.set_precision(a.precision)
                _config.set_x3_backward("bf16")
                _config.set_x3_edge("split")
                st = per = _ = pf = None          # (model, batches, events: nothing of this sub-object may live into the next one)
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        if headline_config3:
            sub("stack_fwdbwd", "fwdbwd", a.n_iter, steps=10, warmup=3,
                note="rounds 1-5's headline: the layer stack ALONE (4 FABind layers + out layer, hidden 512) on the same 1500 / 40 batch, "
                     "forward + backward of (X*X).mean() + 1e-6 (H*H).mean(), fresh batch per step")
        sub("pocket", "fwdbwd", a.n_iter, steps=30, warmup=5, n_prot=100,
            note="the headline step at the size the 4-layer stack sees in production (SURVEY 0.4 / 8(d): the 20 A pocket, 100 protein / 40 "
                 "ligand nodes per complex), fresh batch per step")
        sub("gate_mode", "fwdbwd", a.n_iter, precision="bf16x3", steps=10, warmup=2,
            note="the headline step in the split-bf16 mode (fp32 storage; forward and activation-gradient chain with three bf16 MFMAs per "
                 "product term): the FAST mode whose OUTPUTS meet the 1e-4 A parity gate at n_iter 1, 2 and 8 (tests/test_gpu_headline.py); "
                 "its weight gradients, fused edge backward and pair-bias adjoint run on bf16 roundings (bf16-grade parameter gradients)")
        sub("gate_mode_exact_bwd", "fwdbwd", a.n_iter, precision="bf16x3", steps=5, warmup=2, x3_backward="exact",
            note="gate_mode with config.set_x3_backward('exact'): weight gradients as split contractions, pair-bias adjoint in fp32, the "
                 "intra-graph edge pipeline unfused (fp32 edge tensors, split contractions in the forward, the input gradients and the weight "
                 "gradients): no bf16 rounding anywhere in the adjoint")
        sub("gate_mode_bf16_edge", "fwdbwd", a.n_iter, precision="bf16x3", steps=8, warmup=2, x3_edge="bf16",
            note="gate_mode with config.set_x3_edge('bf16'): the intra-graph edge pipeline on the bf16 kernels (forward 2.2 instead of 4.9 ms per "
                 "launch); the coordinate / loss gates still hold with less margin (ligand RMSD 3.5e-6 / 7.0e-6 / 3.0e-5 A at n_iter 1 / 2 / 8, full "
                 "IaBNet 5.1e-5 A: tests/test_gpu_headline.py, test_gpu_production.py), node features to 3.4e-4 only -- an option, not the default")
        sub("fp32", "fwdbwd", a.n_iter, precision="fp32",
            note="the headline step in fp32 mode (exact-fp32 MFMA, unfused edge pipeline): the exact reference arithmetic")
        sub("train_mode", "fwdbwd", a.n_iter, train_mode=True, steps=5, warmup=2,
            note="the headline step with model.train(): dropout p=0.1 at the reference's six sites (per-edge: in the fused edge kernels; "
                 "ahead of the residuals and after the input Linear: in the GEMM epilogues, masks regenerated by the adjoint)")
        sub("n_iter8", "fwdbwd", 8, note="production refinement loop: 8 stack passes, gradient on the last one; bf16 with split-precision sites "
                                           "level 3: 7.8e-5 A from the fp32 oracle at this shape -- the 1e-4 A gate is met and asserted "
                                           "(tests/test_gpu_headline.py); round 5 (level 2: 1.9e-4 A, gate missed) ran this loop at 276 /s",
            steps=5, warmup=1)
        sub("n_iter8_gate", "fwdbwd", 8, precision="bf16x3", steps=3, warmup=1,
            note="n_iter8 in the gate-meeting split-bf16 mode (2.8e-6 A at the headline shape, tests/test_gpu_headline.py)")
        sub("n_iter8_gate_bf16_edge", "fwdbwd", 8, precision="bf16x3", steps=3, warmup=1, x3_edge="bf16",
            note="n_iter8_gate with config.set_x3_edge('bf16') (3.0e-5 A at the headline shape: see gate_mode_bf16_edge)")
        sub("fwd", "fwd", a.n_iter, steps=10, warmup=3, note="forward only, one stack pass")
        sub("model_fwdbwd", "model", a.n_iter, steps=12, warmup=3,
            note="full IaBNet (pocket model on 1500 residues -> pocket crop -> 4-layer complex model -> heads) with the reference's "
                 "six-term loss (pocket-cls + pocket-centre + contact x2 + distill + coord), eval mode")
        sub("model_gate", "model", a.n_iter, precision="bf16x3", steps=10, warmup=2,
            note="model_fwdbwd in the gate-meeting split-bf16 mode (production-size parity: tests/test_gpu_production.py)")
        if not headline_config3:
            sub("config3_whole_graph", "model", a.n_iter, steps=10, warmup=2, whole_pocket=True,
                note="BASELINE configs[2] read literally: the same synthetic batch with an unbounded pocket radius, so the 4-layer hidden-512 "
                     "complex model AND the heads run on all 1500 protein / 40 ligand nodes -- pocket-cls + pocket-centre + coord + both "
                     "distance-map losses + distill (the reference's six-term loss) -- fwd+bwd, eval mode")
        else:
            sub("config3_gate", "model", a.n_iter, precision="bf16x3", steps=6, warmup=2, whole_pocket=True,
                note="the headline step in the split-bf16 mode (fp32 storage, three bf16 MFMAs per product term)")
        sub("model_fwdbwd_train_n_iter8", "model", 8, train_mode=True, precision="bf16x3",
            note="full IaBNet with model.train() (dropout, Gumbel noise) and n_iter=8: the reference's training configuration, in the "
                 "gate-meeting split-bf16 mode (production-size parity at n_iter 8: 2.3e-6 / 2.6e-6 A, tests/test_gpu_production.py)")
        sub("model_fwdbwd_train_n_iter8_bf16", "model", 8, train_mode=True,
            note="the same in bf16 (split sites level 3): production-size parity at n_iter 8 9.1e-5 A in stage 1 (gate met) but 1.3e-4 A in "
                 "stage 2 (gate MISSED by 1.3x) -- reported, not the creditable number")
        n_it8 = 8
        sub("plus_train", "plus_train", a.n_iter, steps=6, warmup=2,
            note="one FABind+ training step (5-layer LN-MLP stack, train mode, 7-term loss with the permutation-invariant term; round 5: the edge "
                 "MLP's first Linear with its LayerNorm folded into per-node projections under autograd)")
        sub("plus_train_gate", "plus_train", a.n_iter, precision="bf16x3", steps=3, warmup=1,
            note="plus_train in the gate-meeting split-bf16 mode")
        sub("plus_sampling", "plus_sampling", n_it8, steps=2,
            note="FABind+ sampling-mode inference (BASELINE configs[4]): n_iter 8, dropout sampling, DBSCAN centre, ranking head, "
                 "%d poses per complex and step" % a.poses)
        if rank == 0:
            out.update(extras)

    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    # reported baseline: rank 0 at N=1 only, and only for the modes whose workload the stack oracle restates (the full-model and
    # FABind+ sampling modes would otherwise carry a baseline of a different workload)
    if not a.no_cpu_baseline and world == 1 and a.mode in ("fwd", "fwdbwd"):
        out["cpu_baseline"] = cpu_baseline(a.hidden, a.layers, a.n_iter, a.n_prot, a.n_lig, backward=(a.mode == "fwdbwd"))
    if not a.no_cpu_baseline and world == 1 and headline_config3:
        out["cpu_baseline"] = cpu_baseline_config3(a.hidden, a.layers, a.n_iter, a.n_prot, a.n_lig)
    if headline_config3:
        out["north_star_targets"] = north_star_targets(out["value"], world, out.get("fwd", {}).get("value"), out.get("stack_fwdbwd", {}).get("value"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
