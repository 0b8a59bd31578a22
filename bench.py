"""bench.py -- throughput of the FABind docking hot path on N MI355X GPUs (one process per GPU).

python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path (EfficientMCAttModel: 4 FABind layers + out layer, hidden 512,
refinement iterations = --n-iter) over one resident synthetic batch of --batch complexes of
1500 protein / 40 ligand nodes per GPU (BASELINE.json configs[1]/[2]).  Prints ONE JSON line on rank 0.
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
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


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


def build_model(hidden, layers, n_iter, seed=0):
    from fabind_amd.models.att_model import EfficientMCAttModel
    torch.manual_seed(seed)
    m = EfficientMCAttModel(stack_args(hidden, layers, n_iter), hidden, hidden, 1, n_layers=layers, n_iter=n_iter,
                            dropout=0.0, normalize_coord=lambda x: x / 5.0, unnormalize_coord=lambda x: x * 5.0)
    return m


def make_batch(batch, n_prot, n_lig, hidden, seed):
    """Seeded synthetic complexes; 4 distinct geometries tiled to `batch` (host generation cost)."""
    from fabind_amd import synthetic
    uniq = min(batch, 4)
    base = synthetic.make_stack_batch([(n_prot, n_lig)] * uniq, hidden, seed=seed, snap=False)
    if batch == uniq:
        return base
    reps = (batch + uniq - 1) // uniq
    n = base["X"].shape[0]
    out = {}
    for k in ("X", "H", "segment_id", "mask", "is_global", "coord_LAS"):
        out[k] = torch.cat([base[k]] * reps)[: n // uniq * batch]
    per = n // uniq
    out["batch_id"] = torch.repeat_interleave(torch.arange(batch), per)
    shift = lambda e: torch.cat([e + i * n for i in range(reps)], 1)
    ce, le = shift(base["compound_edge_index"]), shift(base["LAS_edge_index"])
    out["compound_edge_index"] = ce[:, ce[0] < per * batch]
    out["LAS_edge_index"] = le[:, le[0] < per * batch]
    return out


def cpu_baseline(hidden, layers, n_iter, n_prot, n_lig, budget_s=25.0, backward=False):
    """The oracle (CPU restatement of the reference algorithm, `kind: port`) timed on the host cores.  backward=True: the
    same pass as the GPU step of the default mode -- forward, the same scalar loss, backward to every parameter (autograd
    through the oracle) -- so that `cpu_baseline.value` and `value` measure the same work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fabind_oracle as orc
    # torch's default (128 threads on the 256-CPU host of the GPU box) is the WORST setting for this workload: forward
    # 0.12 complexes/s at 128 threads, 0.26 at 64, 0.39 at 32, 0.38 at 16, 0.31 at 8 (tools/probes/cpu_threads.py)
    default_threads = torch.get_num_threads()
    torch.set_num_threads(min(default_threads, 32))
    m = build_model(hidden, layers, n_iter)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if backward:
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
    inp = make_batch(1, n_prot, n_lig, hidden, seed=0)

    def run():
        X, Hh = orc.stack_forward(sd, "", inp["X"], inp["H"], inp["batch_id"], inp["segment_id"], inp["mask"],
                                  inp["is_global"], inp["compound_edge_index"], inp["LAS_edge_index"],
                                  inp["coord_LAS"], layers, n_iter)[:2]
        if backward:
            for v in sd.values():
                v.grad = None
            ((X * X).mean() + (Hh * Hh).mean() * 1e-6).backward()
    with torch.set_grad_enabled(backward):
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
    return dict(value=1.0 / dt, unit="complexes/s", cores=cores, kind="port",
                sample="oracle stack %s, B=1, %d/%d nodes, hidden %d, %d layers, n_iter=%d, fp32, %d timed run(s) after one warm-up"
                       % ("forward + backward" if backward else "forward", n_prot, n_lig, hidden, layers, n_iter, reps))


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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--poses", type=int, default=20, help="plus_sampling: poses sampled per complex and step")
    ap.add_argument("--mode", default="fwdbwd", choices=["fwd", "fwdbwd", "model", "plus_sampling"],
                    help="fwd / fwdbwd: the layer stack on the whole graph (SURVEY 8(d), the headline); model: the full "
                         "IaBNet (pocket model on the whole protein -> pocket crop -> complex model -> heads) with the "
                         "pocket-cls + coord + distmap losses, fwd+bwd (BASELINE configs[2] read literally); plus_sampling: "
                         "FABind+ sampling-mode inference (BASELINE configs[4]: dropout sampling, DBSCAN centre choice, "
                         "confidence head, --poses per complex)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
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
    if a.mode == "plus_sampling":
        from fabind_amd import synthetic
        from fabind_amd.plus.models import get_model as get_model_plus

        class _Log:
            def log_message(self, m):
                pass
        margs = stack_args(a.hidden, a.layers if a.layers != 4 else 5, a.n_iter)
        for k_, v_ in dict(use_ln_mlp=True, mlp_hidden_scale=1, dropout=0.1, mha_heads=4, rel_dis_pair_bias="no",
                           inter_additional_mlp=False, only_last_LAS=False, geom_reg_steps=1, use_for_radius_pred="ligand",
                           dis_map_thres=15.0, pocket_radius_buffer=5.0, min_pocket_radius=20.0, force_fix_radius=False,
                           use_clustering=True, dbscan_eps=9.0, dbscan_min_samples=2, choose_cluster_prob=0.5,
                           confidence_training=True, stack_mlp=True, confidence_use_ln_mlp=True, confidence_dropout=0.2,
                           confidence_mlp_hidden_scale=1).items():
            setattr(margs, k_, v_)
        torch.manual_seed(0)
        model = get_model_plus(margs, _Log()).to(dev)
        model.train()                                              # --infer-dropout: dropout on, ranking head in eval
        for name_, sub_ in model.named_modules():
            if name_.startswith("confidence") or name_.startswith("ranking"):
                sub_.eval()
        uniq = min(a.batch, 4)
        hb = synthetic.make_hetero_batch([(a.n_prot, a.n_lig)] * a.batch if a.batch <= 8 else
                                         [(a.n_prot, a.n_lig)] * uniq * ((a.batch + uniq - 1) // uniq), seed=rank).to(dev)
        a.batch = int(hb["compound"].batch.max().item()) + 1
        t = None
    elif a.mode == "model":
        from fabind_amd import synthetic
        from fabind_amd.models import get_model
        from fabind_amd.models.model import compute_loss

        class _Log:
            def log_message(self, m):
                pass
        margs = stack_args(a.hidden, a.layers, a.n_iter)
        torch.manual_seed(0)
        model = get_model(margs, _Log(), dev).to(dev)
        model.eval()
        uniq = min(a.batch, 4)
        hb = synthetic.make_hetero_batch([(a.n_prot, a.n_lig)] * a.batch if a.batch <= 8 else
                                         [(a.n_prot, a.n_lig)] * uniq * ((a.batch + uniq - 1) // uniq), seed=rank)
        hb = hb.to(dev)
        a.batch = int(hb["compound"].batch.max().item()) + 1
        t = None
    else:
        model = build_model(a.hidden, a.layers, a.n_iter).to(dev)
        model.eval()
        inp = make_batch(a.batch, a.n_prot, a.n_lig, a.hidden, seed=rank)
        t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    params = [p for p in model.parameters()]
    cot = None

    def step():
        if a.mode == "plus_sampling":
            for _ in range(a.poses):
                model.inference(hb.clone())
            return
        if a.mode == "model":
            for p in params:
                p.grad = None
            data = hb.clone()
            out = model(data, stage=1, train=False)
            loss, _ = compute_loss(out, data)
            loss.backward()
            if world > 1:
                from fabind_amd import parallel
                parallel.allreduce_gradients(params, world)
            return
        X0 = t["X"].clone()
        if a.mode == "fwd":
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
            if world > 1:
                from fabind_amd import parallel
                parallel.allreduce_gradients(params, world)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    K.PROFILE = {} if rank == 0 else None
    t0 = time.time()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.time() - t0
    prof = K.PROFILE
    K.PROFILE = None
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    value = a.batch * world * a.steps / dt * (a.poses if a.mode == "plus_sampling" else 1)
    out = {
        "metric": ("poses/sec, FABind+ sampling-mode inference (dropout sampling + DBSCAN centre + confidence head)"
                   if a.mode == "plus_sampling" else
                   "complexes/sec fwd+bwd, full IaBNet (pocket model + pocket crop + complex model + heads) with "
                   "pocket-cls + coord + distmap losses" if a.mode == "model" else
                   "complexes/sec %s (1500p/40l nodes), one stack pass per refinement iteration" % (
                       "fwd+bwd" if a.mode == "fwdbwd" else "fwd")),
        "value": value, "unit": "poses/s" if a.mode == "plus_sampling" else "complexes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": ("synthetic batch=%d/GPU, %d protein / %d ligand nodes, FABind+ model (5-layer LN-MLP stack, hidden "
                                "%d, n_iter=%d), %d poses per complex and step" % (a.batch, a.n_prot, a.n_lig, a.hidden,
                                                                                    a.n_iter, a.poses))
                   if a.mode == "plus_sampling" else
                   "synthetic batch=%d/GPU, %d protein / %d ligand nodes, %d-layer FABind stack + out layer, "
                   "hidden %d, n_iter=%d, %s" % (a.batch, a.n_prot, a.n_lig, a.layers, a.hidden, a.n_iter, a.mode),
                   "global_batch": a.batch * world, "n_iter": a.n_iter, "pass": a.mode},
    }
    # ---- roofline of the dominant kernel (live HIP-event timing of every GEMM launch in the timed region)
    if prof:
        best = None
        for label, evs in prof.items():
            ms = sum(s_.elapsed_time(e_) for s_, e_, _ in evs)
            if best is None or ms > best[1]:
                best = (label, ms, len(evs), evs[0][2])
        label, ms, cnt, flops = best
        avg_s = ms / cnt * 1e-3
        peak = MFMA_PEAK_TFLOPS[a.precision]
        out["roofline"] = {"bound": "mfma", "achieved": flops / avg_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": flops / avg_s / 1e12 / peak, "traffic": None, "kernel": label,
                           "flop_per_launch": flops, "launches": cnt, "avg_us": avg_s * 1e6,
                           "share_of_step": ms / (1e3 * dt)}
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM"), per launch; null when no PMC summary is available
    pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if "roofline" in out and os.path.exists(pmc):
        try:
            tr = json.load(open(pmc))
            for key, val in tr.get("per_kernel", {}).items():
                if key in out["roofline"]["kernel"]:
                    out["roofline"]["traffic"] = val
        except Exception:
            pass
    # reported baseline: rank 0 at N=1 only, and only for the modes whose workload the stack oracle restates (the full-model and
    # FABind+ sampling modes would otherwise carry a baseline of a different workload)
    if not a.no_cpu_baseline and world == 1 and a.mode in ("fwd", "fwdbwd"):
        out["cpu_baseline"] = cpu_baseline(a.hidden, a.layers, a.n_iter, a.n_prot, a.n_lig, backward=(a.mode == "fwdbwd"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
