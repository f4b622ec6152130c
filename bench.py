#!/usr/bin/env python3
"""Benchmark of the ADMM projection hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config resnet50_tt]

One *step* = one ADMM projection iteration over every compressed layer of the rank table
(ADMM.update(update_u=True) semantics: Z <- proj(W+U), U += W-Z, ||W-Z||^2).  Inputs are synthetic
(N(0, 2/fan_in), seed 0) and resident in HBM before the timed region.  With N > 1 (launched by
torch.distributed.run) the layers are sharded over the ranks by LPT and the only collective is one
RCCL all-reduce of the scalar residual per step; total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor figure; = fp64 vector peak on CDNA4)
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def layer_entries(model, hp, fmt, dev):
    from tadmm._cabi import KIND_SVD, KIND_TT_CONV, KIND_TT_LINEAR
    entries, names = [], []
    for name, p in model.named_parameters():
        if name not in hp.ranks:
            continue
        w = p.data.to(dev).contiguous()
        if fmt == "tt":
            kind = KIND_TT_CONV if w.dim() == 4 else KIND_TT_LINEAR
            e = dict(kind=kind, tt_shapes=list(hp.tt_shapes[name]), ranks=list(hp.ranks[name]))
        else:
            e = dict(kind=KIND_SVD, ranks=hp.ranks[name])
        e["W"] = w
        entries.append(e)
        names.append(name)
    return entries, names


def cpu_baseline(config, max_seconds=30.0):
    """The oracle (numpy -> LAPACK sgesdd, same call sequence as ttd.py/admm.py) timed on the host cores."""
    from oracle import tt_oracle as O
    from tadmm import workloads
    model, hp, fmt = workloads.build(config, seed=0)
    w = {k: p.detach().numpy() for k, p in model.named_parameters()}
    u = {k: np.zeros_like(v) for k, v in w.items()}
    ranks = {k: (list(v) if not isinstance(v, int) else v) for k, v in hp.ranks.items()}
    tts = getattr(hp, "tt_shapes", None)
    t0 = time.perf_counter()
    O.admm_update(w, u, fmt, ranks, tts)         # warm-up sweep (BLAS thread pool, page faults)
    warm = time.perf_counter() - t0
    times = []
    while sum(times) + warm < max_seconds and len(times) < 3:
        t0 = time.perf_counter()
        O.admm_update(w, u, fmt, ranks, tts)
        times.append(time.perf_counter() - t0)
        if times[-1] > max_seconds / 2:
            break
    if not times:
        times = [warm]
    best = float(np.median(times))
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        th = [i.get("num_threads") for i in threadpool_info() if i.get("user_api") == "blas"]
        if th:
            threads = int(max(th))
    except Exception:
        pass
    return dict(value=1.0 / best, unit="iters/s", cores=threads, kind="port",
                sample=f"{len(times)} full {config} sweep(s) after 1 warm-up sweep, median, numpy {np.__version__} LAPACK sgesdd",
                seconds_per_sweep=best)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="resnet50_tt")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N>1: weak = one table per rank (headline, value = tables x it/s), with the strong variant "
                         "timed beside it; strong = only ONE table LPT-sharded over the ranks")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: time every part of an N-way layer shard one after the other on this GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev)          # one rank per GPU (rehearsals may fold ranks onto one card)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TADMM_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from tadmm import ops, sched, workloads
    model, hp, fmt = workloads.build(args.config, seed=0)
    entries, names = layer_entries(model, hp, fmt, dev)
    flops = [sched.layer_flops(e["kind"], list(e["W"].shape), e.get("tt_shapes"), e["ranks"]) for e in entries]
    costs = [f["svd"] + f["rec"] for f in flops]
    if args.emulate_world > 1:
        emu = sched.lpt_partition(costs, args.emulate_world)
        res = []
        for part in emu:
            for i in part:
                entries[i]["U"] = torch.zeros_like(entries[i]["W"])
                entries[i]["Z"] = torch.empty_like(entries[i]["W"])
            pl = ops.ProjectionPlan([entries[i] for i in part])
            for _ in range(2):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            res.append(dict(layers=len(part), ms=1e3 * (time.perf_counter() - t0) / args.steps,
                            gflop=sum(costs[i] for i in part) / 1e9, names=[names[i] for i in part][:4]))
            pl.close()
        print(json.dumps({"emulate_world": args.emulate_world, "max_ms": max(r["ms"] for r in res), "parts": res}))
        return
    total_resid = torch.zeros(1, dtype=torch.float64, device=dev)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(plan_):
        """W warm-up + K timed iterations of `plan_` (None = this rank owns no layer); max over ranks, seconds."""
        def step():
            if plan_ is not None:
                r = plan_.run(update_u=True)
                total_resid.copy_(r.sum().reshape(1))
            else:
                total_resid.zero_()
            if dist is not None:
                dist.all_reduce(total_resid)          # the ONE collective of the path: scalar residual
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el

    def make_plan(ents, idx):
        for i in idx:
            ents[i]["U"] = torch.zeros_like(ents[i]["W"])
            ents[i]["Z"] = torch.empty_like(ents[i]["W"])
        return ops.ProjectionPlan([ents[i] for i in idx]) if idx else None

    # Layers are independent units (admm.py:43), so the N-GPU job shards them with no data-path collective.
    # Headline (weak scaling, per-GPU work fixed): N tables of the configuration (table r seeded r, e.g. N models
    # or N rho/rank settings being compressed at once), rank r projecting table r; value = tables x iterations/s.
    # The strong-scaling variant north_star also asks for (ONE table LPT-sharded over the ranks) is timed right
    # after and reported beside it under "strong_scaling" -- it is bounded by the latency of one layer's chain of
    # eigen-solves.
    strong = None
    if world > 1 and args.scaling == "weak":
        parts = sched.lpt_partition(costs, world)
        splan = make_plan(entries, parts[rank])
        sel = timed(splan)
        strong = {"value": args.steps / sel, "unit": "iters/s", "ms_per_step": 1e3 * sel / args.steps,
                  "layers_per_rank": [len(p_) for p_ in parts], "residual_sq": float(total_resid[0]),
                  "note": "ONE table, layers LPT-sharded over the ranks, same K/W"}
        if splan is not None:
            splan.close()
        del splan
        model, hp, fmt = workloads.build(args.config, seed=rank)
        entries, names = layer_entries(model, hp, fmt, dev)
        mine = list(range(len(entries)))
        parts = [mine] * world
        tables = world
    else:
        parts = sched.lpt_partition(costs, world)
        mine = parts[rank]
        tables = 1
    plan = make_plan(entries, mine)
    elapsed = timed(plan)

    ms_per_step = 1e3 * elapsed / args.steps
    iters_per_s = tables * args.steps / elapsed
    tot = {k: sum(f[k] for f in flops) for k in ("svd", "rec", "gram", "proj", "eig", "numel")}
    out = {
        "metric": "ADMM projection iters/sec (all layers) + per-layer SVD GFLOP/s, ResNet-50 TT ranks",
        "value": iters_per_s, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if tables > 1 or world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f32 (Gram + eigen-solve accumulate in f64)", "data": "synthetic",
        "config": {"workload": f"{args.config}: {len(entries)} layers, {int(tot['numel'])} weights, "
                               f"hp table {workloads.CONFIGS[args.config][0]}",
                   "tables": tables, "layers_per_rank": [len(p) for p in parts],
                   "parallelism": f"layer-shard x{world}" + (f" ({tables} tables, one per rank)" if tables > 1 else "")},
        "svd_gflops_per_s": (tot["svd"] + tot["rec"]) * iters_per_s / 1e9,
        "algorithmic_gflop_per_iter": {k: v / 1e9 for k, v in tot.items() if k != "numel"},
        "residual_sq": float(total_resid[0]),
    }
    if strong is not None:
        out["strong_scaling"] = strong

    if rank == 0 and plan is not None and not args.no_roofline:
        # instrumented pass: HIP events around each phase on the launch stream (adds syncs, so it is
        # separate from the timed region above)
        plan.enable_timing(True)
        acc = None
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            plan.run(update_u=True)
            t = plan.last_timing()
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        ph = {k: v / reps for k, v in acc.items()}
        plan.enable_timing(False)
        my_gram = sum(flops[i]["gram"] for i in mine)
        my_mfma32 = sum(flops[i]["proj"] + flops[i]["rec"] for i in mine)
        my_bytes = 16.0 * sum(flops[i]["numel"] for i in mine)
        gram_tf = my_gram / (ph["gram_ms"] * 1e-3) / 1e12 if ph["gram_ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
               # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); bench.py cannot run rocprofv3 itself
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if args.config == "resnet50_tt" and world == 1:
                traffic = pm["kernels"]["gram_partial_kernel"]["hbm_bytes_per_launch_corrected"]
                traffic_src = "profiles/r01_pmc_traffic.json (per launch of gram_partial_kernel)"
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": "gram_partial_kernel+gram_reduce_kernel (fp64 MFMA 16x16x4)",
                           "achieved": gram_tf, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": gram_tf / PEAK_F64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                           "launches_per_step": 4,
                           "note": "dominant MATRIX-CORE kernel; algorithmic 2*M*N^2 per unfolding (18.63 GFLOP/step) / "
                                   "HIP-event time of the Gram launches on rank 0.  The dominant kernel by TIME is the "
                                   "latency-bound Jacobi tick (see phases_ms / roofline_other.eig_time_share)."}
        gemm_ms = ph["project_ms"] + ph["reconstruct_ms"]
        sweep_ms = ph["unfold_ms"] + ph["fold_update_ms"]
        out["phases_ms"] = ph
        out["roofline_other"] = {
            "gemm_f32_mfma": {"achieved_tflops": my_mfma32 / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                              "peak_tflops": PEAK_F32_MFMA_TFLOPS},
            "hbm_sweeps": {"achieved_gbs": my_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0,
                           "peak_gbs": PEAK_HBM_GBS, "algorithmic_bytes": my_bytes},
            "eig_time_share": ph["eig_ms"] / max(1e-9, sum(v for k, v in ph.items() if k.endswith("_ms"))),
            # the eigen-solve against the fp64 matrix peak, with the implementation-independent 8*N^3 model
            "eig_f64": {"algorithmic_tflops": sum(flops[i]["eig"] for i in mine) / (ph["eig_ms"] * 1e-3) / 1e12
                        if ph["eig_ms"] > 0 else 0.0, "peak_tflops": PEAK_F64_MFMA_TFLOPS},
        }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.config)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
