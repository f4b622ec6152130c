#!/usr/bin/env python3
"""Benchmark of the ADMM projection hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config resnet50_tt]

One *step* = one ADMM projection iteration over every compressed layer of the rank table
(ADMM.update(update_u=True) semantics: Z <- proj(W+U), U += W-Z, ||W-Z||^2).  Inputs are synthetic
(N(0, 2/fan_in), seed 0) and resident in HBM before the timed region.

`--gpus N` with N > 1 and no torch.distributed environment starts the N ranks itself (a child
`python -m torch.distributed.run ... bench.py`, before this process touches the GPU).  At N > 1 `value` is ONE
table's iterations/s with its layers sharded over the ranks (strong scaling, BASELINE.json north_star; the only
collective is one RCCL all-reduce of the scalar residual per step); the throughput of N tables, one per rank, is
reported beside it under `weak_scaling`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor figure; = fp64 vector peak on CDNA4)
# what v_mfma_f64_16x16x4 sustains on this part, register operands only (scripts/micro/mfma_f64_peak.hip): 24 TF/s on one
# accumulator chain, 33-35 on 4-8 independent ones at one wave per SIMD, 45.3 at two waves per SIMD
PEAK_F64_MFMA_MEASURED_TFLOPS = 45.3
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="resnet50_tt")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-per-layer", action="store_true")
    ap.add_argument("--no-forward", action="store_true", help="skip the factorised-layer forward block (hot path B)")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the one-table-per-rank (weak scaling) pass")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: time every part of an N-way layer shard one after the other on this GPU")
    return ap.parse_args()


def self_launch(args):
    """`--gpus N` without a rendezvous environment: start the N ranks as a child launcher.  Nothing in this process
    has touched the GPU yet (torch is not even imported), so no process that initialised HIP is ever replaced."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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
    """The oracle (numpy -> LAPACK sgesdd, same call sequence as ttd.py/admm.py) timed on the host cores: one sweep at
    each of a few BLAS thread counts (the box's full core count oversubscribes these small SVDs), the fastest one is
    reported with the thread count it used."""
    import numpy as np
    from oracle import tt_oracle as O
    from tadmm import workloads
    model, hp, fmt = workloads.build(config, seed=0)
    w = {k: p.detach().numpy() for k, p in model.named_parameters()}
    u = {k: np.zeros_like(v) for k, v in w.items()}
    ranks = {k: (list(v) if not isinstance(v, int) else v) for k, v in hp.ranks.items()}
    tts = getattr(hp, "tt_shapes", None)
    ncpu = os.cpu_count() or 1
    tried = {}
    t_start = time.perf_counter()
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    # (the <= 64-column SVDs of the Tucker table are too small to thread: try 1 and 4 as well)
    candidates = [c for c in ((1, 4, 16) if fmt == "tk" else (16, 8, 32, ncpu)) if c <= ncpu] or [ncpu]
    seen = set()
    for th in candidates:
        if th in seen or (time.perf_counter() - t_start) > max_seconds:
            continue
        seen.add(th)
        def sweep():
            t0 = time.perf_counter()
            O.admm_update(w, u, fmt, ranks, tts)
            return time.perf_counter() - t0
        if threadpool_limits is not None:
            with threadpool_limits(limits=th, user_api="blas"):
                if not tried:
                    sweep()                      # warm-up (thread pool, page faults), not counted
                tried[th] = sweep()
        else:
            tried[ncpu] = sweep()
            break
    best_th = min(tried, key=tried.get)
    best = tried[best_th]
    return dict(value=1.0 / best, unit="iters/s", cores=int(best_th), kind="port",
                sample=f"one full {config} sweep per BLAS thread count {sorted(tried)} after a warm-up sweep; fastest "
                       f"reported; numpy {np.__version__} LAPACK sgesdd; host has {ncpu} cores",
                seconds_per_sweep=best, seconds_by_threads={str(k): v for k, v in tried.items()})


def load_pmc_traffic(config):
    """Newest committed PMC pass of this configuration (profiles/r*_pmc_traffic*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes; bench.py cannot run rocprofv3 itself).
    -> (kernels dict, meta dict)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config", "resnet50_tt") == config:
            best = (f, d)                      # sorted by name: the latest round wins
    if best is None:
        return {}, {}
    f, d = best
    return d.get("kernels", {}), {"file": os.path.relpath(f, ROOT), "commit": d.get("commit", "unknown (pass predates provenance)"),
                                  "plan_runs_in_pass": d.get("plan_runs_in_pass")}


def pmc_traffic_for(pm, meta, kernel, launches_per_step, world):
    """HBM bytes per launch of `kernel` from the committed PMC pass -- only if that pass ran the same number of launches per
    step as this run (+-10 %): a figure from another binary / another schedule is dropped (null + reason), not stamped on."""
    k = pm.get(kernel)
    if not k or world != 1:
        return None, (None if world == 1 else "single-GPU PMC pass only")
    lps = k.get("launches_per_step")
    if lps is None:
        return None, f"{meta.get('file')}: pass predates launch-count provenance; dropped"
    if abs(lps - launches_per_step) > 0.10 * max(lps, launches_per_step):
        return None, (f"{meta.get('file')} (commit {meta.get('commit')}): {lps:.0f} launches per step in the PMC pass vs "
                      f"{launches_per_step:.0f} in this run (> 10 % apart); dropped")
    return k.get("hbm_bytes_per_launch_corrected"), (f"{meta.get('file')} (commit {meta.get('commit')}, {lps:.0f} launches per "
                                                     f"step in the pass, {launches_per_step:.0f} here)")


def tucker_flops(shape, ranks, sweeps):
    """Dense-contraction FLOPs of one Tucker-2 projection (HOSVD Grams + `sweeps` HOOI sweeps + Z), and the 8 N^3 model of
    its eigen-solves: dict(mfma=.., eig=..).  admm.py:113-127 -> tensorly partial_tucker."""
    O, I = int(shape[0]), int(shape[1])
    k2 = 1
    for d in shape[2:]:
        k2 *= int(d)
    ro, ri = min(int(ranks[0]), O), min(int(ranks[1]), I)

    def gram(m, n):
        return 2.0 * max(m, n) * min(m, n) ** 2

    def eig(m, n):
        return 8.0 * min(m, n) ** 3
    hosvd = gram(O, I * k2) + gram(I, O * k2)
    per = 2.0 * O * k2 * I * ri + gram(O, k2 * ri) + 2.0 * I * k2 * O * ro + gram(I, k2 * ro) + 2.0 * ro * k2 * I * ri
    final = 2.0 * ro * k2 * ri * I + 2.0 * O * ro * k2 * I
    eigs = eig(O, I * k2) + eig(I, O * k2) + sweeps * (eig(O, k2 * ri) + eig(I, k2 * ro))
    return dict(mfma=hosvd + sweeps * per + final, eig=eigs, numel=O * I * k2)


def bench_tucker(args):
    """`--config resnet32_tk` (BASELINE config 2): the 'tk' branch of ADMM.update for the whole table -- one grouped
    HOSVD + HOOI plan on the device (csrc/tucker_plan.hip) -- same JSON schema as the TT configurations.
    PARITY UNPINNED for this branch (tensorly absent, no reference fixtures): DESIGN.md 3."""
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per requested GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ndev >= world:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from tadmm import ops, sched, workloads
    model, hp, fmt = workloads.build(args.config, seed=0)
    names, layers = [], []
    for name, p in model.named_parameters():
        if name in hp.ranks:
            w = p.data.to(dev).contiguous()
            layers.append(dict(W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=list(hp.ranks[name])))
            names.append(name)
    # layers are independent units (admm.py:43): LPT on the one-sweep contraction cost; no data-path collective
    parts = sched.lpt_partition([tucker_flops(L["W"].shape, L["ranks"], 3)["mfma"] for L in layers], world)
    mine = parts[rank]
    plan = ops.TuckerPlan([layers[i] for i in mine]) if mine else None
    total_resid = torch.zeros(1, dtype=torch.float64, device=dev)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if plan is not None:
            total_resid.copy_(plan.run(update_u=True).sum().reshape(1))
        else:
            total_resid.zero_()
        if dist is not None:
            dist.all_reduce(total_resid)

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
    ms_per_step = 1e3 * el / args.steps
    its, errs = plan.iterations() if plan is not None else ([], [])
    fl = [tucker_flops(layers[i]["W"].shape, layers[i]["ranks"], its[j]) for j, i in enumerate(mine)]
    mfma_able = sum(f["mfma"] for f in fl)
    out = {
        "metric": "ADMM projection iters/sec (all layers) + per-layer SVD GFLOP/s, ResNet-50 TT ranks",
        "value": args.steps / el, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None, "dtype": "f32 (Gram and eigen-solve in f64)", "data": "synthetic",
        "config": {"workload": f"{args.config}: {len(layers)} layers, {sum(f['numel'] for f in fl)} weights on rank 0, "
                               f"hp table {workloads.CONFIGS[args.config][0]} (Tucker-2: HOSVD + HOOI, tol 1e-4, <= 100 sweeps)",
                   "tables": 1, "layers_per_rank": [len(p) for p in parts],
                   "parallelism": f"layer-shard x{world} (LPT, one scalar all-reduce per step)"},
        "parity": "UNPINNED for the Tucker branch: tensorly is absent and the reference holds no fixtures (DESIGN.md 3); "
                  "device == the oracle's float64 restatement <= 5e-5 with identical HOOI sweep counts",
        "hooi_sweeps": {"max": max(its) if its else 0, "per_layer": its},
        "residual_sq": float(total_resid[0]),
    }
    out["roofline_sweep"] = {"bound": "mfma", "achieved": mfma_able / (ms_per_step * 1e-3) / 1e12,
                             "peak": PEAK_F32_MFMA_TFLOPS * world, "unit": "TFLOP/s",
                             "frac": mfma_able / (ms_per_step * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                             "note": "Gram + mode-product FLOPs of HOSVD, the HOOI sweeps each layer actually ran and Z / "
                                     "ms_per_step / fp32 MFMA peak; 0.46 M weights: the configuration is launch- and "
                                     "latency-bound (SURVEY 8d config 2)"}
    if rank == 0 and plan is not None and not args.no_roofline:
        plan.enable_timing(True)
        reps, acc = max(3, min(args.steps, 10)), None
        for _ in range(reps):
            plan.run(update_u=True)
            t = plan.last_timing()
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        plan.enable_timing(False)
        eig_ms, nl = acc["eig_ms"] / reps, acc["eig_launches"] / reps
        tf = acc["eig_model_flops"] / (acc["eig_ms"] * 1e-3) / 1e12 if acc["eig_ms"] > 0 else 0.0
        out["roofline"] = {
            "bound": "mfma", "kernel": "single-launch symmetric eigen-solver of the <= 64-column Gram matrices (one workgroup per "
                                       "problem; csrc/jacobi.hip / csrc/tridiag.hip)",
            "achieved": tf, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F64_MFMA_TFLOPS,
            "traffic": None, "traffic_note": "no PMC pass for this configuration (the problems live in LDS: 30 x 32 KiB per launch)",
            "launches_per_step": nl, "avg_launch_us": 1e3 * eig_ms / max(1.0, nl),
            "time_share": eig_ms / max(1e-9, acc["total_ms"] / reps),
            "flops_per_step": acc["eig_model_flops"] / reps,
            "note": "achieved = 8 N^3 model of a full symmetric eigen-decomposition, summed over the problems of a launch / "
                    "HIP-event time of the launch on the launch stream.  The launches are latency-bound (30 workgroups on 256 "
                    "CUs, a dependent chain inside each): the fraction says how little of the chip the configuration can use"}
        out["phases_ms"] = {"eig_ms": eig_ms, "total_ms_instrumented": acc["total_ms"] / reps}
    if rank == 0 and world == 1 and not args.no_per_layer:
        seen, per_layer = {}, []
        for i, L in enumerate(layers):
            sig = (tuple(L["W"].shape), tuple(L["ranks"]))
            if sig in seen:
                seen[sig]["count"] += 1
                continue
            one = dict(W=L["W"], U=torch.zeros_like(L["W"]), Z=torch.empty_like(L["W"]), ranks=L["ranks"])
            pl = ops.TuckerPlan([one])
            for _ in range(2):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            it1, _ = pl.iterations()
            pl.close()
            f = tucker_flops(L["W"].shape, L["ranks"], it1[0])
            svd = 0.0          # thin-SVD model 4MN^2 + 8N^3 of every unfolding the run decomposed
            O, I = L["W"].shape[0], L["W"].shape[1]
            k2 = L["W"][0, 0].numel()
            ro, ri = L["ranks"]
            svd += sched.svd_flops(O, I * k2) + sched.svd_flops(I, O * k2)
            svd += it1[0] * (sched.svd_flops(O, k2 * min(ri, I)) + sched.svd_flops(I, k2 * min(ro, O)))
            rec = {"layer": names[i], "shape": list(L["W"].shape), "ranks": list(L["ranks"]), "count": 1, "hooi_sweeps": it1[0],
                   "svd_gflop": svd / 1e9, "ms_alone": ms, "svd_gflops_per_s": svd / (ms * 1e-3) / 1e9}
            seen[sig] = rec
            per_layer.append(rec)
        out["per_layer_svd_gflops"] = per_layer
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.config)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.config.endswith("_tk"):
        return bench_tucker(args)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per requested GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev)          # one rank per GPU; a rehearsal on fewer cards folds ranks
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TADMM_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl" and ndev >= world:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo" if ndev < world else backend, rank=rank, world_size=world)

    from tadmm import ops, sched, workloads
    model, hp, fmt = workloads.build(args.config, seed=0)
    entries, names = layer_entries(model, hp, fmt, dev)
    flops = [sched.layer_flops(e["kind"], list(e["W"].shape), e.get("tt_shapes"), e["ranks"]) for e in entries]
    costs = [f["svd"] + f["rec"] for f in flops]
    profiles = [sched.layer_latency_profile(e["kind"], list(e["W"].shape), e.get("tt_shapes"), e["ranks"]) for e in entries]
    if args.emulate_world > 1:
        emu = sched.latency_partition(profiles, args.emulate_world)
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
                            model_ms=sched.rank_time_us([profiles[i] for i in part]) / 1e3,
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

    # ---- headline: ONE table; its layers are independent units (admm.py:43) sharded over the ranks by the latency
    #      model of sched.py; no data-path collective ----
    parts = sched.latency_partition(profiles, world)
    mine = parts[rank]
    plan = make_plan(entries, mine)
    elapsed = timed(plan)
    resid_headline = float(total_resid[0])
    ms_per_step = 1e3 * elapsed / args.steps
    iters_per_s = args.steps / elapsed
    tot = {k: sum(f[k] for f in flops) for k in ("svd", "rec", "gram", "proj", "eig", "numel")}
    mfma_able = tot["gram"] + tot["proj"] + tot["rec"]
    out = {
        "metric": "ADMM projection iters/sec (all layers) + per-layer SVD GFLOP/s, ResNet-50 TT ranks",
        "value": iters_per_s, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f32 (Gram, eigen-solve and its filter products accumulate in f64)", "data": "synthetic",
        "config": {"workload": f"{args.config}: {len(entries)} layers, {int(tot['numel'])} weights, "
                               f"hp table {workloads.CONFIGS[args.config][0]}",
                   "tables": 1, "layers_per_rank": [len(p) for p in parts],
                   "parallelism": f"layer-shard x{world} (latency-model partition, one scalar all-reduce per step)"},
        "svd_gflops_per_s": (tot["svd"] + tot["rec"]) * iters_per_s / 1e9,
        "svd_gflops_note": "thin-SVD model 4MN^2 + 8N^3 per unfolding + tt2ten chain, over ALL TT steps of the table -- "
                           "including the identity steps the Z-only projection skips (kept rank = row count; "
                           "TADMM_FLAG_SKIP_ROTATIONS) and the eigen-problems solved for their leading r vectors only",
        "algorithmic_gflop_per_iter": {k: v / 1e9 for k, v in tot.items() if k != "numel"},
        "residual_sq": resid_headline,
    }
    if ndev < world:
        out["folded_on"] = ndev      # rehearsal: several ranks share a card, the figure is not a scaling point
    if world > 1:
        out["strong_scaling_model_ms"] = {"per_rank": [sched.rank_time_us([profiles[i] for i in p]) / 1e3 for p in parts],
                                          "note": "sched.rank_time_us: one layer's chain of eigen-solves is latency-bound, "
                                                  "so sharding layers cannot shorten the longest chain (DESIGN.md)"}

    # north_star's yardstick: MFMA-able FLOPs of the whole sweep against the fp32 matrix peak
    out["roofline_sweep"] = {"bound": "mfma", "achieved": mfma_able / (ms_per_step * 1e-3) / 1e12,
                             "peak": PEAK_F32_MFMA_TFLOPS * world, "unit": "TFLOP/s",
                             "frac": mfma_able / (ms_per_step * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                             "note": "Gram + projection + chain GEMM FLOPs of one iteration (SURVEY 8d) / ms_per_step / "
                                     "fp32 MFMA peak: the whole-sweep figure north_star's 40 % target refers to"}

    if rank == 0 and plan is not None and not args.no_roofline:
        # instrumented pass: HIP events around each phase on the launch stream, and around every launch of the
        # dominant kernel (adds syncs, so it is separate from the timed region above)
        plan.enable_timing(True)
        acc, ft = None, dict(gemm_ms=0.0, gemm_launches=0, gemm_flops=0.0)
        jt = dict(tick_ms=0.0, tick_launches=0, tick_flops=0.0, tick_wgs=0.0)
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            plan.run(update_u=True)
            t = plan.last_timing()
            f = plan.filter_timing()
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
            ft = {k: ft[k] + f[k] for k in ft}
            j = plan.jacobi_timing()
            jt = {k: jt[k] + j[k] for k in jt}
        ph = {k: v / reps for k, v in acc.items()}
        plan.enable_timing(False)
        fstats = plan.filter_stats()
        my_gram = sum(flops[i]["gram"] for i in mine)
        my_mfma32 = sum(flops[i]["proj"] + flops[i]["rec"] for i in mine)
        my_bytes = 16.0 * sum(flops[i]["numel"] for i in mine)
        pm, pm_meta = load_pmc_traffic(args.config)
        total_ms = sum(v for k, v in ph.items() if k.endswith("_ms"))
        roof_tick = None
        if jt["tick_launches"] > 0 and jt["tick_ms"] > 0:
            tick_tf = jt["tick_flops"] / (jt["tick_ms"] * 1e-3) / 1e12
            ttraffic, tsrc = pmc_traffic_for(pm, pm_meta, "jacobi_tick3_kernel", jt["tick_launches"] / reps, world)
            roof_tick = {
                "bound": "mfma", "kernel": "jacobi_tick3_kernel (block-Jacobi tournament of the Rayleigh-Ritz and full eigen-solves: "
                                           "fp64 MFMA Gram / column updates around two serial 16x16 rotation solves)",
                "achieved": tick_tf, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tick_tf / PEAK_F64_MFMA_TFLOPS,
                "peak_measured": PEAK_F64_MFMA_MEASURED_TFLOPS, "frac_of_measured": tick_tf / PEAK_F64_MFMA_MEASURED_TFLOPS,
                "traffic": ttraffic, "traffic_source": tsrc,
                "launches_per_step": jt["tick_launches"] / reps,
                "avg_launch_us": 1e3 * jt["tick_ms"] / jt["tick_launches"],
                "workgroups_per_launch": jt["tick_wgs"] / jt["tick_launches"],
                "time_share": (jt["tick_ms"] / reps) / max(1e-9, total_ms),
                "flops_per_step": jt["tick_flops"] / reps,
                "note": "achieved = matrix-core flops the launches executed (2560 x row length per workgroup: cross Gram and two "
                        "rounds of column updates of a 32-column pair) / their HIP-event time, each launch timed on the launch "
                        "stream.  The kernel is latency-bound, not throughput-bound: a launch has 20-110 workgroups on 256 CUs "
                        "and 7.6 of its 14 us are two serial 16x16 rotation solves on one wave each (DESIGN.md 5, "
                        "scripts/stamp_jacobi.sh), so the fraction of the matrix peak says how little of the chip one "
                        "dependent chain can use, not how well the tile is written"}
        roof_gemm = None
        if ft["gemm_launches"] > 0 and ft["gemm_ms"] > 0:
            gemm_tf = ft["gemm_flops"] / (ft["gemm_ms"] * 1e-3) / 1e12
            traffic, gsrc = pmc_traffic_for(pm, pm_meta, "dgemm_nt_tile_kernel", ft["gemm_launches"] / reps, world)
            roof_gemm = {
                "bound": "mfma", "kernel": "dgemm_nt_tile_kernel<32,2> (fp64 MFMA 16x16x4, 64x32 tiles, eight waves: block products of the "
                                           "filtered eigen-solver)",
                "achieved": gemm_tf, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / PEAK_F64_MFMA_TFLOPS,
                "peak_measured": PEAK_F64_MFMA_MEASURED_TFLOPS, "frac_of_measured": gemm_tf / PEAK_F64_MFMA_MEASURED_TFLOPS,
                "traffic": traffic, "traffic_source": gsrc,
                "launches_per_step": ft["gemm_launches"] / reps,
                "avg_launch_us": 1e3 * ft["gemm_ms"] / ft["gemm_launches"],
                "time_share": (ft["gemm_ms"] / reps) / max(1e-9, total_ms),
                "flops_per_step": ft["gemm_flops"] / reps,
                "note": "peak = vendor fp64 matrix figure; peak_measured = what v_mfma_f64_16x16x4 "
                        "sustains here with register operands at two waves per SIMD (scripts/micro/mfma_f64_peak.hip; 33-35 TF/s at "
                        "one wave per SIMD).  achieved = 2*M*N*K of every product the launches executed "
                        "(gated-off problems excluded; read back from the device) / their HIP-event time, each launch "
                        "timed on the launch stream.  These FLOPs are the work of the filter, not part of the "
                        "thin-SVD model"}
        # `roofline` = whichever of the two kernels took more time in THIS run; the other one stays beside it
        cands = [r for r in (roof_gemm, roof_tick) if r]
        cands.sort(key=lambda r: -r["time_share"])
        if cands:
            out["roofline"] = dict(cands[0], dominant_by="summed HIP-event time of its launches in the instrumented pass")
        roof_second = cands[1] if len(cands) > 1 else None
        gemm_ms = ph["project_ms"] + ph["reconstruct_ms"]
        sweep_ms = ph["unfold_ms"] + ph["fold_update_ms"]
        out["phases_ms"] = ph
        lanes = plan.lanes() if hasattr(plan, "lanes") else [0] * len(mine)
        out["lanes"] = {"n": 1 + max(lanes), "layers_per_lane": [lanes.count(i) for i in range(1 + max(lanes))],
                        "note": "two lanes = two sub-plans on two device streams (long eigen-solve chains on a high-"
                                "priority stream, the rest beside them); phases_ms then sums both lanes and exceeds "
                                "ms_per_step, and per-launch kernel times include the other lane's competition"}
        out["filter"] = fstats
        out["roofline_other"] = {
            "gram_f64_mfma": {"achieved_tflops": my_gram / (ph["gram_ms"] * 1e-3) / 1e12 if ph["gram_ms"] > 0 else 0.0,
                              "peak_tflops": PEAK_F64_MFMA_TFLOPS, "algorithmic_gflop": my_gram / 1e9},
            "gemm_f32_mfma": {"achieved_tflops": my_mfma32 / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                              "peak_tflops": PEAK_F32_MFMA_TFLOPS},
            "hbm_sweeps": {"achieved_gbs": my_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0,
                           "peak_gbs": PEAK_HBM_GBS, "algorithmic_bytes": my_bytes},
            "second_kernel": roof_second,
            "eig_time_share": ph["eig_ms"] / max(1e-9, total_ms),
            # the whole eigen phase (filter + Rayleigh-Ritz Jacobi + fallbacks) against the fp64 matrix peak, with the
            # implementation-independent 8*N^3 model of a full symmetric eigen-decomposition
            "eig_f64": {"algorithmic_tflops": sum(flops[i]["eig"] for i in mine) / (ph["eig_ms"] * 1e-3) / 1e12
                        if ph["eig_ms"] > 0 else 0.0, "peak_tflops": PEAK_F64_MFMA_TFLOPS,
                        "jacobi_tick3": pm.get("jacobi_tick3_kernel")},
        }

    # ---- per-layer SVD GFLOP/s: every distinct layer shape projected ALONE (own single-layer plan) ----
    if rank == 0 and world == 1 and not args.no_per_layer:
        seen, per_layer = {}, []
        for i, e in enumerate(entries):
            sig = (tuple(e["W"].shape), tuple(e.get("tt_shapes") or ()), str(e["ranks"]))
            if sig in seen:
                seen[sig]["count"] += 1
                continue
            one = dict(e)
            one["U"] = torch.zeros_like(e["W"])
            one["Z"] = torch.empty_like(e["W"])
            pl = ops.ProjectionPlan([one])
            for _ in range(2):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                pl.run(update_u=True)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            pl.close()
            rec = {"layer": names[i], "shape": list(e["W"].shape), "count": 1, "svd_gflop": costs[i] / 1e9, "ms_alone": ms,
                   "svd_gflops_per_s": costs[i] / (ms * 1e-3) / 1e9}
            seen[sig] = rec
            per_layer.append(rec)
        out["per_layer_svd_gflops"] = per_layer

    # ---- weak scaling beside the headline: N tables of the configuration, table r seeded r on rank r ----
    if world > 1 and not args.no_weak:
        if plan is not None:
            plan.close()
        del plan
        model, hp, fmt = workloads.build(args.config, seed=rank)
        wentries, _ = layer_entries(model, hp, fmt, dev)
        wplan = make_plan(wentries, list(range(len(wentries))))
        wel = timed(wplan)
        out["weak_scaling"] = {"value": world * args.steps / wel, "unit": "tables x iters/s", "tables": world,
                               "ms_per_step": 1e3 * wel / args.steps,
                               "note": "one table per rank (table r seeded r), no data-path collective, same K/W"}
    # ---- hot path B: forward of the factorised layers against the dense layers they replace (tadmm/fwdbench.py)
    if rank == 0 and world == 1 and not args.no_forward:
        from tadmm import fwdbench
        rows = fwdbench.run(dev)
        head = next((r for r in rows if "qkv" in r["layer"] and r["dtype"] == "bf16" and "roofline" in r), None)
        out["forward"] = {"note": "module call (inference, weights packed once) vs the dense torch op of the same shape and "
                                  "dtype.  speedup_vs_dense compares the two under hipGraph replay (20 calls per graph: device "
                                  "time, the host out of the loop); ms / dense_ms / eager_speedup_vs_dense are eager calls, "
                                  "which at these sizes are host-bound on both sides (~20 us of dispatch around 10-40 us "
                                  "kernels) and measure Python overhead; roofline = chain kernel alone, executed bf16 MFMA flops",
                          "layers": rows, "roofline": None if head is None else head["roofline"]}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.config)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
