"""A/B of CREATION-time switches: a fresh plan per setting and round, alternating, median ms per iteration.
    python3 scripts/ab_create.py resnet50_tt TADMM_LANE_THRESHOLD=0.6 TADMM_LANE_THRESHOLD=0.7 [...]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from tadmm import ops, workloads
cfg, settings = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build(cfg, seed=0)
res = {s: [] for s in settings}
lanes = {}
for rnd in range(4):
    for s in settings:
        for kv in s.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        entries, names = bench.layer_entries(model, hp, fmt, dev)
        for e in entries:
            e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
        plan = ops.ProjectionPlan(entries)
        for _ in range(3):
            plan.run(update_u=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            plan.run(update_u=True)
        torch.cuda.synchronize()
        res[s].append(1e3 * (time.perf_counter() - t0) / 20)
        ln = plan.lanes() if hasattr(plan, "lanes") else []
        lanes[s] = [ln.count(0), ln.count(1)]
        plan.close()
for s in settings:
    print("%-44s median %.3f ms  (min %.3f max %.3f) lanes %s" % (s, statistics.median(res[s]), min(res[s]), max(res[s]), lanes[s]))
