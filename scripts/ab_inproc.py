"""In-process A/B of run-time switches (environment variables the library reads on every run): alternates the settings
over several rounds on the SAME plan and reports the median ms per iteration of each.
    python3 scripts/ab_inproc.py resnet50_tt TADMM_FILTER_GUARD=3 TADMM_FILTER_GUARD=0 [...]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from tadmm import ops, workloads

cfg, settings = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build(cfg, seed=0)
entries, names = bench.layer_entries(model, hp, fmt, dev)
for e in entries:
    e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
plan = ops.ProjectionPlan(entries)
res = {s: [] for s in settings}
for rnd in range(6):
    for s in settings:
        for kv in s.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        for _ in range(3):
            plan.run(update_u=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            plan.run(update_u=True)
        torch.cuda.synchronize()
        res[s].append(1e3 * (time.perf_counter() - t0) / 15)
for s in settings:
    print("%-40s median %.3f ms  (min %.3f max %.3f)" % (s, statistics.median(res[s]), min(res[s]), max(res[s])))
print("filter", plan.filter_stats())
