"""Three lanes experiment: long chains / mid layers / the rest, three streams, three host threads."""
import os, sys, threading, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
os.environ["TADMM_LANES"] = "1"          # the sub-plans here must not split again
from tadmm import ops, workloads, sched
from tadmm._cabi import KIND_TT_CONV
from bench import layer_entries
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build("resnet50_tt", seed=0)
entries, names = layer_entries(model, hp, fmt, dev)
for e in entries:
    e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
lat = []
for (n, p) in model.named_parameters():
    prof = sched.layer_latency_profile(KIND_TT_CONV, list(p.shape), hp.tt_shapes[n], list(hp.ranks[n]))
    lat.append(sum(prof[0]))
lmax = max(lat)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def trial(thresholds, prios):
    groups = [[] for _ in range(len(thresholds) + 1)]
    for e, l in zip(entries, lat):
        k = sum(1 for t in thresholds if l < t * lmax)
        groups[k].append(e)
    groups = [g for g in groups if g]
    plans = [ops.ProjectionPlan(g) for g in groups]
    streams = [torch.cuda.Stream(priority=pr) for pr in prios[:len(plans)]]
    solo = [timeit(lambda p=p: p.run(True), 5) for p in plans]
    def both():
        def w(p, s):
            with torch.cuda.stream(s): p.run(True)
        ts = [threading.Thread(target=w, args=(p, s)) for p, s in zip(plans, streams)]
        for t in ts: t.start()
        for t in ts: t.join()
    ms = timeit(both)
    print("thresholds %s sizes %s solo %s -> %.2f ms" % (thresholds, [len(g) for g in groups], ["%.2f" % x for x in solo], ms), flush=True)
    for p in plans: p.close()
trial([0.6], [-1, 0])
trial([0.6, 0.3], [-1, 0, 0])
trial([0.6, 0.3], [-1, -1, 0])
trial([0.95, 0.6], [-1, -1, 0])
trial([0.95, 0.6, 0.3], [-1, -1, 0, 0])
trial([0.6, 0.12], [-1, 0, 0])
