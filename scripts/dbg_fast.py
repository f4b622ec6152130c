"""Filter products at fp32 accuracy (dgemm3): stats of a few ResNet-50 iterations, fast on / off."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
import torch
from tadmm import ops, workloads
from bench import layer_entries
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build("resnet50_tt", seed=0)
entries, names = layer_entries(model, hp, fmt, dev)
for e in entries:
    e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
pl = ops.ProjectionPlan(entries)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl.run(True); torch.cuda.synchronize()
    print("iter %d: %.2f ms  %s" % (it, 1e3 * (time.perf_counter() - t0), pl.filter_stats()), flush=True)
pl.enable_timing(True)
pl.run(True); torch.cuda.synchronize()
print("fp64 launches", pl.filter_timing(), "fast launches", pl.filter_timing_fast())
