"""Homogeneous tables (DeiT-small, ResNet-18): does an even two-way split on two streams help?"""
import os, sys, threading, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
os.environ["TADMM_LANES"] = "1"
from tadmm import ops, workloads
from bench import layer_entries
dev = torch.device("cuda:0")
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for cfg in ("deit_small_tt", "resnet18_tt"):
    model, hp, fmt = workloads.build(cfg, seed=0)
    entries, names = layer_entries(model, hp, fmt, dev)
    for e in entries:
        e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
    pall = ops.ProjectionPlan(entries); t_all = timeit(lambda: pall.run(True)); pall.close()
    for label, A, B in (("even/odd", entries[0::2], entries[1::2]), ("first/second half", entries[:len(entries) // 2], entries[len(entries) // 2:])):
        pa, pb = ops.ProjectionPlan(A), ops.ProjectionPlan(B)
        sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
        def both():
            def w(p, s):
                with torch.cuda.stream(s): p.run(True)
            ts = [threading.Thread(target=w, args=(pa, sa)), threading.Thread(target=w, args=(pb, sb))]
            for t in ts: t.start()
            for t in ts: t.join()
        print(cfg, label, "one plan %.2f ms, two lanes %.2f ms" % (t_all, timeit(both)), flush=True)
        pa.close(); pb.close()
