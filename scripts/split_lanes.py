"""Times the ResNet-50 table split into its two independent lanes (multi-step 3x3 chains / single-step 1x1 layers),
each alone and both concurrently from two host threads on two streams."""
import os, sys, threading, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
from tadmm import ops, workloads
from bench import layer_entries
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build("resnet50_tt", seed=0)
entries, names = layer_entries(model, hp, fmt, dev)
for e in entries:
    e["U"] = torch.zeros_like(e["W"]); e["Z"] = torch.empty_like(e["W"])
A = [e for e, n in zip(entries, names) if ".conv2." in n]
B = [e for e, n in zip(entries, names) if ".conv2." not in n]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
pall = ops.ProjectionPlan(entries); print("all   %.2f ms" % timeit(lambda: pall.run(True))); pall.close()
pa = ops.ProjectionPlan(A); print("lane A (%d layers) %.2f ms" % (len(A), timeit(lambda: pa.run(True))))
pb = ops.ProjectionPlan(B); print("lane B (%d layers) %.2f ms" % (len(B), timeit(lambda: pb.run(True))))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    def ra():
        with torch.cuda.stream(sa): pa.run(True)
    def rb():
        with torch.cuda.stream(sb): pb.run(True)
    ta, tb = threading.Thread(target=ra), threading.Thread(target=rb)
    ta.start(); tb.start(); ta.join(); tb.join()
print("A || B on two streams / threads %.2f ms" % timeit(both))
