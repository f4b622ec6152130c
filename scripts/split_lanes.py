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
# --- with stream priorities: the long chains first
lo, hi = -1, 0
try:
    import ctypes
    rng = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None
    print("priority range", rng)
except Exception as ex:
    print("no priority_range", ex)
for pa_prio, pb_prio in ((-1, 0), (0, -1)):
    sa, sb = torch.cuda.Stream(priority=pa_prio), torch.cuda.Stream(priority=pb_prio)
    print("A prio %d, B prio %d: %.2f ms" % (pa_prio, pb_prio, timeit(both)))
pa.close(); pb.close()
# --- other splits
def split(pred, label):
    global pa, pb, sa, sb
    A2 = [e for e, n in zip(entries, names) if pred(n)]
    B2 = [e for e, n in zip(entries, names) if not pred(n)]
    pa, pb = ops.ProjectionPlan(A2), ops.ProjectionPlan(B2)
    ta_, tb_ = timeit(lambda: pa.run(True)), timeit(lambda: pb.run(True))
    for pr in ((0, 0), (-1, 0)):
        sa, sb = torch.cuda.Stream(priority=pr[0]), torch.cuda.Stream(priority=pr[1])
        print("%s: A %d layers %.2f ms, B %d layers %.2f ms, both (prio %s) %.2f ms" % (label, len(A2), ta_, len(B2), tb_, pr, timeit(both)))
    pa.close(); pb.close()
split(lambda n: n.startswith("layer4.") and ".conv2." in n, "layer4.conv2")
split(lambda n: (n.startswith("layer4.") or n.startswith("layer3.")) and ".conv2." in n, "layer3+4.conv2")
