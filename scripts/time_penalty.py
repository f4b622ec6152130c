"""Time the augmented-Lagrangian penalty (append_admm_loss, admm.py:80-85) forward + backward on the ResNet-50 table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import workloads
from tadmm.admm import ADMM
dev = torch.device("cuda", 0)
model, hp, fmt = workloads.build("resnet50_tt", seed=0)
model = model.to(dev)
a = ADMM(model, 1e-3, hp, fmt, dev)
a.update(update_u=False)
a.update()
numel = sum(p.numel() for p in model.parameters())


def step(backward):
    for p in model.parameters():
        p.grad = None
    loss = a.append_admm_loss(torch.zeros((), device=dev))
    if backward:
        loss.backward()
    return loss


for bw in (False, True):
    for _ in range(5):
        step(bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step(bw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    byts = numel * 4 * (4 if bw else 3)
    print("penalty %s: %.3f ms  (%d weights, algorithmic %.0f MB -> %.2f TB/s)" %
          ("fwd+bwd" if bw else "fwd", ms, numel, byts / 1e6, byts / ms / 1e9))

# the reference's formulation (admm.py:80-85) in plain torch ops on the same device, for context
names = [n for n, _ in a._named()]
params = [p for _, p in a._named()]


def ref_step(backward):
    for p in params:
        p.grad = None
    loss = torch.zeros((), device=dev)
    for n, p in zip(names, params):
        loss = loss + 0.5 * a.rho * (torch.norm(p - a.z[n] + a.u[n], p=2) ** 2)
    if backward:
        loss.backward()
    return loss


for bw in (False, True):
    for _ in range(5):
        ref_step(bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        ref_step(bw)
    torch.cuda.synchronize()
    print("torch-op formulation %s: %.3f ms" % ("fwd+bwd" if bw else "fwd", 1e3 * (time.perf_counter() - t0) / n))
