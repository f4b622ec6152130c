"""Time ADMM.update on the Tucker configuration (BASELINE config 2: ResNet-32 CIFAR, tk_resnet32_hp 3x)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import workloads
from tadmm.admm import ADMM
dev = torch.device("cuda", 0)
model, hp, fmt = workloads.build("resnet32_tk", seed=0)
model = model.to(dev)
a = ADMM(model, 1e-3, hp, fmt, dev)
for _ in range(2):
    a.update()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    a.update()
torch.cuda.synchronize()
print("resnet32_tk ADMM.update: %.2f ms / iteration (%d layers)" % (1e3 * (time.perf_counter() - t0) / n, len(a.z)))
