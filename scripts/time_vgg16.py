"""Times ADMM.update on the VGG-16 Tucker tables (4096-wide classifier Grams: streamed Jacobi pairs).
usage (GPU box): python scripts/time_vgg16.py [vgg16_bn_tk|vgg16_tk]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")]
import torch
from tadmm import workloads
from tadmm.admm import ADMM

cfg = sys.argv[1] if len(sys.argv) > 1 else "vgg16_bn_tk"
dev = torch.device("cuda:0")
model, hp, fmt = workloads.build(cfg, seed=0)
model = model.to(dev)
a = ADMM(model, 1e-3, hp, fmt, dev, log=True)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    a.update(update_u=(i > 0))
    torch.cuda.synchronize(); print(f"{cfg} update {i}: {(time.time() - t0) * 1e3:.1f} ms", flush=True)
if a._tk_names:
    its, errs = a._tk.plan.iterations()
    print("HOOI sweeps", dict(zip(a._tk_names, its)))
