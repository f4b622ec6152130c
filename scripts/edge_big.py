"""Constant / rank-1 weights at resident (N <= 1152) and streamed (N > 1152) eigen-problem sizes: max |Z - W| and the residual.
usage (GPU box): python scripts/edge_big.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")]
import torch
from tadmm import ops
from tadmm._cabi import KIND_SVD
dev = torch.device("cuda:0")
for m in (512, 1024, 1152, 1280, 1408, 2048):
    for name, w in (("const", torch.ones(m, 1600) * 0.25),
                    ("ramp", torch.outer(torch.arange(m, dtype=torch.float32) / m - 0.3, torch.ones(1600)))):
        for r in (64, 300):
            L = dict(kind=KIND_SVD, W=w.to(dev), U=torch.zeros(m, 1600, device=dev), Z=torch.empty(m, 1600, device=dev), ranks=r)
            plan = ops.ProjectionPlan([L])
            resid = float(plan.run(update_u=True).cpu()[0])
            z = L["Z"].cpu()
            print(f"N={m:5d} {name:5s} r={r:3d}: max|Z-W| = {float((z - w).abs().max()):.3e}  rel = "
                  f"{float((z - w).norm() / w.norm()):.3e}  resid^2 = {resid:.3e}", flush=True)
            plan.close()
