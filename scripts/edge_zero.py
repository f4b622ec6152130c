import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tadmm import ops
from tadmm._cabi import KIND_SVD, KIND_TT_CONV
from oracle import tt_oracle as O
dev = torch.device("cuda:0")
w = torch.zeros(64, 48)
L = dict(kind=KIND_SVD, W=w.to(dev), U=torch.zeros_like(w).to(dev), Z=torch.ones_like(w).to(dev), ranks=5)
pl = ops.ProjectionPlan([L]); r = pl.run(update_u=False); torch.cuda.synchronize()
print("svd zeros: device Z finite", bool(torch.isfinite(L["Z"]).all()), "max|Z|", L["Z"].abs().max().item(), "resid", r.tolist())
z = O.prune_linear_rank_svd(w.numpy(), 5)
print("  oracle finite", np.isfinite(z).all())
w4 = torch.zeros(16, 12, 3, 3)
L = dict(W=w4.to(dev), U=torch.zeros_like(w4).to(dev), Z=torch.ones_like(w4).to(dev), ranks=[6, 5])
pl = ops.TuckerPlan([L]); pl.run(update_u=False); torch.cuda.synchronize()
print("tucker zeros: device Z finite", bool(torch.isfinite(L["Z"]).all()), "max|Z|", L["Z"].abs().max().item() if torch.isfinite(L["Z"]).all() else None)
z = O.prune_rank_tk(w4.numpy(), [6, 5])
print("  oracle finite", np.isfinite(z).all())
w5 = torch.zeros(16, 16, 3, 3)
L = dict(kind=KIND_TT_CONV, W=w5.to(dev), U=torch.zeros_like(w5).to(dev), Z=torch.ones_like(w5).to(dev), tt_shapes=[4, 4, 9, 4, 4], ranks=[1, 4, 12, 12, 4, 1])
pl = ops.ProjectionPlan([L]); pl.run(update_u=False); torch.cuda.synchronize()
print("tt zeros: device Z finite", bool(torch.isfinite(L["Z"]).all()), "max|Z|", L["Z"].abs().max().item())
