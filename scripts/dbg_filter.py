"""Debug driver of the filtered eigen-solver: low-rank fallback case and a Gaussian case, with TADMM_DEBUG output."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dnn-compression-tensor-admm_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tadmm import ops
from tadmm._cabi import KIND_TT_CONV
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
shape, tts, ranks = (1024, 256, 1, 1), [1024, 1, 256], [1, 75, 75, 1]
w = (rng.standard_normal(shape) * np.sqrt(2.0 / 256)).astype(np.float32)
m = w.reshape(1024, 256)
u, s, vt = np.linalg.svd(m, full_matrices=False)
wl = ((u[:, :40] * s[:40]) @ vt[:40]).reshape(shape).astype(np.float32)
for name, ww in (("lowrank", wl), ("gauss", w)):
    for filt in ("0", "1"):
        os.environ["TADMM_FILTER"] = filt
        t = torch.from_numpy(ww).to(dev)
        L = dict(kind=KIND_TT_CONV, W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), tt_shapes=tts, ranks=list(ranks))
        plan = ops.ProjectionPlan([L])
        plan.run(update_u=False, use_u=False)
        z = L["Z"].cpu().numpy()
        mm = ww.reshape(1024, 256).astype(np.float64)
        uu, ss, vv = np.linalg.svd(mm, full_matrices=False)
        ref = (uu[:, :75] * ss[:75]) @ vv[:75]
        print(name, "filter", filt, plan.filter_stats(), "rel err vs svd %.3e" % (np.linalg.norm(z.reshape(1024, 256) - ref) / np.linalg.norm(ref)),
              "sv", plan.singular_values(0, 0)[:3], ss[:3], flush=True)
        plan.close()
