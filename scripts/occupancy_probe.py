"""Does a second co-resident tick3 workgroup per CU double the tournament throughput?  N = 128 problems (LDS 63 KB per
workgroup: two fit a CU), 4 workgroups per problem and tick, one lane: time per projection against the number of problems
(GPU box).  256 workgroups = one per CU; 512 = two per CU if they co-reside."""
import os, sys, time
os.environ["TADMM_LANES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import numpy as np, torch
from tadmm import ops
from tadmm._cabi import KIND_SVD
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for N in (128, 256):
    for nprob in (16, 32, 64, 96, 128, 192):
        layers = []
        for i in range(nprob):
            w = torch.from_numpy(rng.standard_normal((N, 512)).astype(np.float32)).to(dev)
            layers.append(dict(kind=KIND_SVD, W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=N - 8))
        plan = ops.ProjectionPlan(layers)
        for _ in range(2): plan.run(update_u=False)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): plan.run(update_u=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5 * 1e3
        wgs = nprob * (N // 32) // 2
        print("N", N, "problems", nprob, "workgroups per tick", wgs, "ms per projection %.2f" % dt, "-> us per 256 workgroups of tick work %.1f" % (dt * 1e3 * 256 / max(wgs, 256)), flush=True)
        plan.close()
