"""DeiT-like tournament (N = 288 kept 256, ld = 384, 885 KB image per problem) against the number of problems in flight:
is the level bound by the L2 capacity of an XCD (4 MB: ~4 images)?  One lane and two lanes (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import numpy as np, torch
lanes = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["TADMM_LANES"] = lanes
from tadmm import ops
from tadmm._cabi import KIND_SVD
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for nprob in (8, 16, 24, 32, 48):
    layers = []
    for i in range(nprob):
        w = torch.from_numpy(rng.standard_normal((288, 384)).astype(np.float32)).to(dev)
        layers.append(dict(kind=KIND_SVD, W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=256))
    plan = ops.ProjectionPlan(layers)
    for _ in range(2): plan.run(update_u=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): plan.run(update_u=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5 * 1e3
    print("lanes", lanes, "problems", nprob, "workgroups per tick", nprob * 9 // 2 * 1, "ms per projection %.2f" % dt, "ms per problem %.3f" % (dt / nprob), flush=True)
    plan.close()
