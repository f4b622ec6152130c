import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
for M, N in ((32, 32), (32, 128), (32, 256), (32, 512), (32, 1152), (256, 128), (256, 256), (256, 1152)):
    a = torch.randn(M, N, dtype=torch.float64, device="cuda")
    g = torch.randn(N, N, dtype=torch.float64, device="cuda")
    reps = 200
    ops.dgemm3(a, g, repeats=reps); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.dgemm3(a, g, repeats=reps); e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1)
    e0.record(); ops.dgemm3(a, g, repeats=1); e1.record(); torch.cuda.synchronize()
    t0 = e0.elapsed_time(e1)
    print("M=%d N=%d: %.2f us per launch (fixed part of the call %.1f us)" % (M, N, (t1 - t0) / (reps - 1) * 1e3, t0 * 1e3), flush=True)
