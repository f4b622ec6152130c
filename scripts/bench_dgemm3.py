"""fp64 tile GEMM vs the three-plane bf16 product (dgemm3) at the filter's shapes: accuracy and time per launch."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
torch.manual_seed(0)
for M, N in ((256, 1152), (192, 480), (192, 512), (256, 576), (64, 256)):
    a = torch.randn(M, N, dtype=torch.float64, device="cuda")
    w = torch.randn(N, 2 * N, dtype=torch.float64, device="cuda")
    g = (w @ w.t()).contiguous()
    ref = a @ g.t()
    c3 = ops.dgemm3(a, g)
    c64 = ops.dgemm(a, g, True)
    err3 = ((c3 - ref).abs().max() / ref.abs().max()).item()
    err64 = ((c64 - ref).abs().max() / ref.abs().max()).item()
    reps = 50
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.dgemm3(a, g, repeats=reps); torch.cuda.synchronize()
    e0.record(); ops.dgemm3(a, g, repeats=reps); e1.record(); torch.cuda.synchronize()
    t3 = e0.elapsed_time(e1) / reps * 1e3
    for _ in range(3): ops.dgemm(a, g, True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): ops.dgemm(a, g, True)
    e1.record(); torch.cuda.synchronize()
    t64 = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * M * N * N
    print("M=%d N=%d: dgemm3 %.1f us (%.1f TF alg, err %.1e) | fp64 tile (incl. host setup) %.1f us (err %.1e)" %
          (M, N, t3, fl / t3 / 1e6, err3, t64, err64), flush=True)
