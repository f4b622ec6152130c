"""Launch duration of the fp64 tile GEMM against K at the filter's shape (M = 192 block rows, N = 512): what a launch costs
before and after its MFMAs.  Run under rocprofv3 --kernel-trace; prints the dispatch durations in order."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
for K in (32, 64, 128, 256, 512, 1024):
    a = torch.randn(192, K, dtype=torch.float64, device="cuda")
    b = torch.randn(512, K, dtype=torch.float64, device="cuda")
    for _ in range(6):
        ops.dgemm(a, b, True)
torch.cuda.synchronize()
