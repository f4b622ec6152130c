"""Micro-benchmark of the fp64 matrix-core GEMM kernels (tadmm_dgemm_f64); the entry's upload + sync overhead is
measured on a minimal problem and subtracted."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dnn-compression-tensor-admm_amd"))
from tadmm import ops
dev = torch.device("cuda:0")
def timeit(M, N, K, n=20):
    a = torch.randn(M, K, dtype=torch.float64, device=dev)
    b = torch.randn(N, K, dtype=torch.float64, device=dev)
    for _ in range(3): ops.dgemm(a, b, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); ops.dgemm(a, b, True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]
base = timeit(64, 32, 32)
print("entry overhead %.1f us" % base)
for (M, N, K) in [(64, 32, 512), (64, 64, 512), (224, 512, 512), (224, 512, 2048), (2048, 4096, 512), (4096, 4096, 4096)]:
    t = timeit(M, N, K) - base
    print(f"M={M} N={N} K={K}: {t:.1f} us  {2*M*N*K/t/1e6:.2f} TFLOP/s")
