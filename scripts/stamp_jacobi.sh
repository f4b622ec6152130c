#!/bin/bash
# Rebuilds the library with cycle stamps in the Jacobi tick kernels (GPU box, scratch copy only) and prints the per-launch
# phase times of jacobi_tick3_kernel for the Rayleigh-Ritz shape (N = 192, rows of 256).
cd $GRAFT_REPO_ROOT/dnn-compression-tensor-admm_amd/csrc && touch jacobi.hip plan.hip && make CXXEXTRA=-DTADMM_STAMPS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && TADMM_STAMPS_DUMP=1 python3 - <<'PY'
import sys
sys.path.insert(0, "dnn-compression-tensor-admm_amd")
import torch
from tadmm import ops
for N in (192, 128):
    a = torch.randn(N, 4 * N, dtype=torch.float64, device="cuda")
    g = a @ a.T
    ev, vec, sweeps = ops.eigh(g)
    print("N", N, "sweeps", sweeps, file=sys.stderr)
PY
