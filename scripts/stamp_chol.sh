#!/bin/bash
# Rebuilds the library with cycle stamps in chol_factor_kernel (GPU box, scratch copy only) and prints one factorisation's timeline.
cd $GRAFT_REPO_ROOT/dnn-compression-tensor-admm_amd/csrc && touch chol.hip && make CXXEXTRA=-DTADMM_CHOL_STAMPS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && TADMM_CHOL_STAMPS_DUMP=1 python3 - <<'PY'
import sys
sys.path.insert(0, "dnn-compression-tensor-admm_amd")
import torch
from tadmm import ops
y = torch.randn(192, 512, dtype=torch.float64, device="cuda")
for _ in range(3):
    ops.cholqr_(y.clone())
torch.cuda.synchronize()
PY
