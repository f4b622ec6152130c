#!/bin/bash
# Rebuilds the library with cycle stamps in the fp64 tile GEMM (GPU box, scratch copy only) and dumps one workgroup's timeline.
cd $GRAFT_REPO_ROOT/dnn-compression-tensor-admm_amd/csrc && touch dgemm.hip && make CXXEXTRA=-DTADMM_DGEMM_STAMPS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && TADMM_DGEMM_STAMPS_DUMP=1 python3 - <<'PY'
import os, sys
sys.path.insert(0, "dnn-compression-tensor-admm_amd")
import torch
from tadmm import ops
a = torch.randn(192, 512, dtype=torch.float64, device="cuda")
b = torch.randn(512, 512, dtype=torch.float64, device="cuda")
for _ in range(4):
    ops.dgemm(a, b, True)
torch.cuda.synchronize()
PY
