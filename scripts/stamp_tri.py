"""Cycle stamps of the direct small eigen-solver (csrc/tridiag.hip built with -DTADMM_TRI_STAMPS as
libtadmm_hip_stamp.so): python3 scripts/stamp_tri.py  (GPU box; TADMM_LIB picks the stamped library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
os.environ["TADMM_TRI_STAMPS_DUMP"] = "1"
import numpy as np
import torch
from tadmm import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for N, M in ((32, 4608), (30, 200), (64, 576), (48, 300), (16, 100)):
    a = rng.standard_normal((N, M))
    G = torch.from_numpy(a @ a.T).to(dev)
    for rep in range(2):
        ev, vec, sweeps = ops.eigh(G)
    ref = np.linalg.eigvalsh(a @ a.T)[::-1]
    print("N", N, "sweeps(0 = direct)", sweeps, "max eval err", np.abs(ev.cpu().numpy() - ref).max() / ref[0], flush=True)
    sys.stderr.flush()
