"""Experiment: Jacobi sweeps with X = G versus X = Cholesky factor of G (standalone eigh entry point)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (N, M) in [(512, 945), (480, 4608), (512, 2048), (256, 2304), (384, 1536)]:
    A = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
    G = A @ A.T
    ev, vec, s1 = ops.eigh(G.contiguous())
    L = torch.linalg.cholesky(G)
    ev2, vec2, s2 = ops.eigh(L.T.contiguous())       # row j of the input image = column j of X = L
    lam_ref = torch.linalg.eigvalsh(G).flip(0)
    e1 = float(((ev - lam_ref).abs() / lam_ref).max())
    e2 = float(((ev2 ** 2 - lam_ref).abs() / lam_ref).max())
    # eigenvectors: rows of vec2 are left singular vectors of L = eigenvectors of G
    r1 = float((G @ vec.T - vec.T * ev).norm() / G.norm())
    r2 = float((G @ vec2.T - vec2.T * ev2 ** 2).norm() / G.norm())
    print(f"N={N} M={M}: sweeps X=G {s1}  X=L {s2}   eig rel err {e1:.1e} / {e2:.1e}   residual {r1:.1e} / {r2:.1e}")
