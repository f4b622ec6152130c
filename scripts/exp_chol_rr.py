"""Experiment: Jacobi sweeps on the Rayleigh-Ritz sizes with X = H versus X = Cholesky factor of H (standalone eigh entry)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for N in (128, 192, 224):
    for kind in ("wishart4", "wishart1.5", "decay"):
        if kind.startswith("wishart"):
            M = int(N * float(kind[7:]))
            A = torch.randn(N, M, generator=g, dtype=torch.float64).to(dev)
            H = A @ A.T
        else:
            Q, _ = torch.linalg.qr(torch.randn(N, N, generator=g, dtype=torch.float64).to(dev))
            lam = torch.exp(-4.0 * torch.arange(N, dtype=torch.float64, device=dev) / N)
            H = (Q * lam) @ Q.T
            H = 0.5 * (H + H.T)
        ev, vec, s1 = ops.eigh(H.contiguous())
        L = torch.linalg.cholesky(H)
        ev2, vec2, s2 = ops.eigh(L.T.contiguous())       # row j of the input image = column j of X = L
        # pivoted variant: order the diagonal descending first
        p = torch.argsort(torch.diagonal(H), descending=True)
        Hp = H[p][:, p]
        Lp = torch.linalg.cholesky(Hp)
        ev3, vec3, s3 = ops.eigh(Lp.T.contiguous())
        ref = torch.linalg.eigvalsh(H).flip(0)
        e2 = float(((ev2 ** 2 - ref).abs() / ref).max())
        print(f"N={N} {kind}: sweeps X=H {s1}  X=L {s2}  X=L(diag-sorted) {s3}   eig rel err (L) {e2:.1e}")
