"""Tucker-2 projection on the MI355X (reference: admm.py:113-127 -> tensorly `partial_tucker`).

PARITY UNPINNED: the reference's arithmetic for this branch lives in the un-vendored, un-pinned
third-party `tensorly`, absent from the image.  This module restates the published tensorly<=0.7
algorithm -- HOSVD initialisation (truncated SVD of the mode-0 and mode-1 unfoldings) followed by HOOI
sweeps, n_iter_max=100, tol=1e-4 on successive relative reconstruction errors, checked from the third
sweep -- and is verified against the oracle's float64 restatement and by invariants.

tensorly's own truncated SVD (`partial_svd`) already takes the Gram + symmetric eigen-solve route; here
the Gram runs on the fp64 matrix cores, the eigen-solve is the block-Jacobi solver and every mode product
is a strided GEMM on the fp32 matrix cores.  The data-dependent stopping rule needs one scalar on the
host per sweep.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops


def _leading_left_vectors(mat: torch.Tensor, r: int) -> torch.Tensor:
    """Top-r left singular vectors of a float32 (m,n) device matrix -> (m, r) float32."""
    m, n = mat.shape
    mat = mat.contiguous()
    G = ops.gram(mat).contiguous()
    ev, vec, _ = ops.eigh(G)                   # rows = eigenvectors, descending
    k = min(r, vec.shape[0])
    if m <= n:
        U = vec[:k].t().float()
    else:
        V = vec[:k].t().float().contiguous()   # (n, k) right singular vectors
        sig = ev[:k].clamp_min(0).sqrt().float()
        U = ops.mm(mat, V) / sig.clamp_min(1e-30)
    if k < r:
        # more vectors requested than the unfolding has singular values (tensorly pads with an arbitrary
        # orthonormal completion via full_matrices=True).  The completion multiplies zero rows of the core,
        # so Z does not depend on it: pad with zero columns.
        U = torch.cat([U, torch.zeros(m, r - k, dtype=U.dtype, device=U.device)], dim=1)
    return U.contiguous()


def _unfold0(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(t.shape[0], -1)


def _unfold1(t: torch.Tensor) -> torch.Tensor:
    return t.transpose(0, 1).reshape(t.shape[1], -1)


def _mode0_T(t: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """t x_0 f^T : (O, ...) -> (r, ...)"""
    out = ops.mm(f.t(), t.reshape(t.shape[0], -1))
    return out.reshape(f.shape[1], *t.shape[1:])


def _mode1_T(t: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """t x_1 f^T : (O, I, ...) -> (O, r, ...)"""
    o, i = t.shape[0], t.shape[1]
    rest = t.numel() // (o * i)
    tt = t.reshape(o, i, rest).permute(0, 2, 1).reshape(-1, i)          # (O*rest, I)
    out = ops.mm(tt.contiguous(), f)                                    # (O*rest, r)
    return out.reshape(o, rest, f.shape[1]).permute(0, 2, 1).reshape(o, f.shape[1], *t.shape[2:]).contiguous()


def partial_tucker(w: torch.Tensor, ranks: Sequence[int], n_iter_max: int = 100, tol: float = 1e-4):
    """Tucker-2 over modes (0,1).  Returns core (r_out, r_in, ...), [U_out (O,r_out), U_in (I,r_in)], errors.

    Runs on the batched device plan (csrc/tucker_plan.hip) with a single layer; `errors` holds the final relative
    reconstruction error repeated once per HOOI sweep the layer ran (the per-sweep history stays on the device).
    """
    core, factors, errs, _ = _plan_decompose([w], [ranks], n_iter_max, tol)[0]
    return core, factors, errs


def _plan_decompose(ws, ranks_list, n_iter_max: int = 100, tol: float = 1e-4):
    """HOSVD + HOOI of several weights in ONE grouped plan.  -> [(core, [U_out, U_in], errs, Z)]"""
    layers = []
    for w, r in zip(ws, ranks_list):
        w = w.detach().float().contiguous()
        layers.append(dict(W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=[int(r[0]), int(r[1])]))
    plan = ops.TuckerPlan(layers, n_iter_max=n_iter_max, tol=tol)
    plan.run(update_u=False, use_u=False)
    its, errs = plan.iterations()
    out = []
    for i, L in enumerate(layers):
        core, u_out, u_in = plan.factors(i)
        out.append((core, [u_out, u_in], [errs[i]] * max(1, its[i]), L["Z"]))
    plan.close()
    return out


def partial_tucker_eager(w: torch.Tensor, ranks: Sequence[int], n_iter_max: int = 100, tol: float = 1e-4):
    """The same algorithm driven step by step from Python on the Gram / eigh / GEMM primitives (one layer, one
    host round trip per sweep).  Kept as an independent cross-check of the batched plan in the tests."""
    w = w.detach().float().contiguous()
    r_out, r_in = int(ranks[0]), int(ranks[1])
    factors = [_leading_left_vectors(_unfold0(w), r_out), _leading_left_vectors(_unfold1(w), r_in)]
    norm_w = float(torch.linalg.vector_norm(w.double()))
    errs: List[float] = []
    core = None
    for it in range(n_iter_max):
        proj = _mode1_T(w, factors[1])
        factors[0] = _leading_left_vectors(_unfold0(proj), r_out)
        proj = _mode0_T(w, factors[0])
        factors[1] = _leading_left_vectors(_unfold1(proj), r_in)
        core = _mode1_T(proj, factors[1])
        nc = float(torch.linalg.vector_norm(core.double()))
        errs.append(float(np.sqrt(abs(norm_w ** 2 - nc ** 2)) / norm_w))
        if it > 1 and abs(errs[-1] - errs[-2]) < tol:
            break
    return core, factors, errs


def tucker_to_tensor(core: torch.Tensor, factors) -> torch.Tensor:
    """core x_0 U_out x_1 U_in"""
    u_out, u_in = factors
    t = ops.mm(u_out, core.reshape(core.shape[0], -1)).reshape(u_out.shape[0], *core.shape[1:])
    return _mode1_T(t, u_in.t().contiguous())


def project(w: torch.Tensor, ranks) -> Tuple[torch.Tensor, List[float]]:
    _, _, errs, z = _plan_decompose([w], [ranks])[0]
    return z.reshape(w.shape), errs


def project_numpy(z: np.ndarray, ranks, device) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32)).to(device)
    out, _ = project(t, ranks)
    return out.cpu().numpy()


class TuckerProjector:
    """The 'tk' branch of ADMM.update for the layers named in `names` (admm.py:47-50, :59-62): all of them in ONE
    grouped device plan, W/U/Z captured by pointer."""

    def __init__(self, admm, names):
        self.admm = admm
        self.names = list(names)
        a = admm
        params = dict(a._named())
        layers = [dict(W=params[n].data, U=a.u[n], Z=a.z[n], ranks=list(a.hp_dict.ranks[n])) for n in self.names]
        self.plan = ops.TuckerPlan(layers)

    def run(self, update_u: bool) -> Dict[str, float]:
        r = self.plan.run(update_u=update_u, use_u=True)
        if not update_u:
            return {}
        vals = r.sqrt().cpu().tolist()
        return {n: vals[i] for i, n in enumerate(self.names)}
