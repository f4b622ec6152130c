"""TT-SVD on the MI355X behind the reference's `ttd` API (ttd.py:10-43).

``ten2tt(x, tt_shapes, tt_ranks)`` and ``tt2ten(tt_cores, tt_shapes)`` keep the reference's
signatures and side effects (the in-place rank clamp of ttd.py:18-19) but run the decomposition
as Gram (fp64 MFMA) + Jacobi eigen-solve + fp32 MFMA contractions in libtadmm_hip.so.  numpy in ->
numpy out (as the reference); torch device tensors in -> torch device tensors out.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import ops
from ._cabi import KIND_TT_LINEAR, TadmmError


def _device():
    if not torch.cuda.is_available():
        raise TadmmError(-3, "no HIP device: the TT-SVD runs on the MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def tt_decompose(w: torch.Tensor, tt_shapes: Sequence[int], tt_ranks: Sequence[int], kind=KIND_TT_LINEAR):
    """Device TT-SVD of one weight.  Returns (cores as device tensors, clamped ranks, Z, plan)."""
    w = w.detach().contiguous().float()
    u = torch.zeros_like(w)
    z = torch.empty_like(w)
    plan = ops.ProjectionPlan([dict(kind=kind, W=w, U=u, Z=z, tt_shapes=list(tt_shapes), ranks=list(tt_ranks))],
                              want_cores=True)
    plan.run(update_u=False, use_u=False)
    cores = [c.clone() for c in plan.core_tensors(0)]
    return cores, plan.ranks[0], z, plan


def ten2tt(x, tt_shapes, tt_ranks) -> List:
    """Reference ttd.py:10-31.  `tt_ranks` is clamped in place when it is a mutable list."""
    is_np = isinstance(x, np.ndarray)
    dev = _device() if is_np or not x.is_cuda else x.device
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if is_np else x
    n = 1
    for s in tt_shapes:
        n *= int(s)
    if xt.numel() != n:
        raise ValueError(f"cannot reshape array of size {xt.numel()} into shape {list(tt_shapes)}")
    flat = xt.to(dev).contiguous().reshape(int(tt_shapes[0]), -1)   # any 2-D row-major view works
    cores, ranks, _, _ = tt_decompose(flat, tt_shapes, tt_ranks, KIND_TT_LINEAR)
    if isinstance(tt_ranks, list):
        tt_ranks[:] = ranks                          # ttd.py:18-19 side effect
    if is_np:
        return [c.cpu().numpy() for c in cores]
    return cores


def tt2ten(tt_cores, tt_shapes):
    """Reference ttd.py:34-43: chain of GEMMs on the fp32 matrix cores."""
    is_np = isinstance(tt_cores[0], np.ndarray)
    dev = _device() if is_np else tt_cores[0].device
    cs = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).to(dev) if is_np else c.contiguous()
          for c in tt_cores]
    acc = cs[0].reshape(-1, cs[0].shape[-1])
    for c in cs[1:]:
        r = c.shape[0]
        acc = ops.mm(acc.reshape(-1, r), c.reshape(r, -1))
    out = acc.reshape(tuple(tt_shapes))
    return out.cpu().numpy() if is_np else out
