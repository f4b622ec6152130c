"""Layer scheduling: SVD cost model and static LPT partition of layers over GPUs.

Layers of one ADMM iteration are independent (the loop at admm.py:43 has no cross-layer data
flow), so the multi-GPU path shards *layers*; the only collective on the path is the all-reduce of
the scalar residual.
"""
from __future__ import annotations

from typing import List, Sequence

from ._cabi import KIND_SVD, KIND_TT_CONV, clamp_ranks


def tt_steps(numel: int, tt_shapes: Sequence[int], ranks: Sequence[int]):
    """[(m, cols, r_keep)] of the unfoldings TT-SVD visits (ttd.py:15-23), ranks already clamped."""
    out = []
    rest = numel
    for i in range(len(tt_shapes) - 1):
        rest //= tt_shapes[i]
        out.append((ranks[i] * tt_shapes[i], rest, ranks[i + 1]))
    return out


def svd_flops(m: int, n: int) -> float:
    """Thin-SVD model 4*M*N^2 + 8*N^3 (the 'per-layer SVD GFLOP/s' numerator)."""
    M, N = max(m, n), min(m, n)
    return 4.0 * M * N * N + 8.0 * N ** 3


def layer_flops(kind, dims, tt_shapes, ranks):
    """dict(svd=.., rec=.., gram=.., proj=.., eig=..) algorithmic FLOPs of one layer projection."""
    numel = 1
    for d in dims:
        numel *= int(d)
    if kind == KIND_SVD:
        r = ranks if isinstance(ranks, int) else ranks[0]
        tt_shapes, ranks = [int(dims[0]), int(dims[1])], [1, int(r), 1]
    tt_shapes = [int(x) for x in tt_shapes]
    ranks = clamp_ranks(tt_shapes, [int(x) for x in ranks])
    svd = gram = proj = eig = 0.0
    for m, cols, r in tt_steps(numel, tt_shapes, ranks):
        svd += svd_flops(m, cols)
        N, M = min(m, cols), max(m, cols)
        gram += 2.0 * M * N * N
        proj += 2.0 * m * cols * r
        eig += 8.0 * N ** 3
    rec = 0.0
    prod = 1
    for i in range(len(tt_shapes)):                       # ttd.py:39-40 chain
        if i > 0:
            rec += 2.0 * prod * ranks[i] * tt_shapes[i] * ranks[i + 1]
        prod *= tt_shapes[i]
    return dict(svd=svd, rec=rec, gram=gram, proj=proj, eig=eig, numel=numel)


def layer_cost(kind, dims, tt_shapes, ranks) -> float:
    f = layer_flops(kind, dims, tt_shapes, ranks)
    return f["svd"] + f["rec"]


def lpt_partition(costs: Sequence[float], nparts: int) -> List[List[int]]:
    """Longest-processing-time-first: heaviest layer to the least loaded part.  Deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * nparts
    parts: List[List[int]] = [[] for _ in range(nparts)]
    for i in order:
        k = min(range(nparts), key=lambda j: (loads[j], j))
        parts[k].append(i)
        loads[k] += costs[i]
    for p in parts:
        p.sort()
    return parts
