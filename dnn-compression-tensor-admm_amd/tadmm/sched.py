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


# ---------------------------------------------------------------------------------------------------------------
# Latency-aware partition.  One layer's chain of eigen-solves is latency-bound (a Jacobi tournament is a chain of
# dependent launches whatever the population of the launch), and layers that share a rank share those launches:
# the time of a rank is  sum over levels of the SLOWEST problem of the level  + a throughput term, not a sum of
# per-layer FLOPs.  An LPT split by FLOPs therefore stacks the three `layer4.x.conv2` chains of ResNet-50 beside
# other work although each of them alone sets the iteration time.
# Constants are the round-2 measurements on one MI355X (profiles/r02_*): microseconds.
# ---------------------------------------------------------------------------------------------------------------
_T_TICK_BASE, _T_TICK_PER_COL = 9.0, 0.02     # one tournament launch: ~13 us at 256-long columns, ~19 us at 512
_T_SMALL = 100.0                              # whole eigen-solve of a problem with <= 64 columns (one launch)
_T_PRODUCT, _T_CHOLQR = 35.0, 170.0           # filtered path: one block product / one Gram + factor + solve
_FILTER_STAGES, _FILTER_DEGREE = 6, 8
_SWEEPS_FULL, _SWEEPS_RR = 11, 7
_THROUGHPUT_FLOPS = 40e12                     # Gram / projection / chain GEMMs (grouped, MFMA)


def filter_block_size(N: int, r: int) -> int:
    """Mirror of csrc/filter_host.h: iteration block of the filtered eigen-solver (0: full solve)."""
    if N < 192 or r < 16:
        return 0
    rp = -(-int(1.45 * r + 0.999) // 32) * 32
    if rp > 256:                      # capped block: still filtered while it keeps >= 1.15 of oversampling
        rp = 256
        if rp < 1.15 * r:
            return 0
    if rp * 100 > 56 * N or rp <= r:
        return 0
    return rp


def problem_latency_us(N: int, r: int) -> float:
    """Modelled latency of one eigen-problem (Gram matrix of size N, keep r) when it is alone on the device."""
    npad = -(-N // 32) * 32
    if npad <= 64:
        return _T_SMALL
    rp = filter_block_size(N, r)
    if rp:
        tick = _T_TICK_BASE + _T_TICK_PER_COL * (-(-rp // 128) * 128)
        return (_FILTER_STAGES + 1) * (_FILTER_DEGREE * _T_PRODUCT + _T_CHOLQR) + (rp // 16 - 1) * _SWEEPS_RR * tick
    tick = _T_TICK_BASE + _T_TICK_PER_COL * (-(-N // 128) * 128)
    return (npad // 16 - 1) * _SWEEPS_FULL * tick


def layer_latency_profile(kind, dims, tt_shapes, ranks, skip_identity: bool = True):
    """(sorted step latencies in us, throughput-bound seconds) of one layer's projection."""
    numel = 1
    for d in dims:
        numel *= int(d)
    if kind == KIND_SVD:
        r = ranks if isinstance(ranks, int) else ranks[0]
        tt_shapes, ranks = [int(dims[0]), int(dims[1])], [1, int(r), 1]
    tt_shapes = [int(x) for x in tt_shapes]
    ranks = clamp_ranks(tt_shapes, [int(x) for x in ranks])
    lat = []
    for m, cols, r in tt_steps(numel, tt_shapes, ranks):
        if skip_identity and m <= cols and r == m:        # identity step of the Z-only mode: no eigen-solve
            continue
        lat.append(problem_latency_us(min(m, cols), r))
    f = layer_flops(kind, dims, tt_shapes, ranks)
    return sorted(lat, reverse=True), (f["gram"] + f["proj"] + f["rec"]) / _THROUGHPUT_FLOPS


def rank_time_us(profiles) -> float:
    """Modelled iteration time of a rank holding the layers with these profiles: levels are shared."""
    depth = max((len(p[0]) for p in profiles), default=0)
    t = 0.0
    for d in range(depth):
        t += max((p[0][d] for p in profiles if len(p[0]) > d), default=0.0)
    return t + 1e6 * sum(p[1] for p in profiles)


def latency_partition(profiles, nparts: int) -> List[List[int]]:
    """Greedy min-max assignment under `rank_time_us`: longest chains first, each to the rank whose modelled time
    grows least (ties: the currently shortest rank).  Deterministic."""
    order = sorted(range(len(profiles)), key=lambda i: (-sum(profiles[i][0]) - 1e6 * profiles[i][1], i))
    parts: List[List[int]] = [[] for _ in range(nparts)]
    times = [0.0] * nparts
    for i in order:
        best, best_key = 0, None
        for k in range(nparts):
            t_new = rank_time_us([profiles[j] for j in parts[k]] + [profiles[i]])
            key = (max(t_new, max(times[:k] + times[k + 1:], default=0.0)), t_new, k)
            if best_key is None or key < best_key:
                best, best_key = k, key
        parts[best].append(i)
        times[best] = rank_time_us([profiles[j] for j in parts[best]])
    for p in parts:
        p.sort()
    return parts
