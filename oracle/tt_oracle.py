"""CPU oracle for the ADMM low-rank projection path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm for the hot path
(SURVEY.md section 8a).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product (``dnn-compression-tensor-admm_amd/``) never does and
fails loudly when its HIP library is missing.

Parity status
-------------
* TT / SVD branches: **pinned** -- checked against golden vectors generated in
  the build container by importing the reference's own ``ttd.py`` / ``admm.py``
  / ``TTConv.py`` / ``TTLinear.py`` (``tests/golden/make_golden.py``,
  fixtures G1..G6 under ``tests/golden/``).
* Tucker branch: **parity unpinned** -- the arithmetic lives in the un-vendored,
  un-pinned third-party ``tensorly`` (``partial_tucker`` / ``tucker_to_tensor``),
  which is absent from the image.  ``partial_tucker`` below restates the
  published tensorly<=0.7 algorithm (HOSVD init + HOOI, n_iter_max=100,
  tol=1e-4) and is checked by invariants only.

Every function cites the reference file:line it follows (paths are relative to
the reference repository root).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# TT-SVD   (reference: ttd.py:10-31  ten2tt,  ttd.py:34-43  tt2ten)
# --------------------------------------------------------------------------


def ten2tt(x, tt_shapes, tt_ranks):
    """Sequential truncated-SVD tensor-train decomposition (ttd.py:10-31).

    ``tt_ranks`` is clamped IN PLACE when an unfolding has fewer singular
    values than the requested rank (ttd.py:18-19) -- callers rely on that side
    effect (admm.py:94,97 passes the hp_dict list itself).
    Returns a list of d cores shaped (r_i, n_i, r_{i+1}) in x's dtype.
    """
    d = len(tt_shapes)
    rest = np.asarray(x)
    cores = []
    for i in range(d - 1):
        rows = tt_ranks[i] * tt_shapes[i]
        mat = rest.reshape(rows, -1)                                   # ttd.py:16
        u, s, vt = np.linalg.svd(mat, full_matrices=False)              # ttd.py:17
        if s.shape[0] < tt_ranks[i + 1]:                               # ttd.py:18-19
            tt_ranks[i + 1] = int(s.shape[0])
        r = tt_ranks[i + 1]
        cores.append(u[:, :r].reshape(tt_ranks[i], tt_shapes[i], r))   # ttd.py:21,25
        rest = np.dot(np.diag(s[:r]), vt[:r, :])                        # ttd.py:26 (dense diag product)
    cores.append(rest.reshape(tt_ranks[d - 1], tt_shapes[d - 1], tt_ranks[d]))  # ttd.py:27-28
    return cores


def tt2ten(tt_cores, tt_shapes):
    """Chain-GEMM reconstruction of a TT tensor (ttd.py:34-43)."""
    acc = tt_cores[0]
    for core in tt_cores[1:]:
        r = core.shape[0]
        acc = np.dot(acc.reshape(-1, r), core.reshape(r, -1))          # ttd.py:39-40
    return acc.reshape(tt_shapes)                                      # ttd.py:42


# --------------------------------------------------------------------------
# ADMM projections   (reference: admm.py:91-149)
# --------------------------------------------------------------------------


def conv_unfold(z):
    """(O,I,kh,kw) -> (O, k^2, I) mode re-ordering  (admm.py:96, TTConv.py:97-99)."""
    o, i, kh, kw = z.shape
    return np.transpose(z.reshape(o, i, kh * kw), (0, 2, 1))


def conv_fold(t, kernel_shape):
    """(O, k^2, I) -> (O,I,kh,kw)  (admm.py:99)."""
    return np.transpose(t, (0, 2, 1)).reshape(kernel_shape)


def prune_conv_rank_tt(z, tt_shapes, tt_ranks, return_cores=False):
    """TT projection of a conv kernel (admm.py:91-101).  ``tt_ranks`` is mutated
    by the clamp exactly as the reference mutates ``hp_dict.ranks[name]``."""
    shape = z.shape
    t = conv_unfold(z)
    cores = ten2tt(t, tt_shapes, tt_ranks)                              # admm.py:97
    rec = tt2ten(cores, (shape[0], shape[2] * shape[3], shape[1]))      # admm.py:98
    out = conv_fold(rec, shape)
    return (out, cores) if return_cores else out


def prune_linear_rank_tt(z, tt_shapes, tt_ranks, return_cores=False):
    """TT projection of a linear weight (admm.py:103-111): no permutation and the
    rank list is COPIED (admm.py:105) so the caller's table is not clamped."""
    ranks = list(tt_ranks)
    cores = ten2tt(z.reshape(tt_shapes), tt_shapes, ranks)
    out = tt2ten(cores, z.shape)
    return (out, cores) if return_cores else out


def _svd_rank(rank_entry):
    return rank_entry if isinstance(rank_entry, int) else rank_entry[0]   # admm.py:130,142


def prune_conv_rank_svd(z, rank_entry):
    """Rank-r truncated SVD of a squeezed 1x1 kernel, re-expanded to 4-D
    (admm.py:129-139)."""
    r = _svd_rank(rank_entry)
    u, s, vt = np.linalg.svd(np.squeeze(z), full_matrices=False)
    out = u[:, :r] @ np.diag(s[:r]) @ vt[:r, :]
    return out[:, :, None, None]


def prune_linear_rank_svd(z, rank_entry):
    """Rank-r truncated SVD of a 2-D weight (admm.py:141-149)."""
    r = _svd_rank(rank_entry)
    u, s, vt = np.linalg.svd(z, full_matrices=False)
    return u[:, :r] @ np.diag(s[:r]) @ vt[:r, :]


# --------------------------------------------------------------------------
# Tucker-2   (reference: admm.py:113-127 -> tensorly partial_tucker; UNPINNED)
# --------------------------------------------------------------------------


def unfold(t, mode):
    """tensorly.unfold: mode-n matricisation with C-order of the remaining modes."""
    return np.moveaxis(t, mode, 0).reshape(t.shape[mode], -1)


def _top_left_vectors(mat, r):
    """Leading r left singular vectors (tensorly partial_svd semantics: full SVD
    when r >= min(shape), Gram + symmetric eigen-solve otherwise)."""
    u, _, _ = np.linalg.svd(mat, full_matrices=False)
    if r > u.shape[1]:
        # tensorly pads with full_matrices=True; keep what exists.
        uf, _, _ = np.linalg.svd(mat, full_matrices=True)
        return uf[:, :r]
    return u[:, :r]


def mode_dot_T(t, factor, mode):
    """t x_mode factor^T  (tensorly multi_mode_dot(..., transpose=True))."""
    moved = np.moveaxis(t, mode, 0)
    res = np.tensordot(factor.T, moved, axes=(1, 0))
    return np.moveaxis(res, 0, mode)


def mode_dot(t, factor, mode):
    moved = np.moveaxis(t, mode, 0)
    res = np.tensordot(factor, moved, axes=(1, 0))
    return np.moveaxis(res, 0, mode)


def partial_tucker(t, ranks, n_iter_max=100, tol=1e-4, return_trace=False):
    """Tucker-2 on modes (0,1): HOSVD init then HOOI (tensorly<=0.7
    ``partial_tucker(tensor, modes=[0,1], rank=ranks, init='svd')`` as called at
    admm.py:116,124 and TKConv.py:79,192,294).  Stops when two successive
    relative reconstruction errors differ by < tol, checked from the 3rd sweep.
    Returns core, [U_out, U_in]."""
    t = np.asarray(t)
    ranks = list(ranks)
    modes = (0, 1)
    factors = [_top_left_vectors(unfold(t, m), ranks[k]) for k, m in enumerate(modes)]
    norm_t = np.linalg.norm(t)
    errs = []
    for it in range(n_iter_max):
        for k, m in enumerate(modes):
            other = modes[1 - k]
            proj = mode_dot_T(t, factors[1 - k], other)
            factors[k] = _top_left_vectors(unfold(proj, m), ranks[k])
        core = mode_dot_T(mode_dot_T(t, factors[0], 0), factors[1], 1)
        err = np.sqrt(abs(norm_t ** 2 - np.linalg.norm(core) ** 2)) / norm_t
        errs.append(float(err))
        if it > 1 and abs(errs[-1] - errs[-2]) < tol:
            break
    if return_trace:
        return core, factors, errs
    return core, factors


def tucker_to_tensor(core, factors):
    """core x_0 U_out x_1 U_in  (tensorly.tucker_to_tensor with 2 factors)."""
    return mode_dot(mode_dot(core, factors[0], 0), factors[1], 1)


def prune_rank_tk(z, ranks, **kw):
    """Tucker projection for conv (admm.py:113-119) and linear (admm.py:121-127)."""
    core, factors = partial_tucker(z, ranks, **kw)
    return tucker_to_tensor(core, factors).astype(z.dtype, copy=False)


# --------------------------------------------------------------------------
# One ADMM iteration over a dict of layers   (reference: admm.py:42-78)
# --------------------------------------------------------------------------


def project_layer(zin, fmt, rank_entry, tt_shape_entry=None):
    """Dispatch of admm.py:47-69 on ndim / format / len(ranks)."""
    multi = (not isinstance(rank_entry, int)) and len(rank_entry) > 1
    if zin.ndim == 4:
        if fmt == "tk" and multi:
            return prune_rank_tk(zin, rank_entry)
        if fmt == "tt" and multi:
            return prune_conv_rank_tt(zin, tt_shape_entry, rank_entry)
        return prune_conv_rank_svd(zin, rank_entry)
    if zin.ndim == 2:
        if fmt == "tk":
            return prune_rank_tk(zin, rank_entry)
        if fmt == "tt":
            return prune_linear_rank_tt(zin, tt_shape_entry, rank_entry)
        return prune_linear_rank_svd(zin, rank_entry)
    raise Exception("ERROR: unsupported layer in ADMM!")              # admm.py:69


def admm_update(weights, u, fmt, ranks, tt_shapes=None, update_u=True):
    """Z <- proj(W+U) for every layer, then U += W-Z (admm.py:42-78).

    weights/u: dict name -> float32 ndarray (u is updated in place).
    Returns (z dict, residual dict name -> ||W-Z||_2)."""
    z, resid = {}, {}
    for name, w in weights.items():
        zin = w + u[name]                                              # admm.py:45
        tts = None if tt_shapes is None else tt_shapes.get(name)
        z[name] = np.ascontiguousarray(project_layer(zin, fmt, ranks[name], tts)).astype(np.float32, copy=False)
        if update_u:
            diff = w - z[name]                                         # admm.py:73
            u[name] += diff                                            # admm.py:74
            resid[name] = float(np.linalg.norm(diff))                  # admm.py:76
    return z, resid


def admm_penalty(weights, z, u, rho):
    """0.5*rho*sum ||W-Z+U||^2 and its gradient rho*(W-Z+U) (admm.py:80-85)."""
    loss = 0.0
    grads = {}
    for name, w in weights.items():
        d = w - z[name] + u[name]
        loss += 0.5 * rho * float(np.sum(d.astype(np.float64) ** 2))
        grads[name] = (rho * d).astype(np.float32)
    return loss, grads


# --------------------------------------------------------------------------
# Factorised forward contractions
# --------------------------------------------------------------------------


def split_tt_modes(tt_shapes, out_dim, conv):
    """Out / (k^2) / in mode split by cumulative product == out_dim
    (TTConv.py:49-59, TTLinear.py:31-40)."""
    prod = 1
    for i, n in enumerate(tt_shapes):
        prod *= n
        if prod == out_dim:
            n_out = i + 1
            break
    else:
        raise AssertionError("tt_shapes do not factor out_dim")
    out_modes = list(tt_shapes[:n_out])
    in_modes = list(tt_shapes[n_out + 1:]) if conv else list(tt_shapes[n_out:])
    return out_modes, in_modes


def ttlinear_m_forward(x, cores, out_features, bias=None):
    """TTLinearM.forward (TTLinear.py:75-93; same algebra as numeric_example3.py:40-61).
    cores: list of (r_i, n_i, r_{i+1}); x: (..., in_features)."""
    shapes = [c.shape[1] for c in cores]
    ranks = [c.shape[0] for c in cores] + [cores[-1].shape[2]]
    out_modes, in_modes = split_tt_modes(shapes, out_features, conv=False)
    n_out, n_in = len(out_modes), len(in_modes)
    out = x
    for i in range(n_in - 1, -1, -1):                                  # TTLinear.py:79-82
        k = in_modes[i] * ranks[i + n_out + 1]
        out = (cores[i + n_out].reshape(-1, k) @ out.reshape(-1, k).T).T
    for i in range(n_out - 1, -1, -1):                                 # TTLinear.py:84-86
        out = cores[i].reshape(-1, ranks[i + 1]) @ out.reshape(-1, ranks[i + 1]).T
        out = out.reshape(ranks[i], -1).T
    out = out.reshape(out_features, -1).T.reshape(list(x.shape[:-1]) + [out_features])  # :88
    if bias is not None:
        out = out + bias
    return out


def tt_recover_weight(cores, out_dim, in_dim):
    """TTLinearR._recover_weight (TTLinear.py:151-157) == tt2ten to (out,in)."""
    return tt2ten(cores, (out_dim, in_dim))


def conv2d_nchw(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """Plain direct convolution (cross-correlation) reference, NCHW / OIHW, float64
    accumulate.  Small shapes only."""
    b, c, h, wd = x.shape
    o, i, kh, kw = w.shape
    assert i == c
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    xp = np.zeros((b, c, h + 2 * ph, wd + 2 * pw), dtype=np.float64)
    xp[:, :, ph:ph + h, pw:pw + wd] = x
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    out = np.zeros((b, o, ho, wo), dtype=np.float64)
    w64 = w.astype(np.float64)
    for a in range(kh):
        for bb in range(kw):
            patch = xp[:, :, a * dh:a * dh + sh * ho:sh, bb * dw:bb * dw + sw * wo:sw]
            out += np.einsum("bchw,oc->bohw", patch, w64[:, :, a, bb])
    return out.astype(x.dtype)


def ttconv2d_m_forward(x, in_cores, core_kernel, out_cores, bias=None,
                       stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """TTConv2dM.forward (TTConv.py:130-153): input-core chain on NHWC -> small
    k x k conv -> output-core chain."""
    b, c, h, w = x.shape
    out = np.transpose(x, (0, 2, 3, 1))
    for core in reversed(in_cores):                                    # TTConv.py:133-136
        r0, n, r1 = core.shape
        out = (core.reshape(r0, n * r1) @ out.reshape(-1, n * r1).T).T
    r_in0 = in_cores[0].shape[0] if in_cores else c
    out = np.transpose(out.reshape(b, h, w, r_in0), (0, 3, 1, 2))      # TTConv.py:137
    out = conv2d_nchw(out, core_kernel, stride, padding, dilation)     # TTConv.py:139
    _, _, h2, w2 = out.shape
    out = np.transpose(out, (0, 2, 3, 1))
    for core in reversed(out_cores):                                   # TTConv.py:143-147
        r0, n, r1 = core.shape
        out = core.reshape(r0 * n, r1) @ out.reshape(-1, r1).T
        out = out.reshape(r0, -1).T
    o = int(np.prod([cc.shape[1] for cc in out_cores]))
    out = np.transpose(out.reshape(o, b, h2, w2), (1, 0, 2, 3))        # TTConv.py:149
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1)
    return out


def tkconv2d_forward(x, first, core, last, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """TKConv2dC / TKConv2dM forward (TKConv.py:93-98, :210-214): 1x1 (I->r_in) ->
    k x k (r_in->r_out) -> 1x1 (r_out->O).  first: (r_in, I), last: (O, r_out)."""
    h = np.einsum("bchw,rc->brhw", x, first)
    h = conv2d_nchw(h, core, stride, padding, dilation)
    y = np.einsum("brhw,or->bohw", h, last)
    if bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    return y


def tklinear_forward(x, first, core, last, bias=None):
    """TKLinearM.forward (TKLinear.py:66-71): x first^T core^T last^T + b."""
    y = ((x @ first.T) @ core.T) @ last.T
    if bias is not None:
        y = y + bias
    return y


# --------------------------------------------------------------------------
# Gauge helpers for per-factor comparison (SVD vectors are sign-ambiguous)
# --------------------------------------------------------------------------


def gauge_align_tt(cores, ref_cores):
    """Flip signs of bond indices so ``cores`` matches ``ref_cores`` as closely as
    possible.  A sign flip on bond j multiplies column j of core_i (last axis)
    and row j of core_{i+1} (first axis) by -1 and leaves the tensor unchanged."""
    out = [np.array(c, copy=True) for c in cores]
    for i in range(len(out) - 1):
        a = out[i].reshape(-1, out[i].shape[2])
        b = ref_cores[i].reshape(-1, ref_cores[i].shape[2])
        sign = np.sign(np.sum(a * b, axis=0))
        sign[sign == 0] = 1.0
        out[i] = out[i] * sign[None, None, :]
        out[i + 1] = out[i + 1] * sign[:, None, None]
    return out
