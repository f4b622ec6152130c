"""GPU tests of the filtered eigen-solver (csrc/dgemm.hip, chol.hip, filter.hip): its building blocks against
torch fp64, the filtered projection against the full Jacobi solve and the oracle, and the fall-back path."""
import os

import numpy as np
import pytest
import torch

from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(32, 32, 16), (224, 512, 512), (96, 480, 224), (192, 192, 480)])
def test_dgemm_nt_and_nn_match_torch_fp64(dev, shape):
    from tadmm import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.float64, generator=g).to(dev)
    bt = torch.randn(N, K, dtype=torch.float64, generator=g).to(dev)
    ref = a @ bt.t()
    got = ops.dgemm(a, bt, True)
    assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) * np.sqrt(K)
    b = bt.t().contiguous()
    got2 = ops.dgemm(a, b, False)
    assert float((got2 - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) * np.sqrt(K)


@pytest.mark.parametrize("n,ncols,cond", [(32, 64, 1e2), (128, 256, 1e4), (224, 512, 1e6), (192, 512, 1e3), (256, 512, 10.0)])
def test_cholqr_orthonormalises_and_keeps_the_span(dev, n, ncols, cond):
    from tadmm import ops
    rng = np.random.default_rng(n + ncols)
    q1, _ = np.linalg.qr(rng.standard_normal((ncols, n)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.exp(np.linspace(0.0, -np.log(cond), n))
    y = (q1 * s) @ q2.T                                   # ncols x n, condition number `cond`
    yt = torch.from_numpy(np.ascontiguousarray(y.T)).to(dev)
    assert ops.cholqr_(yt)
    q = yt.cpu().numpy().T
    err1 = np.abs(q.T @ q - np.eye(n)).max()
    assert err1 <= 1e-13 * cond * cond + 1e-12, err1      # one pass: ~cond^2 * eps
    assert ops.cholqr_(yt)
    q = yt.cpu().numpy().T
    assert np.abs(q.T @ q - np.eye(n)).max() <= 1e-13
    # same column space: projecting the original block onto span(Q) reproduces it
    assert np.linalg.norm(y - q @ (q.T @ y)) <= 1e-9 * np.linalg.norm(y)


def test_cholqr_reports_rank_deficiency(dev):
    from tadmm import ops
    rng = np.random.default_rng(0)
    y = rng.standard_normal((256, 40)) @ rng.standard_normal((40, 64))      # rank 40 < 64 columns
    yt = torch.from_numpy(np.ascontiguousarray(y.T)).to(dev)
    assert not ops.cholqr_(yt)


def _layers(dev, rng, lowrank=None):
    """a 3x3 conv (N = 240 / 256 Gram matrices, kept rank 82) and a 1x1 conv (N = 256, kept rank 75)"""
    from tadmm._cabi import KIND_TT_CONV
    specs = [((256, 256, 3, 3), [16, 16, 9, 16, 16], [1, 15, 82, 82, 15, 1]),
             ((1024, 256, 1, 1), [1024, 1, 256], [1, 75, 75, 1])]
    layers, ws = [], []
    for shape, tts, ranks in specs:
        fan_in = int(np.prod(shape[1:]))
        w = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        if lowrank is not None:
            m = w.reshape(shape[0], -1)
            u, s, vt = np.linalg.svd(m, full_matrices=False)
            w = ((u[:, :lowrank] * s[:lowrank]) @ vt[:lowrank]).reshape(shape).astype(np.float32)
        ws.append(w)
        t = torch.from_numpy(w).to(dev)
        layers.append(dict(kind=KIND_TT_CONV, W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), tt_shapes=tts,
                           ranks=list(ranks)))
    return specs, layers, ws


def _run(layers, filt, monkeypatch, want_cores=False, sv_steps=(1, 2)):
    from tadmm import ops
    monkeypatch.setenv("TADMM_FILTER", "1" if filt else "0")
    plan = ops.ProjectionPlan(layers, want_cores=want_cores)
    plan.run(update_u=False, use_u=False)
    z = [L["Z"].cpu().numpy().copy() for L in layers]
    stats = plan.filter_stats()
    sv = [plan.singular_values(0, k) for k in sv_steps]
    plan.close()
    return z, stats, sv


def test_filtered_projection_matches_full_solve_and_oracle(dev, monkeypatch):
    rng = np.random.default_rng(3)
    specs, layers, ws = _layers(dev, rng)
    zf, st_f, sv_f = _run(layers, True, monkeypatch)
    zn, st_n, sv_n = _run(layers, False, monkeypatch)
    assert st_n["eligible"] == 0 and st_f["eligible"] == 3 and st_f["solves"] == 3, (st_f, st_n)
    assert st_f["fallbacks"] == 0, st_f
    for a, b in zip(zf, zn):
        assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
    for a, b in zip(sv_f, sv_n):
        np.testing.assert_allclose(a, b, rtol=1e-9)
    for (shape, tts, ranks), w, z in zip(specs, ws, zf):
        ref = O.prune_conv_rank_tt(w, tts, list(ranks))
        assert np.linalg.norm(z - ref) <= 1e-5 * np.linalg.norm(ref)


@pytest.mark.parametrize("tm", ["32", "64"])
def test_product_tile_heights_agree(dev, monkeypatch, tm):
    """The filter products run on 32 x 32 tiles (four K quarters per workgroup) for latency-bound levels of narrow blocks and
    on 64 x 32 tiles otherwise (csrc/filter_host.h: filter_tile_m); TADMM_FILTER_TM forces one.  Both against the full solve
    and the oracle on the same layers, no fallback either way."""
    rng = np.random.default_rng(3)
    specs, layers, ws = _layers(dev, rng)
    monkeypatch.setenv("TADMM_FILTER_TM", tm)
    zf, st_f, sv_f = _run(layers, True, monkeypatch)
    monkeypatch.delenv("TADMM_FILTER_TM")
    zn, _, sv_n = _run(layers, False, monkeypatch)
    assert st_f["eligible"] == 3 and st_f["solves"] == 3 and st_f["fallbacks"] == 0, st_f
    for a, b in zip(zf, zn):
        assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
    for a, b in zip(sv_f, sv_n):
        np.testing.assert_allclose(a, b, rtol=1e-9)
    for (shape, tts, ranks), w, z in zip(specs, ws, zf):
        ref = O.prune_conv_rank_tt(w, tts, list(ranks))
        assert np.linalg.norm(z - ref) <= 1e-5 * np.linalg.norm(ref)


def test_filtered_projection_is_deterministic(dev, monkeypatch):
    rng = np.random.default_rng(4)
    _, layers, _ = _layers(dev, rng)
    z1, _, _ = _run(layers, True, monkeypatch)
    z2, _, _ = _run(layers, True, monkeypatch)
    for a, b in zip(z1, z2):
        assert np.array_equal(a, b)


def test_low_rank_input_falls_back_to_the_full_solve(dev, monkeypatch):
    """An exactly rank-40 weight: the filter's block is numerically rank deficient (Cholesky pivot breakdown), the
    problem must be handed to the full Jacobi solve and the projection must return the input."""
    rng = np.random.default_rng(5)
    specs, layers, ws = _layers(dev, rng, lowrank=40)
    zf, st, _ = _run(layers[1:], True, monkeypatch, sv_steps=())
    assert st["eligible"] == 1 and st["fallbacks"] == 1, st
    assert np.linalg.norm(zf[0] - ws[1]) <= 2e-6 * np.linalg.norm(ws[1])


def test_guard_catches_an_eigenvector_the_block_never_contained(dev, monkeypatch):
    """The a-posteriori residual check only sees the Ritz pairs it is shown.  Plant a DOMINANT eigenvector the block can
    never acquire: column k of the unfolding is supported on one row where every other column is zero, so e_k is an
    exact eigenvector of the Gram matrix (G[k][j] = 0 exactly), and the test hook TADMM_FILTER_TEST_ZERO_COL starts the
    block with no component along e_k -- products with G keep that component exactly zero.  Without the guard the
    filtered solve passes its verification and returns a Z that misses the largest singular direction; either guard --
    the default moments guard (||PGP||_F^2 / tr PGP from quantities at hand, csrc/filter.hip: filt_moments_kernel) or the
    opt-in power steps on the deflated operator (filt_guard_kernel, TADMM_FILTER_GUARD=n) -- must send the problem to the
    full solve."""
    from tadmm import ops
    from tadmm._cabi import KIND_TT_CONV
    rng = np.random.default_rng(11)
    shape, tts, ranks, k = (1024, 256, 1, 1), [1024, 1, 256], [1, 75, 75, 1], 37
    a = (rng.standard_normal((1024, 256)) * np.sqrt(2.0 / 256)).astype(np.float32)
    a[0, :] = 0.0
    a[:, k] = 0.0
    a[0, k] = 25.0                                   # sigma_1 = 25, far above the bulk (~3): the dominant direction
    w = a.reshape(shape)
    ref = O.prune_conv_rank_tt(w, tts, list(ranks))
    assert abs(ref[0, k, 0, 0] - 25.0) < 1e-3

    def run(zero_col, guard):
        monkeypatch.setenv("TADMM_FILTER", "1")
        if guard is None:
            monkeypatch.delenv("TADMM_FILTER_GUARD", raising=False)
        else:
            monkeypatch.setenv("TADMM_FILTER_GUARD", guard)
        if zero_col is None:
            monkeypatch.delenv("TADMM_FILTER_TEST_ZERO_COL", raising=False)
        else:
            monkeypatch.setenv("TADMM_FILTER_TEST_ZERO_COL", str(zero_col))
        t = torch.from_numpy(w).to(dev)
        L = dict(kind=KIND_TT_CONV, W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), tt_shapes=tts, ranks=list(ranks))
        plan = ops.ProjectionPlan([L])
        plan.run(update_u=False, use_u=False)
        st = plan.filter_stats()
        plan.close()
        return L["Z"].cpu().numpy(), st

    for guard in (None, "4"):                        # healthy start block: no fallback, both guards silent
        z_ok, st_ok = run(None, guard)
        assert st_ok["eligible"] == 1 and st_ok["fallbacks"] == 0, (guard, st_ok)
        assert np.linalg.norm(z_ok - ref) <= 1e-5 * np.linalg.norm(ref)
    z_blind, st_blind = run(k, "0")                  # guards off: the planted failure is real and goes unnoticed
    assert st_blind["fallbacks"] == 0, st_blind
    assert abs(z_blind[0, k, 0, 0]) < 1.0 and np.linalg.norm(z_blind - ref) > 0.5 * np.linalg.norm(ref)
    for guard in (None, "4"):                        # default (moments) / moments + power steps: rejected, full solve, right answer
        z_g, st_g = run(k, guard)
        assert st_g["fallbacks"] == 1, (guard, st_g)
        assert np.linalg.norm(z_g - ref) <= 1e-5 * np.linalg.norm(ref)


def test_filtered_cores_match_full_solve(dev, monkeypatch):
    """want_cores (ten2tt / --decompose): individual Ritz vectors, not only their span, must agree with the full
    solve; compared through the gauge-invariant reconstruction and the singular values."""
    rng = np.random.default_rng(6)
    _, layers, _ = _layers(dev, rng)
    zf, st, sv_f = _run(layers, True, monkeypatch, want_cores=True)
    zn, _, sv_n = _run(layers, False, monkeypatch, want_cores=True)
    assert st["fallbacks"] == 0
    for a, b in zip(zf, zn):
        assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
    for a, b in zip(sv_f, sv_n):
        np.testing.assert_allclose(a, b, rtol=1e-9)


def test_dgemm3_three_plane_product_matches_fp64(dev):
    """csrc/dgemm3.hip: A G^T with every value rounded to fp32 and split exactly into three bf16 terms, six bf16 MFMA
    products per fp32 product: accuracy of an fp32 GEMM (a few 1e-7 of the largest output) at the filter's shapes,
    including N % 64 == 32 (surplus column tile) and N / 32 odd (zero-filled half chunk)."""
    from tadmm import ops
    g0 = torch.Generator().manual_seed(11)
    for M, N in ((256, 1152), (192, 480), (64, 288), (32, 32)):
        a = torch.randn(M, N, generator=g0, dtype=torch.float64).to(dev)
        w = torch.randn(N, 2 * N, generator=g0, dtype=torch.float64).to(dev)
        g = (w @ w.t()).contiguous()
        ref = a @ g.t()
        got = ops.dgemm3(a, g)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        assert err < 5e-6, (M, N, err)
        f32 = (a.float() @ g.float().t()).double()
        assert err < 4 * ((f32 - ref).abs().max() / ref.abs().max()).item() + 1e-7      # no worse than an fp32 GEMM


def test_fp32_accuracy_filter_stages_are_opt_in_and_verified(dev, monkeypatch):
    """TADMM_FILTER_FAST=1: the early filter stages take the three-plane bf16 products (all stages but the last one the
    level needed in the previous run); every problem still passes the fp64 verification (no fallbacks) and Z agrees with
    the all-fp64 run far inside the parity bar."""
    from tadmm import ops, workloads
    from tadmm._cabi import KIND_TT_CONV
    model, hp, _ = workloads.build("resnet50_tt", seed=5)
    names = [n for n, _ in model.named_parameters() if n.startswith("layer4.") or n.startswith("layer3.0")]

    def run(nrun):
        ls = []
        for n, p in model.named_parameters():
            if n not in names:
                continue
            w = p.detach().to(dev).contiguous()
            ls.append(dict(kind=KIND_TT_CONV, W=w, U=torch.zeros_like(w), Z=torch.zeros_like(w),
                           tt_shapes=list(hp.tt_shapes[n]), ranks=list(hp.ranks[n])))
        pl = ops.ProjectionPlan(ls)
        stats = None
        for _ in range(nrun):
            pl.run(update_u=False)
            stats = pl.filter_stats()
        torch.cuda.synchronize()
        pl.enable_timing(True)
        pl.run(update_u=False)
        fast = pl.filter_timing_fast()
        z = [L["Z"].clone() for L in ls]
        pl.close()
        return z, stats, fast

    monkeypatch.delenv("TADMM_FILTER_FAST", raising=False)
    monkeypatch.delenv("TADMM_FILTER_FAST_STAGES", raising=False)
    z_ref, st_ref, fast_ref = run(2)
    assert fast_ref["launches"] == 0 and st_ref["fallbacks"] == 0          # default: every product in fp64
    monkeypatch.setenv("TADMM_FILTER_FAST", "1")
    z_fast, st_fast, fast = run(3)
    assert fast["launches"] > 0 and fast["flops"] > 0
    assert st_fast["fallbacks"] == 0 and st_fast["solves"] == st_ref["solves"]
    for a, b in zip(z_fast, z_ref):
        rel = ((a.double() - b.double()).norm() / b.double().norm()).item()
        assert rel < 1e-6, rel


@pytest.mark.parametrize("N", [16, 128, 480, 1152])
def test_eigh_of_blocks_of_identical_columns(dev, N):
    """Eight or more (nearly) identical columns in one Jacobi block: after the first annihilations the block works at its
    rounding floor (entries ~1e-32 of the largest), where the fp32 narrowing of the rotation angle used to underflow to
    0 * inf = NaN -- a constant weight tensor was enough to lose the dominant eigenvalue silently (N >= 192) or to end in
    TADMM_ERR_NOCONVERGE (small N).  The operands of the angle are now brought to a common exponent first."""
    from tadmm import ops
    g0 = torch.Generator().manual_seed(N)
    cases = {"ones": torch.ones(N, N, dtype=torch.float64)}
    blk = torch.zeros(N, N, dtype=torch.float64); blk[:8, :8] = 1.0
    cases["8x8 block"] = blk
    st = torch.zeros(N, N, dtype=torch.float64); st[4:12, 4:12] = 1.0; st[4, 4] += 1e-13
    cases["straddling, jittered"] = st
    v = torch.randn(N, 3, generator=g0, dtype=torch.float64)
    cases["rank 3"] = v @ v.t()
    for name, G in cases.items():
        lam, V, sweeps = ops.eigh(G.to(dev))
        ref = torch.linalg.eigvalsh(G).flip(0)
        assert torch.isfinite(lam).all() and torch.isfinite(V).all(), name
        assert (lam.cpu() - ref).abs().max().item() <= 1e-12 * ref.abs().max().item(), name
        assert sweeps <= 12, (name, sweeps)


def test_constant_weight_tensor_projects_like_the_oracle(dev):
    """A layer whose weights are all equal (rank-1 unfoldings, identical Gram columns) beside ordinary layers."""
    from tadmm import ops
    from tadmm._cabi import KIND_TT_CONV
    g0 = torch.Generator().manual_seed(2)
    shapes = [4, 4, 9, 4, 4]
    ranks = [1, 4, 12, 12, 4, 1]
    ws = [torch.full((16, 16, 3, 3), 0.37), torch.randn(16, 16, 3, 3, generator=g0) * 0.1,
          torch.full((16, 16, 3, 3), -2.5e-3), torch.randn(16, 16, 3, 3, generator=g0) * 0.1]
    ls = [dict(kind=KIND_TT_CONV, W=w.to(dev).contiguous(), U=torch.zeros_like(w).to(dev), Z=torch.zeros_like(w).to(dev),
               tt_shapes=list(shapes), ranks=list(ranks)) for w in ws]
    pl = ops.ProjectionPlan(ls)
    pl.run(update_u=False)
    torch.cuda.synchronize()
    for w, L in zip(ws, ls):
        z = O.prune_conv_rank_tt(w.numpy(), list(shapes), list(ranks))
        got = L["Z"].cpu().numpy()
        assert np.isfinite(got).all()
        assert np.linalg.norm(got - z) <= 1e-5 * np.linalg.norm(z)
    pl.close()
