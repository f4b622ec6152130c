"""GPU parity of the individual HIP kernels (through the C ABI) against numpy/torch fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _rand(shape, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dev)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (75, 1024, 256), (130, 33, 77), (1, 5, 3), (105, 4608, 480),
                                   (257, 129, 65)])
@pytest.mark.parametrize("ta,tb,tc", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 1)])
def test_gemm_strided(dev, M, N, K, ta, tb, tc):
    from tadmm import ops
    a = _rand((K, M) if ta else (M, K), 1, dev)
    b = _rand((N, K) if tb else (K, N), 2, dev)
    av = a.t() if ta else a
    bv = b.t() if tb else b
    out = torch.empty((N, M) if tc else (M, N), device=dev)
    ov = out.t() if tc else out
    ops.mm(av, bv, out=ov)
    ref = av.double() @ bv.double()
    err = (ov.double() - ref).abs().max().item()
    scale = (av.double().abs() @ bv.double().abs()).max().item()
    assert err <= 4e-7 * scale, (err, scale)   # fp32 fma chain, K <= 480


def test_gemm_bias_alpha(dev):
    from tadmm import ops
    a, b = _rand((70, 40), 3, dev), _rand((40, 90), 4, dev)
    bn, bm = _rand((90,), 5, dev), _rand((70,), 6, dev)
    out = ops.mm(a, b, alpha=0.5, bias_n=bn, bias_m=bm)
    ref = 0.5 * (a.double() @ b.double()) + bn.double()[None, :] + bm.double()[:, None]
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("m,n", [(8, 4608), (32, 73728), (120, 1152), (75, 1024), (130, 2048), (240, 2304),
                                 (576, 64), (945, 512), (1024, 256), (17, 33), (33, 17), (5, 5)])
def test_gram_fp64_exact_products(dev, m, n):
    from tadmm import ops
    a = _rand((m, n), 10 + m, dev)
    G = ops.gram(a).cpu().numpy()
    a64 = a.cpu().numpy().astype(np.float64)
    ref = a64 @ a64.T if m <= n else a64.T @ a64
    assert G.shape == ref.shape
    # products are exact, only the fp64 summation order differs
    np.testing.assert_allclose(G, ref, rtol=0, atol=1e-12 * np.abs(ref).max() * max(1, np.log2(max(m, n))))
    assert np.array_equal(G, G.T)   # bitwise symmetric by construction


@pytest.mark.parametrize("N", [5, 16, 30, 75, 130, 256, 300, 600, 1100])      # 600 / 1100: pair-kernel fallbacks
@pytest.mark.parametrize("kind", ["gauss", "decay"])
def test_eigh_jacobi(dev, N, kind):
    from tadmm import ops
    rng = np.random.default_rng(N)
    if kind == "gauss":
        a = rng.standard_normal((N, 3 * N))
    else:
        q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        a = q * np.exp(-6.0 * np.arange(N) / N)
    G = a @ a.T
    G = 0.5 * (G + G.T)
    ev, vec, sweeps = ops.eigh(torch.from_numpy(G).to(dev))
    ev, vec = ev.cpu().numpy(), vec.cpu().numpy()
    ref = np.linalg.eigvalsh(G)[::-1]
    np.testing.assert_allclose(ev, ref, rtol=0, atol=1e-12 * ref[0])
    assert np.all(np.diff(ev) <= 0)
    # residual and orthonormality of the leading half (the part the TT truncation uses)
    k = max(1, N // 2)
    V = vec[:k]
    np.testing.assert_allclose(V @ V.T, np.eye(k), atol=1e-10)
    R = G @ V.T - V.T * ev[:k]
    assert np.abs(R).max() <= 1e-10 * ref[0]
    assert sweeps <= 30, sweeps


@pytest.mark.gpu
def test_gram_direct_write_path_large_n(dev):
    """N >= 704 gives >= 256 tile pairs, so the standalone Gram takes the no-split-K path: the product kernel
    writes the tile, its mirror and the zero padding itself (no reduce pass)."""
    from tadmm import ops
    g = torch.Generator().manual_seed(11)
    a = torch.randn(720, 900, generator=g)
    G = ops.gram(a.to(dev)).cpu().numpy()
    ref = a.double().numpy() @ a.double().numpy().T
    assert G.shape == (720, 720)
    np.testing.assert_allclose(G, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    np.testing.assert_array_equal(G, G.T)


@pytest.mark.gpu
def test_tucker_plan_is_deterministic_and_batches_mixed_layers(dev):
    """One grouped Tucker plan over a conv, a 1x1 conv and two linear layers (one with more rows than columns,
    one with r_out > r_in so a factor carries a zero column): run twice -> bitwise identical Z, U, residuals."""
    from tadmm import ops
    g = torch.Generator().manual_seed(4)
    shapes = [(24, 16, 3, 3), (32, 48, 1, 1), (40, 12), (20, 14)]
    ranks = [[10, 8], [12, 20], [6, 5], [5, 4]]
    outs = []
    for _ in range(2):
        layers = []
        gg = torch.Generator().manual_seed(4)
        for s, r in zip(shapes, ranks):
            w = (torch.randn(*s, generator=gg) * 0.2).to(dev)
            layers.append(dict(W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=r))
        plan = ops.TuckerPlan(layers)
        r1 = plan.run(update_u=True).clone()
        r2 = plan.run(update_u=True).clone()
        its, errs = plan.iterations()
        assert all(i >= 3 for i in its) and all(np.isfinite(e) for e in errs)
        outs.append(([L["Z"].clone() for L in layers], [L["U"].clone() for L in layers], r1, r2))
        plan.close()
    for za, zb in zip(outs[0][0], outs[1][0]):
        assert torch.equal(za, zb)
    for ua, ub in zip(outs[0][1], outs[1][1]):
        assert torch.equal(ua, ub)
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 70, 25), (12608, 288, 24), (257, 129, 300), (1, 5, 3)])
def test_gemm_bf16_nt(dev, M, N, K):
    """bf16 A * Bt^T with fp32 accumulation against torch on the same bf16 inputs (result rounded to bf16 once)."""
    from tadmm import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    bt = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    got = ops.mm_nt_bf16(a, bt, bias).float().cpu()
    ref = (a.float().cpu().double() @ bt.float().cpu().double().T + bias.cpu().double())
    err = (got.double() - ref).abs()
    # one bf16 rounding of the result (2^-8 relative) + fp32 accumulation noise
    assert float((err / (ref.abs() + 1e-2 * ref.abs().max())).max()) <= 6e-3


@pytest.mark.gpu
def test_ttlinear_bf16_forward_matches_fp32(dev):
    from tadmm import tt_layers

    class HP:
        tt_shapes = {"w": (6, 8, 4, 6)}
        ranks = {"w": (1, 5, 16, 5, 1)}
    lin = tt_layers.TTLinearM(24, 48, bias=True, hp_dict=HP, name="w").to(dev)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():    # like the reference (TTLinear.py:55), TTLinearM leaves a fresh bias uninitialised
        lin.bias.copy_(torch.randn(48, generator=g))
    x = torch.randn(7, 33, 24, generator=g).to(dev)
    with torch.no_grad():
        ref = lin(x)
        got = lin(x.to(torch.bfloat16))
    assert got.dtype == torch.bfloat16 and got.shape == ref.shape
    rel = float((got.float() - ref).norm() / ref.norm())
    assert rel <= 2e-2, rel
