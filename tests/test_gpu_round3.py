"""Round-3 additions to the GPU suite.

* `append_admm_loss` inside `torch.autocast('cuda', dtype=torch.float16)` with a GradScaler-scaled loss -- the reference
  runs it there when `--fp16` (engines.py:285-289, parse_args.py:167): value and gradients against the golden G3;
* the inference caches of the factorised layers are dropped by everything that can change a weight without changing its
  version counter (`p.data = t`, eval()/train(), load_state_dict, explicit invalidate after `p.data.copy_`);
* a layer with frozen TT cores still back-propagates to its input, core kernel and bias (no silent autograd cut);
* TTLinearM / TKLinearM whose out_features is not 16-byte aligned (a 10-class head) train (backward included).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class HP:
    pass


def test_penalty_under_autocast_with_grad_scaler_matches_g3(golden_dir, dev):
    import _dist_gpu_worker as W
    from tadmm.admm import ADMM
    model, hp, data, meta, names = W.g2_model(dev)
    a = ADMM(model, meta["rho"], hp, "tt", dev, log=True)
    a.update(update_u=False)
    for it in range(3):
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(torch.from_numpy(data[f"w_it{it}__{k}"]))
        a.update()
    scale = 1024.0
    scaler = torch.amp.GradScaler("cuda", init_scale=scale)
    with torch.autocast("cuda", dtype=torch.float16):
        # a task loss produced under autocast (fp16 matmul -> fp32 reduction, as criterion() does), then the penalty
        x = torch.ones(4, 8, device=dev)
        task = (x @ torch.zeros(8, 8, device=dev)).float().sum()
        assert (x @ x.t()).dtype == torch.float16          # autocast really is active here
        total = a.append_admm_loss(task)
    assert total.dtype == torch.float32                     # fp32 parameters keep the penalty in fp32 (norm is an fp32 op)
    assert abs(float(total) - meta["penalty"]) <= 1e-5 * abs(meta["penalty"])
    scaler.scale(total).backward()
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy() / scale, data["pen_grad__" + k], rtol=1e-4, atol=1e-7, err_msg=k)
        assert torch.isfinite(p.grad).all()
    # an fp16 running loss stays fp16, like the reference's in-place `loss += ...` (admm.py:83)
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        run = torch.zeros((), device=dev, dtype=torch.float16)
        t16 = a.append_admm_loss(run)
    assert t16.dtype == torch.float16
    assert abs(float(t16) - meta["penalty"]) <= 2e-3 * abs(meta["penalty"]) + 1e-7      # fp16 rounding of the value
    t16.backward()
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), data["pen_grad__" + k], rtol=2e-3, atol=1e-7, err_msg=k)


def _tt_linear(dev, out_f=48, in_f=24):
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"fc": (6, out_f // 6, 4, in_f // 4)}
    hp.ranks = {"fc": (1, 5, 16, 5, 1)}
    torch.manual_seed(1)
    return tt_layers.TTLinearM(in_f, out_f, bias=True, hp_dict=hp, name="fc").to(dev)


def _tt_conv(dev):
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"c": [4, 4, 9, 4, 4]}
    hp.ranks = {"c": [1, 4, 10, 10, 4, 1]}
    torch.manual_seed(2)
    return tt_layers.TTConv2dM(16, 16, 3, stride=1, padding=1, bias=True, hp_dict=hp, name="c").to(dev)


def _tk_conv(dev):
    from tadmm import tk_layers
    hp = HP()
    hp.ranks = {"k": [10, 7]}
    torch.manual_seed(3)
    return tk_layers.TKConv2dC(12, 16, 3, stride=1, padding=1, bias=True, hp_dict=hp, name="k").to(dev)


@pytest.mark.parametrize("make,xshape,pname", [(_tt_linear, (5, 24), "tt_cores.1"), (_tt_conv, (2, 16, 7, 7), "in_tt_cores.0"),
                                               (_tk_conv, (2, 12, 8, 8), "first_kernel")])
def test_inference_caches_follow_weight_changes(dev, make, xshape, pname):
    layer = make(dev).eval()
    x = torch.randn(*xshape, device=dev)
    p = dict(layer.named_parameters())[pname]

    def fwd():
        with torch.no_grad():
            return layer(x).clone()

    def fresh_reference():
        ref = make(dev).eval()
        ref.load_state_dict(layer.state_dict())
        with torch.no_grad():
            return ref(x)

    y0 = fwd()
    assert torch.equal(y0, fwd())
    # 1. `p.data = t`: new storage, same version counter -> the data_ptr part of the cache key
    new = (p.data * 1.5 + 0.01).clone()
    p.data = new
    y1 = fwd()
    assert (y1 - y0).abs().max() > 1e-4
    np.testing.assert_allclose(y1.cpu().numpy(), fresh_reference().cpu().numpy(), atol=1e-5)
    # 2. in-place write through .data: invisible from the outside -> explicit invalidate
    p.data.copy_(new * 0.5)
    layer.invalidate_caches()
    y2 = fwd()
    assert (y2 - y1).abs().max() > 1e-4
    np.testing.assert_allclose(y2.cpu().numpy(), fresh_reference().cpu().numpy(), atol=1e-5)
    # 3. the usual flow (weights touched while training / loading, then eval()): mode switches drop the caches
    p.data.mul_(-1.0)
    layer.train()
    layer.eval()
    y3 = fwd()
    assert (y3 - y2).abs().max() > 1e-4
    np.testing.assert_allclose(y3.cpu().numpy(), fresh_reference().cpu().numpy(), atol=1e-5)
    # 4. load_state_dict copies in place (copy_ on the parameter bumps the counter, and the hook drops the caches anyway)
    sd = {k: v.clone() * 0.7 for k, v in layer.state_dict().items()}
    layer.load_state_dict(sd)
    y4 = fwd()
    np.testing.assert_allclose(y4.cpu().numpy(), fresh_reference().cpu().numpy(), atol=1e-5)
    assert (y4 - y3).abs().max() > 1e-4


def test_frozen_tt_cores_still_backpropagate_to_input_kernel_and_bias(dev):
    layer = _tt_conv(dev)
    ref = _tt_conv(dev)
    ref.load_state_dict(layer.state_dict())
    for p in list(layer.in_tt_cores) + list(layer.out_tt_cores):
        p.requires_grad_(False)                    # frozen cores; core_kernel and bias stay trainable
    x = torch.randn(2, 16, 7, 7, device=dev, requires_grad=True)      # 7x7: the one-launch path would apply in inference
    y = layer(x)
    assert y.requires_grad and y.grad_fn is not None
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().clone().requires_grad_(True)
    yr = ref(xr)                                   # all parameters trainable: the three differentiable launches
    yr.backward(g)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(layer.core_kernel.grad.cpu().numpy(), ref.core_kernel.grad.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), atol=2e-5)
    # fully frozen layer, input still wants a gradient (saliency, a frozen layer behind trainable ones)
    layer.core_kernel.requires_grad_(False)
    layer.bias.requires_grad_(False)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = layer(x2)
    assert y2.requires_grad
    y2.backward(g)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), xr.grad.cpu().numpy(), atol=2e-5)
    # and with nothing to differentiate the one-launch path is still taken and agrees
    with torch.no_grad():
        y3 = layer(x.detach())
    np.testing.assert_allclose(y3.cpu().numpy(), yr.detach().cpu().numpy(), atol=2e-5)


def test_ten_class_head_trains(dev):
    """out_features = 10 is not a multiple of 4: the fused backward cannot take the gradient rows (16-byte alignment), so
    the layers train through the per-core chain / three GEMMs instead of raising in backward."""
    from tadmm import tk_layers, tt_layers
    hp = HP()
    hp.tt_shapes = {"head": (2, 5, 4, 6)}
    hp.ranks = {"head": (1, 2, 8, 5, 1)}
    torch.manual_seed(4)
    tt = tt_layers.TTLinearM(24, 10, bias=True, hp_dict=hp, name="head").to(dev)
    hk = HP()
    hk.ranks = {"head": [6, 8]}
    tk = tk_layers.TKLinearM(24, 10, bias=True, hp_dict=hk, name="head").to(dev)
    x0 = torch.randn(7, 24, device=dev)
    for layer in (tt, tk):
        x = x0.clone().requires_grad_(True)
        y = layer(x)
        assert y.shape == (7, 10)
        g = torch.randn_like(y)
        y.backward(g)
        # fp64 reference of the same factorisation
        params = [p.detach().double() for p in layer.parameters()]
        if layer is tt:
            cores = [c.detach().double() for c in layer.tt_cores]
            w = cores[0].reshape(-1, cores[0].shape[-1])
            for c in cores[1:]:
                w = w.reshape(-1, c.shape[0]) @ c.reshape(c.shape[0], -1)
            w = w.reshape(10, 24)
        else:
            w = layer.last_factor.detach().double() @ layer.core_tensor.detach().double() @ layer.first_factor.detach().double()
        yr = x0.double() @ w.t() + layer.bias.detach().double()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.cpu().numpy(), atol=2e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), (g.double() @ w).cpu().numpy(), atol=2e-5)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
        with torch.no_grad():                      # inference keeps the one-launch path (no gradient rows involved)
            np.testing.assert_allclose(layer(x0).cpu().numpy(), yr.cpu().numpy(), atol=2e-5)


def test_complement_route_is_opt_in_and_matches_the_oracle(dev, monkeypatch):
    """The complement route (csrc/complement.hip: filter the DISCARDED subspace of G' = cI - G, keep its orthogonal
    complement) on a DeiT-small `proj` layer (288 x 384 unfolding, keep 256 of 288): opt-in, Z equal to the oracle and to
    the default (full solve) path."""
    from oracle import tt_oracle as O
    from tadmm import ops
    from tadmm._cabi import KIND_TT_LINEAR
    rng = np.random.default_rng(21)
    tts, ranks = (24, 16, 16, 24), (1, 18, 256, 18, 1)
    w = (rng.standard_normal((384, 384)) * np.sqrt(2.0 / 384)).astype(np.float32)
    ref = O.prune_linear_rank_tt(w, list(tts), list(ranks))

    def run(flag):
        if flag is None:
            monkeypatch.delenv("TADMM_COMPLEMENT", raising=False)
        else:
            monkeypatch.setenv("TADMM_COMPLEMENT", flag)
        t = torch.from_numpy(w).to(dev)
        L = dict(kind=KIND_TT_LINEAR, W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), tt_shapes=list(tts), ranks=list(ranks))
        plan = ops.ProjectionPlan([L])
        plan.run(update_u=False, use_u=False)
        st = plan.filter_stats()
        plan.close()
        return L["Z"].cpu().numpy(), st

    z0, st0 = run(None)
    assert st0["eligible"] == 0                       # default: the full Jacobi solve
    z1, st1 = run("1")
    assert st1["eligible"] == 1, st1                  # opt-in: the N = 288 step takes the complement route
    for z in (z0, z1):
        assert np.linalg.norm(z - ref) <= 1e-5 * np.linalg.norm(ref)
    assert np.linalg.norm(z1 - z0) <= 2e-6 * np.linalg.norm(z0)


@pytest.mark.parametrize("shape,r", [((256, 1024), 80), ((384, 1024), 120)])
def test_direct_rayleigh_ritz_route_is_opt_in_and_matches_the_oracle(dev, monkeypatch, shape, r):
    """The one-launch direct solve of the 128- / 192-column Rayleigh-Ritz problems (csrc/tridiag_mid.hip, TADMM_MID_DIRECT=1;
    off by default: slower than the tournament as measured in round 3): same Z as the oracle and as the default path, and
    no fallback of the filtered solve either way."""
    from oracle import tt_oracle as O
    from tadmm import ops
    from tadmm._cabi import KIND_SVD
    rng = np.random.default_rng(31)
    # a decaying spectrum: separated Ritz values, the case the route is meant for
    a = rng.standard_normal(shape)
    u, s, vt = np.linalg.svd(a, full_matrices=False)
    w = ((u * (s * np.exp(-np.arange(len(s)) / 40.0))) @ vt).astype(np.float32)
    ref = O.prune_linear_rank_svd(w, r)

    def run(flag):
        if flag is None:
            monkeypatch.delenv("TADMM_MID_DIRECT", raising=False)
        else:
            monkeypatch.setenv("TADMM_MID_DIRECT", flag)
        t = torch.from_numpy(w).to(dev)
        L = dict(kind=KIND_SVD, W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), ranks=r)
        plan = ops.ProjectionPlan([L])
        plan.run(update_u=False, use_u=False)
        st = plan.filter_stats()
        plan.close()
        return L["Z"].cpu().numpy(), st

    z0, st0 = run(None)
    z1, st1 = run("1")
    assert st0["eligible"] == 1 and st1["eligible"] == 1, (st0, st1)
    assert st0["fallbacks"] == 0 and st1["fallbacks"] == 0, (st0, st1)
    for z in (z0, z1):
        assert np.linalg.norm(z - ref) <= 1e-5 * np.linalg.norm(ref)
    assert np.linalg.norm(z1 - z0) <= 2e-6 * np.linalg.norm(z0)


def test_ttlinear_long_k_bf16_inference_uses_the_recovered_weight(dev):
    """DeiT-small fc2 (1536 -> 384, ranks (1,18,256,30,1)) in bf16 inference: the dense product with the recovered weight
    (the dispatch rule of TTLinearM._dense_pays) against the fp32 chain of the same cores; qkv (K = 384) keeps the chain."""
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"fc2": (24, 16, 32, 48), "qkv": (36, 32, 16, 24)}
    hp.ranks = {"fc2": (1, 18, 256, 30, 1), "qkv": (1, 25, 256, 18, 1)}
    torch.manual_seed(7)
    fc2 = tt_layers.TTLinearM(1536, 384, bias=True, hp_dict=hp, name="fc2").to(dev).eval()
    qkv = tt_layers.TTLinearM(384, 1152, bias=True, hp_dict=hp, name="qkv").to(dev).eval()
    x2 = torch.randn(197, 1536, device=dev)
    assert fc2._dense_pays(x2.bfloat16(), 256) and not fc2._dense_pays(x2, 256)
    assert not qkv._dense_pays(torch.zeros(1, 384, device=dev, dtype=torch.bfloat16), 256)
    with torch.no_grad():
        ref = fc2(x2)                                    # fp32: the chain kernel (three bf16 planes)
        got = fc2(x2.bfloat16()).float()
    assert fc2.__dict__["_chain_cache"]["dense"] is not None
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-2 * scale          # bf16 activations and weights: ~3 significant digits
    # weight changes are seen (cache keyed like the chain's)
    with torch.no_grad():
        fc2.tt_cores[0].mul_(2.0)
        got2 = fc2(x2.bfloat16()).float()
        ref2 = fc2(x2)
    assert float((got2 - ref2).abs().max()) <= 3e-2 * float(ref2.abs().max())
    assert float((got2 - got).abs().max()) > 0.1 * scale
