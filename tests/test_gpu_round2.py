"""Round-2 additions to the GPU parity suite: the gaps VERDICT.md (round 1) listed.

* ADMM constructed with the reference's default `torch.device('cuda')` (engines.py:69, :242);
* `forward_flops` against the printed ratios of numeric_example2.py (fixture G6);
* backward of TTConv2dM / TKConv2dC / TKConv2dM against fp64 autograd of the reference's forward formula;
* the reference behaviour TTConv2dM's per-channel bias REPLACES (TTConv.py:150-151 broadcasts along W);
* two penalties alive in one step keep their own gradients;
* the HOOI error history is non-increasing (eager path returns the real per-sweep history);
* `ADMM.update` on every distinct ResNet-50 layer shape (12) against the oracle;
* bf16 TTLinearM chain at a DeiT-S shape (BASELINE config 5).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu

REL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class HP:
    pass


def _small_model(dev):
    hp = HP()
    hp.tt_shapes = {"conv.weight": [4, 4, 9, 4, 4], "fc.weight": (6, 8, 4, 6)}
    hp.ranks = {"conv.weight": [1, 4, 12, 12, 4, 1], "fc.weight": (1, 5, 20, 5, 1)}

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(0)
            self.conv = torch.nn.Conv2d(16, 16, 3, bias=False)
            self.fc = torch.nn.Linear(24, 48, bias=False)
            with torch.no_grad():
                self.conv.weight.copy_(torch.randn(16, 16, 3, 3, generator=g) * 0.1)
                self.fc.weight.copy_(torch.randn(48, 24, generator=g) * 0.2)

    return M().to(dev), hp


@pytest.mark.parametrize("spec", ["cuda", "torch.device('cuda')", "cuda:0"])
def test_admm_accepts_unindexed_cuda_device(dev, spec):
    """The reference calls ADMM(..., torch.device(args.device)) with the default '--device cuda'."""
    from tadmm.admm import ADMM
    model, hp = _small_model(dev)
    d = torch.device("cuda") if spec.startswith("torch") else spec
    a = ADMM(model, 1e-3, hp, "tt", d, log=True)
    assert a.device == torch.device("cuda", torch.cuda.current_device())
    a.update(update_u=False)
    a.update()
    w = {k: p.detach().cpu().numpy() for k, p in model.named_parameters()}
    z = O.prune_conv_rank_tt(w["conv.weight"], [4, 4, 9, 4, 4], [1, 4, 12, 12, 4, 1])
    np.testing.assert_allclose(a.z["conv.weight"].cpu().numpy(), z, atol=REL * np.abs(w["conv.weight"]).max())
    loss = a.append_admm_loss(torch.zeros((), device=dev))
    loss.backward()
    assert model.conv.weight.grad is not None


def test_two_penalties_in_one_step_keep_their_gradients(dev):
    """A second append_admm_loss forward (other rho) before the first backward must not clobber the first's
    gradient rho*(W-Z+U) (admm.py:80-85)."""
    from tadmm.admm import ADMM
    model, hp = _small_model(dev)
    a = ADMM(model, 1e-3, hp, "tt", dev)
    a.update(update_u=False)
    a.update()
    l1 = a.append_admm_loss(torch.zeros((), device=dev))
    a.rho = 7e-3
    l2 = a.append_admm_loss(torch.zeros((), device=dev))
    l1.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    l2.backward()
    for k, p in model.named_parameters():
        ref = (p.data - a.z[k] + a.u[k])
        np.testing.assert_allclose(g1[k].cpu().numpy(), (1e-3 * ref).cpu().numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(p.grad.cpu().numpy(), (7e-3 * ref).cpu().numpy(), rtol=1e-5, atol=1e-9)
    assert abs(float(l2) / float(l1) - 7.0) < 1e-4


def test_forward_flops_matches_numeric_example2(golden_dir, dev):
    """G6: numeric_example2.py prints params/tt_params and flops/tt_flops of a TT conv (16->32, 3x3, 38x38,
    out modes [8,4] ranks [1,8,16], in modes [4,4] ranks [10,4,1]); TTConv2dM.forward_flops (TTConv.py:155-195)
    on the same configuration must reproduce both ratios and the output size."""
    from tadmm import tt_layers
    g6 = json.load(open(os.path.join(golden_dir, "g6_numeric_examples.json")))["numeric_example2.py"]
    want_cr = float(g6[0].split(":")[1])
    want_su = float(g6[1].split(":")[1])
    want_numel = int(g6[2])
    hp = HP()
    hp.tt_shapes = {"c": [8, 4, 9, 4, 4]}
    hp.ranks = {"c": [1, 8, 16, 10, 4, 1]}
    layer = tt_layers.TTConv2dM(16, 32, 3, stride=1, padding=1, bias=False, hp_dict=hp, name="c").to(dev)
    x = torch.randn(1, 16, 38, 38, device=dev)
    out, base_flops, tt_flops = layer.forward_flops(x)
    assert out[0].numel() == want_numel
    assert abs(base_flops / tt_flops - want_su) <= 1e-9 * want_su
    tt_params = sum(p.numel() for p in layer.parameters())
    assert abs(16 * 32 * 9 / tt_params - want_cr) <= 1e-12 * want_cr
    # the TK layers' accounting (TKConv.py:111-134): stage maps x kernel sizes
    from tadmm import tk_layers
    hk = HP()
    hk.ranks = {"k": [10, 7]}
    tk = tk_layers.TKConv2dC(12, 16, 3, stride=2, padding=1, bias=False, hp_dict=hk, name="k").to(dev)
    xk = torch.randn(2, 12, 9, 9, device=dev)
    _, bf, cf = tk.forward_flops(xk)
    assert abs(bf - 5 * 5 * 9 * 12 * 16 / 1e6) < 1e-12
    assert abs(cf - (9 * 9 * 7 * 12 + 5 * 5 * 10 * 7 * 9 + 5 * 5 * 16 * 10) / 1e6) < 1e-12


def _ref_ttconv_m(x, in_cores, core_kernel, out_cores, stride, padding, out_channels):
    """fp64 statement of TTConv2dM.forward (TTConv.py:130-153), bias excluded."""
    b, _, h, w = x.shape
    out = x.permute(0, 2, 3, 1)
    for c in reversed(in_cores):
        k = c.shape[1] * c.shape[2]
        out = c.reshape(c.shape[0], k).mm(out.reshape(-1, k).t()).t()
    out = out.reshape(b, h, w, in_cores[0].shape[0]).permute(0, 3, 1, 2)
    out = F.conv2d(out, core_kernel, None, stride, padding)
    _, _, h2, w2 = out.shape
    out = out.permute(0, 2, 3, 1)
    for c in reversed(out_cores):
        out = c.reshape(-1, c.shape[2]).mm(out.reshape(-1, c.shape[2]).t())
        out = out.reshape(c.shape[0], -1).t()
    return out.reshape(out_channels, b, h2, w2).permute(1, 0, 2, 3)


def test_ttconv2dm_backward_matches_fp64_autograd(dev):
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"c": [4, 4, 9, 4, 4]}
    hp.ranks = {"c": [1, 4, 10, 10, 4, 1]}
    torch.manual_seed(3)
    layer = tt_layers.TTConv2dM(16, 16, 3, stride=2, padding=1, bias=True, hp_dict=hp, name="c").to(dev)
    x = torch.randn(2, 16, 9, 7, device=dev, requires_grad=True)
    y = layer(x)
    g = torch.randn_like(y)
    y.backward(g)
    ic = [c.detach().double().requires_grad_(True) for c in layer.in_tt_cores]
    oc = [c.detach().double().requires_grad_(True) for c in layer.out_tt_cores]
    ck = layer.core_kernel.detach().double().requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    yr = _ref_ttconv_m(xr, ic, ck, oc, 2, 1, 16) + layer.bias.detach().double().view(1, -1, 1, 1)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), atol=2e-5)
    yr.backward(g.double())
    pairs = [("x", x.grad, xr.grad), ("core_kernel", layer.core_kernel.grad, ck.grad),
             ("bias", layer.bias.grad, g.double().sum((0, 2, 3)))]
    pairs += [(f"in{i}", layer.in_tt_cores[i].grad, ic[i].grad) for i in range(len(ic))]
    pairs += [(f"out{i}", layer.out_tt_cores[i].grad, oc[i].grad) for i in range(len(oc))]
    for n, a, b in pairs:
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-5 * max(1.0, float(b.abs().max())),
                                   err_msg=n)


def test_reference_ttconv2dm_bias_broadcast_is_what_we_replace(dev):
    """The reference adds its (O,) bias to the (B,O,H',W') output (TTConv.py:150-151): PyTorch broadcasting aligns
    it with the LAST axis, so the reference raises unless W' == O and otherwise adds bias[w] along the width.
    This pins the replaced behaviour so the deliberate deviation (per-channel bias) stays visible."""
    out = torch.zeros(2, 16, 5, 7)
    bias = torch.arange(16.0)
    with pytest.raises(RuntimeError):
        out + bias                                   # W'=7 != O=16: the reference's forward fails
    sq = torch.zeros(2, 16, 5, 16) + bias            # W' == O: runs, but adds bias along the WIDTH axis
    assert torch.equal(sq[0, 3, 2], bias) and float(sq[0, 3, 2, 5]) == 5.0


def test_tk_conv_backward_matches_fp64_autograd(dev):
    from tadmm import tk_layers
    hp = HP()
    hp.ranks = {"k": [10, 7]}
    torch.manual_seed(5)
    x0 = torch.randn(2, 12, 9, 9, device=dev)
    for cls in (tk_layers.TKConv2dC, tk_layers.TKConv2dM):
        layer = cls(12, 16, 3, stride=2, padding=1, bias=True, hp_dict=hp, name="k").to(dev)
        with torch.no_grad():
            layer.bias.normal_()
        x = x0.clone().requires_grad_(True)
        y = layer(x)
        g = torch.randn_like(y)
        y.backward(g)
        params = {n: p for n, p in layer.named_parameters()}
        ref = {n: p.detach().double().requires_grad_(True) for n, p in params.items()}
        xr = x0.double().requires_grad_(True)
        if cls is tk_layers.TKConv2dC:               # TKConv.py:93-98
            first, last = ref["first_kernel"], ref["last_kernel"]
        else:                                        # TKConv.py:210-214: factors as linear maps on NHWC
            first = ref["first_factor"].reshape(7, 12, 1, 1)
            last = ref["last_factor"].reshape(16, 10, 1, 1)
        yr = F.conv2d(F.conv2d(F.conv2d(xr, first), ref["core_kernel"], None, 2, 1), last, ref["bias"])
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), atol=3e-5)
        yr.backward(g.double())
        np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), atol=3e-5 * float(xr.grad.abs().max()))
        for n, p in params.items():
            b = ref[n].grad
            np.testing.assert_allclose(p.grad.cpu().numpy(), b.reshape(p.shape).cpu().numpy(), rtol=0,
                                       atol=3e-5 * max(1.0, float(b.abs().max())), err_msg=f"{cls.__name__}.{n}")


def test_hooi_error_history_is_non_increasing(dev):
    """HOOI never increases the reconstruction error: checked on the real per-sweep history of the
    primitive-driven path, whose final value must agree with the batched plan's."""
    from tadmm import tucker
    rng = np.random.default_rng(5)
    for shape, ranks in [((16, 12, 3, 3), [6, 5]), ((64, 64, 3, 3), [40, 40]), ((20, 14), [5, 4])]:
        w = torch.from_numpy((rng.standard_normal(shape) * 0.1).astype(np.float32)).to(dev)
        _, _, hist = tucker.partial_tucker_eager(w, ranks)
        assert len(hist) >= 3 and len(set(hist)) > 1
        assert all(hist[i + 1] <= hist[i] + 1e-7 for i in range(len(hist) - 1)), hist
        _, _, errs = tucker.partial_tucker(w, ranks)
        assert len(errs) == len(hist) and abs(errs[-1] - hist[-1]) <= 1e-6


def test_hooi_warm_started_jacobi_equals_cold(dev, monkeypatch):
    """HOOI solves of a mode start from the previous sweep's eigenvectors (EigDesc::warm): same projection, same sweep
    counts and errors as with cold starts (TADMM_TUCKER_WARM=0), on layers of <= 64 channels and a run repeated on the
    same plan (the warm images must be invalidated between runs)."""
    from tadmm import ops
    rng = np.random.default_rng(11)
    shapes = [((64, 64, 3, 3), [25, 23]), ((32, 16, 3, 3), [12, 9]), ((48, 40), [10, 12]), ((16, 16, 3, 3), [16, 16])]

    def run(flag):
        monkeypatch.setenv("TADMM_TUCKER_WARM", flag)
        layers = []
        for i, (shape, ranks) in enumerate(shapes):
            w = torch.from_numpy((np.random.default_rng(100 + i).standard_normal(shape) * 0.1).astype(np.float32)).to(dev)
            layers.append(dict(W=w, U=torch.zeros_like(w), Z=torch.zeros_like(w), ranks=ranks))
        plan = ops.TuckerPlan(layers)
        out = []
        for _ in range(2):                       # second run: U has changed, the plan is the same
            plan.run(update_u=True)
            torch.cuda.synchronize()
            out.append(([L["Z"].clone() for L in layers], plan.iterations()))
        plan.close()
        return out

    cold, warm = run("0"), run("1")
    for (zc, (itc, errc)), (zw, (itw, errw)) in zip(cold, warm):
        assert itc == itw, (itc, itw)
        for a, b, ec, ew in zip(zc, zw, errc, errw):
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
            if max(ec, ew) > 1e-3:           # (a full-rank layer's error is sqrt of fp32 rounding noise: ~1e-4 either way)
                assert abs(ec - ew) <= 1e-6


def test_admm_update_every_distinct_resnet50_shape_vs_oracle(dev):
    """Headline table: Z, U and the logged residual of EVERY distinct layer shape (12) against the oracle."""
    from tadmm import workloads
    from tadmm.admm import ADMM
    model, hp, fmt = workloads.build("resnet50_tt", seed=0)
    ref_w = {k: p.detach().numpy().copy() for k, p in model.named_parameters()}
    model = model.to(dev)
    a = ADMM(model, 1e-3, hp, fmt, dev, log=True)
    a.update(update_u=False)
    a.update()
    _, hp2, _ = workloads.build("resnet50_tt", seed=0)
    seen = set()
    for name, w in ref_w.items():
        sig = (w.shape, tuple(hp2.tt_shapes[name]), tuple(hp2.ranks[name]))
        if sig in seen:
            continue
        seen.add(sig)
        z = O.project_layer(w, fmt, list(hp2.ranks[name]), hp2.tt_shapes[name])
        got = a.z[name].cpu().numpy()
        err = np.linalg.norm((got - z).astype(np.float64)) / np.linalg.norm(z.astype(np.float64))
        assert err <= REL, (name, err)
        np.testing.assert_allclose(a.u[name].cpu().numpy(), w - got, atol=1e-6)
        # (a full-rank layer projects onto itself: both residuals are fp32 rounding noise of order eps * ||w||, and the
        # device's got SMALLER with the compensated GEMM accumulation of round 3 -- hence the eps-scaled absolute term)
        noise = 4 * np.finfo(np.float32).eps * np.linalg.norm(w.astype(np.float64))
        assert abs(a.logger[name][0] - np.linalg.norm((w - z).astype(np.float64))) <= 1e-4 * a.logger[name][0] + noise
    assert len(seen) == 12


def test_ttlinear_bf16_chain_deit_small_shape(dev):
    """BASELINE config 5: TTLinearM of a DeiT-small qkv layer (384 -> 1152, tt_shapes (36,32,16,24), ranks
    (1,25,256,18,1)) on the bf16 matrix-core chain against the fp32 chain of the same cores."""
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"qkv": (36, 32, 16, 24)}
    hp.ranks = {"qkv": (1, 25, 256, 18, 1)}
    torch.manual_seed(7)
    layer = tt_layers.TTLinearM(384, 1152, bias=True, hp_dict=hp, name="qkv").to(dev)
    with torch.no_grad():
        layer.bias.normal_(std=0.1)
    x = torch.randn(4, 197, 384, device=dev)
    with torch.no_grad():
        y32 = layer(x)
        yb = layer(x.to(torch.bfloat16))
    assert yb.dtype == torch.bfloat16 and yb.shape == y32.shape
    # reference for the bf16 chain: fp32 chain on bf16-rounded cores and input; bf16 carries 8 bits
    err = float((yb.float() - y32).norm() / y32.norm())
    assert err <= 2e-2, err
    dense = tt_layers.TTLinearR(384, 1152, bias=True, hp_dict=hp, name="qkv").to(dev)
    dense.load_state_dict(layer.state_dict())
    with torch.no_grad():
        yd = dense(x)
    np.testing.assert_allclose(y32.cpu().numpy(), yd.cpu().numpy(), atol=2e-5 * float(yd.abs().max()))


def test_two_lane_plan_equals_one_lane_plan(dev, monkeypatch):
    """ResNet-50 table: the plan splits into two lanes (the 3x3 kernels of layer3/layer4 on the high-priority stream).
    Z, U and the residuals of the two-lane run must agree with those of the one-lane run (TADMM_LANES=1) to rounding
    level, and the caller's stream must see them."""
    from tadmm import ops, workloads
    from tadmm._cabi import KIND_TT_CONV
    model, hp, _ = workloads.build("resnet50_tt", seed=3)
    names = [n for n, _ in model.named_parameters()]

    def build():
        ls = []
        for n, p in model.named_parameters():
            w = p.detach().to(dev).contiguous()
            ls.append(dict(kind=KIND_TT_CONV, W=w, U=torch.full_like(w, 1e-3), Z=torch.zeros_like(w),
                           tt_shapes=list(hp.tt_shapes[n]), ranks=list(hp.ranks[n])))
        return ls

    two = build()
    p2 = ops.ProjectionPlan(two)
    lanes = p2.lanes()
    long_chain = [n for n, l in zip(names, lanes) if l == 0]
    assert max(lanes) == 1 and len(long_chain) == 9
    assert all(".conv2." in n and (n.startswith("layer3.") or n.startswith("layer4.")) for n in long_chain)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                      # a non-default caller stream: joined in front and behind
        r2 = p2.run(update_u=True).clone()
        z2 = [L["Z"].clone() for L in two]
    side.synchronize()
    sv = p2.singular_values(names.index("layer4.0.conv2.weight"), 1)
    p2.close()
    monkeypatch.setenv("TADMM_LANES", "1")
    one = build()
    p1 = ops.ProjectionPlan(one)
    assert max(p1.lanes()) == 0
    r1 = p1.run(update_u=True)
    torch.cuda.synchronize()
    sv1 = p1.singular_values(names.index("layer4.0.conv2.weight"), 1)
    # not bit-equal: a group sweeps until its slowest problem has converged, so a problem grouped differently may get
    # one more (or fewer) Jacobi sweep past its own convergence -- differences stay at rounding level
    worst = 0.0
    for a, b, n in zip(two, one, names):
        dz = (a["Z"].double() - b["Z"].double()).norm().item() / b["Z"].double().norm().item()
        du = (a["U"].double() - b["U"].double()).norm().item() / b["U"].double().norm().item()
        worst = max(worst, dz, du)
        assert dz <= 2e-6 and du <= 2e-6, (n, dz, du)
    print("two lanes vs one lane: worst relative difference %.2e" % worst)
    for z, a in zip(z2, two):
        assert torch.equal(z, a["Z"])                  # what the caller's stream saw is the final state
    np.testing.assert_allclose(r2.cpu().numpy(), r1.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(sv, sv1, rtol=1e-9)
    p1.close()


def test_two_lane_plans_can_be_created_and_destroyed_repeatedly(dev):
    """A two-lane plan owns two streams and a worker thread: creating, running and destroying many of them (one per
    ADMM object / epoch in a training script) must neither hang nor leak threads."""
    import threading
    from tadmm import ops, workloads
    from tadmm._cabi import KIND_TT_CONV
    model, hp, _ = workloads.build("resnet18_tt", seed=1)
    ls = []
    for n, p in model.named_parameters():
        w = p.detach().to(dev).contiguous()
        ls.append(dict(kind=KIND_TT_CONV, W=w, U=torch.zeros_like(w), Z=torch.zeros_like(w),
                       tt_shapes=list(hp.tt_shapes[n]), ranks=list(hp.ranks[n])))
    before = threading.active_count()
    ref = None
    for it in range(6):
        pl = ops.ProjectionPlan(ls)
        assert max(pl.lanes()) == 1
        r = pl.run(update_u=False).clone()
        torch.cuda.synchronize()
        if ref is None:
            ref = r
        else:
            assert torch.equal(r, ref)                      # same plan, same inputs: bitwise the same residuals
        if it % 2:
            pl.close()                                      # explicit and implicit (garbage-collected) destruction
        del pl
    assert threading.active_count() == before               # the workers are native threads and are joined in destroy
