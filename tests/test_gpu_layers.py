"""GPU parity of the factorised layers (TT*: golden vectors from the real reference; TK*: oracle +
invariants, parity unpinned) and of the Tucker projection."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class HP:
    pass


def _g4(golden_dir):
    return (np.load(os.path.join(golden_dir, "g4_layers.npz")),
            json.load(open(os.path.join(golden_dir, "g4_layers.json"))))


def _hp(m):
    hp = HP()
    hp.tt_shapes = {m["name"]: list(m["tt_shapes"])}
    hp.ranks = {m["name"]: list(m["ranks"])}
    return hp


def _make(m, dense_w=None, dense_b=None):
    from tadmm import tt_layers
    cls = getattr(tt_layers, m["cls"])
    if m["cls"].startswith("TTConv"):
        return cls(m["i"], m["o"], m["k"], stride=m["stride"], padding=m["padding"], bias=m["bias"], hp_dict=_hp(m),
                   name=m["name"], dense_w=dense_w, dense_b=dense_b)
    return cls(m["i"], m["o"], bias=m["bias"], hp_dict=_hp(m), name=m["name"], dense_w=dense_w, dense_b=dense_b)


def test_tt_layers_load_reference_state_dict_and_match_forward(golden_dir, dev):
    """A reference `state_dict` loads key-for-key and the forward output matches the reference's."""
    data, meta = _g4(golden_dir)
    for key, m in meta.items():
        layer = _make(m)
        sd = {k: torch.from_numpy(data[f"{key}_sd__{k}"]) for k in m["state_keys"]}
        assert set(layer.state_dict().keys()) == set(sd.keys()), key
        layer.load_state_dict(sd)
        layer = layer.to(dev)
        x = torch.from_numpy(data[key + "_x"]).to(dev)
        y = layer(x)
        ref = data[key + "_y"]
        assert tuple(y.shape) == ref.shape
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=3e-5 * max(1.0, np.abs(ref).max()),
                                   err_msg=key)


def test_tt_layers_decompose_dense_weight_on_device(golden_dir, dev):
    """--decompose hand-off: dense_w -> cores on the device; forward equals the reference layer's forward."""
    data, meta = _g4(golden_dir)
    for key, m in meta.items():
        w = torch.from_numpy(data[key + "_w"]).to(dev)
        b = torch.from_numpy(data[key + "_b"]).to(dev) if m["bias"] else None
        layer = _make(m, dense_w=w, dense_b=b).to(dev)
        for k in m["state_keys"]:
            assert tuple(layer.state_dict()[k].shape) == data[f"{key}_sd__{k}"].shape, (key, k)
        x = torch.from_numpy(data[key + "_x"]).to(dev)
        y = layer(x).detach().cpu().numpy()
        ref = data[key + "_y"]
        np.testing.assert_allclose(y, ref, rtol=0, atol=5e-5 * max(1.0, np.abs(ref).max()), err_msg=key)
        if m["cls"] == "TTConv2dM":
            # structural identity: TT-M forward == dense conv with the ADMM-projected kernel
            zk = O.prune_conv_rank_tt(data[key + "_w"], list(m["tt_shapes"]), list(m["ranks"]))
            s, p = m["stride"], m["padding"]
            dense = F.conv2d(x, torch.from_numpy(zk).to(dev), None, s, p).cpu().numpy()
            np.testing.assert_allclose(y, dense, atol=5e-5 * max(1.0, np.abs(dense).max()))


def test_ttconv2dm_bias_is_per_channel(dev):
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"c": [4, 4, 9, 4, 4]}
    hp.ranks = {"c": [1, 4, 10, 10, 4, 1]}
    torch.manual_seed(0)
    layer = tt_layers.TTConv2dM(16, 16, 3, padding=1, bias=True, hp_dict=hp, name="c").to(dev)
    with torch.no_grad():
        layer.bias.copy_(torch.arange(16.0))
    x = torch.randn(2, 16, 5, 7, device=dev)
    y = layer(x)
    layer.bias.data.zero_()
    y0 = layer(x)
    np.testing.assert_allclose((y - y0).detach().cpu().numpy(),
                               np.broadcast_to(np.arange(16.0).reshape(1, 16, 1, 1), (2, 16, 5, 7)), atol=1e-5)


def _torch_ttlinear_m(x, cores, out_features, bias):
    """plain-torch statement of TTLinearM.forward (TTLinear.py:75-93) for gradient checks."""
    shapes = [c.shape[1] for c in cores]
    ranks = [c.shape[0] for c in cores] + [1]
    prod, q = 1, 0
    for i, n in enumerate(shapes):
        prod *= n
        if prod == out_features:
            q = i + 1
            break
    out = x
    for i in range(len(cores) - q - 1, -1, -1):
        k = shapes[i + q] * ranks[i + q + 1]
        out = cores[i + q].reshape(-1, k).mm(out.reshape(-1, k).t()).t()
    for i in range(q - 1, -1, -1):
        out = cores[i].reshape(-1, ranks[i + 1]).mm(out.reshape(-1, ranks[i + 1]).t())
        out = out.reshape(ranks[i], -1).t()
    out = out.reshape(out_features, -1).t().reshape(list(x.shape[:-1]) + [out_features])
    return out + bias if bias is not None else out


def test_ttlinear_backward_matches_torch(dev):
    from tadmm import tt_layers
    hp = HP()
    hp.tt_shapes = {"l": (6, 8, 4, 6)}
    hp.ranks = {"l": (1, 5, 16, 5, 1)}
    torch.manual_seed(1)
    layer = tt_layers.TTLinearM(24, 48, hp_dict=hp, name="l").to(dev)
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(3, 7, 24, device=dev, requires_grad=True)
    y = layer(x)
    g = torch.randn_like(y)
    y.backward(g)
    got = [p.grad.clone() for p in layer.parameters()] + [x.grad.clone()]
    cores = [c.detach().double().requires_grad_(True) for c in layer.tt_cores]
    bias = layer.bias.detach().double().requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    yr = _torch_ttlinear_m(xr, cores, 48, bias)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), atol=1e-5)
    yr.backward(g.double())
    ref = [bias.grad] + [c.grad for c in cores] + [xr.grad]
    names = [n for n, _ in layer.named_parameters()] + ["x"]
    assert names[0] == "bias"
    for n, a, b in zip(names, got, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5 * max(1.0, float(b.abs().max())),
                                   err_msg=n)


def test_tucker_projection_vs_oracle_unpinned(dev):
    """Tucker branch (parity UNPINNED: tensorly absent): device HOSVD+HOOI against the float64 restatement
    and the invariants of SURVEY 8c."""
    from tadmm import tucker
    rng = np.random.default_rng(5)
    for shape, ranks in [((16, 12, 3, 3), [6, 5]), ((64, 64, 3, 3), [40, 40]), ((32, 16, 3, 3), [24, 12]), ((20, 14), [5, 4])]:
        w = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        core, (uo, ui), errs = tucker.partial_tucker(torch.from_numpy(w).to(dev), ranks)
        z = tucker.tucker_to_tensor(core, [uo, ui]).cpu().numpy().reshape(shape)
        uo_, ui_ = uo.cpu().numpy(), ui.cpu().numpy()
        eye_o = np.eye(ranks[0])
        if len(shape) == 2 and ranks[0] > ranks[1]:
            eye_o[ranks[1]:, ranks[1]:] = 0      # only r_in left vectors exist; the padding columns are zero
        np.testing.assert_allclose(uo_.T @ uo_, eye_o, atol=1e-5)
        np.testing.assert_allclose(ui_.T @ ui_, np.eye(ranks[1]), atol=1e-5)
        # (monotonicity of the HOOI error history: test_gpu_round2.py::test_hooi_error_history_is_non_increasing,
        #  on the path that returns the real per-sweep history)
        # core identity: core = W x0 Uo^T x1 Ui^T
        c_ref = np.einsum("oi...,or,is->rs...", w.astype(np.float64), uo_.astype(np.float64), ui_.astype(np.float64))
        np.testing.assert_allclose(core.cpu().numpy(), c_ref, atol=1e-5)
        # against the oracle: same stopping rule -> same sweep count and the same Z
        c64, f64, e64 = O.partial_tucker(w.astype(np.float64), ranks, return_trace=True)
        z64 = O.tucker_to_tensor(c64, f64)
        assert len(errs) == len(e64)
        rel = np.linalg.norm(z - z64) / np.linalg.norm(z64)
        assert rel <= 1e-5, (shape, rel)
        # idempotence
        z2, _ = tucker.project(torch.from_numpy(z).to(dev), ranks)
        assert float(np.linalg.norm(z2.cpu().numpy() - z) / np.linalg.norm(z)) <= 1e-5


def test_tk_layers_forward_equals_dense_conv_with_recovered_weight(dev):
    from tadmm import tk_layers
    hp = HP()
    hp.ranks = {"k": [10, 7], "l": [6, 5]}
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(16, 12, 3, 3, generator=g) * 0.2).to(dev)
    b = torch.randn(16, generator=g).to(dev)
    x = torch.randn(2, 12, 9, 9, generator=g).to(dev)
    zk = torch.from_numpy(O.prune_rank_tk(w.cpu().numpy(), [10, 7])).to(dev)
    ref = F.conv2d(x, zk, b, 2, 1)
    outs = []
    for cls in (tk_layers.TKConv2dC, tk_layers.TKConv2dM, tk_layers.TKConv2dR):
        layer = cls(12, 16, 3, stride=2, padding=1, bias=True, hp_dict=hp, name="k", dense_w=w, dense_b=b.clone()).to(dev)
        y = layer(x)
        outs.append(y)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref.cpu().numpy(), atol=2e-4)
        y.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
    c = tk_layers.TKConv2dC(12, 16, 3, stride=2, padding=1, hp_dict=hp, name="k", dense_w=w).to(dev)
    out, feats = c.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(2, 7, 9, 9), (2, 10, 5, 5), (2, 16, 5, 5)]
    # linear variants
    wl = (torch.randn(20, 14, generator=g) * 0.3).to(dev)
    xl = torch.randn(4, 3, 14, generator=g).to(dev)
    zl = torch.from_numpy(O.prune_rank_tk(wl.cpu().numpy(), [6, 5])).to(dev)
    for cls in (tk_layers.TKLinearM, tk_layers.TKLinearR):
        layer = cls(14, 20, bias=False, hp_dict=hp, name="l", dense_w=wl).to(dev)
        np.testing.assert_allclose(layer(xl).detach().cpu().numpy(), (xl @ zl.t()).cpu().numpy(), atol=2e-4)


def test_admm_tk_format_on_resnet32_table(dev):
    """Config 2 plumbing: ResNet-32 Tucker table, one ADMM iteration on a few layers vs the oracle."""
    from tadmm import workloads
    from tadmm.admm import ADMM
    model, hp, fmt = workloads.build("resnet32_tk", seed=0)
    keep = ["layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer3.4.conv2.weight"]
    hp.ranks = {k: hp.ranks[k] for k in keep}
    w = {k: p.detach().numpy().copy() for k, p in model.named_parameters() if k in keep}
    model = model.to(dev)
    a = ADMM(model, 1e-3, hp, fmt, dev, log=True)
    a.update(update_u=False)
    a.update()
    for k in keep:
        z = O.prune_rank_tk(w[k], hp.ranks[k])
        got = a.z[k].cpu().numpy()
        assert np.linalg.norm(got - z) / np.linalg.norm(z) <= 2e-5, k
        np.testing.assert_allclose(a.u[k].cpu().numpy(), w[k] - got, atol=1e-6)
        # (full-rank entries of the table project to themselves: their residual is rounding noise)
        assert abs(a.logger[k][0] - np.linalg.norm(w[k] - z)) <= 1e-4 * a.logger[k][0] + 1e-6 * np.linalg.norm(w[k])


def test_tucker_batched_plan_full_resnet32_table(dev):
    """Config 2: the whole ResNet-32 Tucker table (30 layers) in ONE grouped device plan (tadmm_tucker_*), two ADMM
    iterations against the oracle's per-layer restatement, and the batched plan against the step-by-step
    primitive-driven path.  Parity UNPINNED (tensorly absent): the checks are oracle agreement + invariants."""
    from tadmm import tucker, workloads
    from tadmm.admm import ADMM
    model, hp, fmt = workloads.build("resnet32_tk", seed=0)
    names = [k for k, _ in model.named_parameters() if k in hp.ranks]
    assert len(names) == 30
    w = {k: p.detach().numpy().copy() for k, p in model.named_parameters() if k in hp.ranks}
    model = model.to(dev)
    a = ADMM(model, 1e-3, hp, fmt, dev, log=True)
    a.update(update_u=False)
    its, errs = a._tk.plan.iterations()
    assert all(3 <= i <= 100 for i in its) and all(0.0 <= e < 1.0 for e in errs)
    worst = 0.0
    for k in names:
        z = O.prune_rank_tk(w[k], hp.ranks[k])
        got = a.z[k].cpu().numpy()
        rel = np.linalg.norm(got - z) / np.linalg.norm(z)
        worst = max(worst, rel)
        assert rel <= 5e-5, (k, rel)
    a.update()
    u = {k: np.zeros_like(v) for k, v in w.items()}
    for k in names:     # second iteration: Z = proj(W + U) with U = 0 still, then U = W - Z
        got = a.z[k].cpu().numpy()
        np.testing.assert_allclose(a.u[k].cpu().numpy(), w[k] - got, atol=1e-6)
        assert abs(a.logger[k][0] - np.linalg.norm(w[k] - got)) <= 1e-4 * a.logger[k][0] + 1e-6 * np.linalg.norm(w[k])
    # batched plan == primitive-driven path (same algorithm, independent orchestration)
    for k in names[::7]:
        wt = torch.from_numpy(w[k]).to(dev)
        c1, (uo1, ui1), e1 = tucker.partial_tucker(wt, hp.ranks[k])
        c2, (uo2, ui2), e2 = tucker.partial_tucker_eager(wt, hp.ranks[k])
        assert len(e1) == len(e2), k
        z1 = tucker.tucker_to_tensor(c1, [uo1, ui1])
        z2 = tucker.tucker_to_tensor(c2, [uo2, ui2])
        assert float(torch.linalg.vector_norm(z1 - z2) / torch.linalg.vector_norm(z2)) <= 1e-5, k
        assert abs(e1[-1] - e2[-1]) <= 1e-6


def test_tucker_plan_odd_shapes_vs_oracle(dev):
    """Shapes where an unfolding has fewer singular values than the requested rank, wide / tall linear layers and
    1x1 convs, all in ONE plan, against the oracle restatement (parity UNPINNED: tensorly absent)."""
    from tadmm import ops
    rng = np.random.default_rng(12)
    cases = [((12, 40), [5, 6]), ((40, 12), [6, 5]), ((64, 4, 1, 1), [8, 3]), ((8, 8, 3, 3), [8, 8]),
             ((24, 20, 3, 3), [20, 4]), ((16, 48, 1, 1), [10, 12])]
    ws = [(rng.standard_normal(s) * 0.3).astype(np.float32) for s, _ in cases]
    layers = []
    for w, (_, r) in zip(ws, cases):
        t = torch.from_numpy(w).to(dev)
        layers.append(dict(W=t, U=torch.zeros_like(t), Z=torch.empty_like(t), ranks=r))
    plan = ops.TuckerPlan(layers)
    resid = plan.run(update_u=True).cpu().numpy()
    for i, (w, (shape, r)) in enumerate(zip(ws, cases)):
        z = O.prune_rank_tk(w, r)
        got = layers[i]["Z"].cpu().numpy()
        rel = np.linalg.norm(got - z) / max(np.linalg.norm(z), 1e-30)
        assert rel <= 5e-5, (shape, r, rel)
        np.testing.assert_allclose(layers[i]["U"].cpu().numpy(), w - got, atol=1e-6)
        assert abs(resid[i] - np.linalg.norm((w - got).astype(np.float64)) ** 2) <= 1e-5 * max(resid[i], 1e-12) + 1e-10
        core, u_out, u_in = plan.factors(i)
        assert tuple(core.shape)[:2] == (r[0], r[1]) and tuple(u_out.shape) == (shape[0], r[0]) and \
            tuple(u_in.shape) == (shape[1], r[1])
    plan.close()


def test_decompose_state_dict_handoff(dev):
    """--decompose hand-off: dense *_model.pt state_dict -> factorised state_dict (reference keys), all table
    layers in one grouped device plan; loads into the layer classes and reproduces their forwards."""
    from tadmm import decompose, tt_layers

    class HPx:
        tt_shapes = {"l1.conv.weight": [4, 4, 9, 4, 4], "l2.conv.weight": [32, 1, 16], "head.weight": (6, 8, 4, 6)}
        ranks = {"l1.conv.weight": [1, 4, 10, 10, 4, 1], "l2.conv.weight": [1, 9, 9, 1], "head.weight": (1, 5, 16, 5, 1)}

    g = torch.Generator().manual_seed(9)
    dense = {"l1.conv.weight": torch.randn(16, 16, 3, 3, generator=g) * 0.2,
             "l1.bn.weight": torch.randn(16, generator=g), "l1.bn.running_mean": torch.randn(16, generator=g),
             "l2.conv.weight": torch.randn(32, 16, 1, 1, generator=g) * 0.2,
             "head.weight": torch.randn(48, 24, generator=g) * 0.3, "head.bias": torch.randn(48, generator=g)}
    x = torch.randn(2, 16, 7, 7, generator=g).to(dev)
    xl = torch.randn(5, 24, generator=g).to(dev)
    for variant, conv_cls, lin_cls in (("M", tt_layers.TTConv2dM, tt_layers.TTLinearM),
                                      ("R", tt_layers.TTConv2dR, tt_layers.TTLinearR)):
        sd = decompose.decompose_state_dict(dense, HPx, "tt", variant, device=dev)
        assert torch.equal(sd["l1.bn.weight"], dense["l1.bn.weight"]) and "head.bias" in sd      # pass-through
        assert "l1.conv.weight" not in sd and "head.weight" not in sd
        # the emitted keys are exactly what the layer classes register
        c1 = conv_cls(16, 16, 3, padding=1, bias=False, hp_dict=HPx, name="l1.conv.weight")
        c2 = conv_cls(16, 32, 1, bias=False, hp_dict=HPx, name="l2.conv.weight")
        ln = lin_cls(24, 48, bias=True, hp_dict=HPx, name="head.weight")
        c1.load_state_dict({k[len("l1.conv."):]: v for k, v in sd.items() if k.startswith("l1.conv.")})
        c2.load_state_dict({k[len("l2.conv."):]: v for k, v in sd.items() if k.startswith("l2.conv.")})
        ln.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")})
        # and they reproduce the constructor path (same device TT-SVD) -> identical forwards
        ref1 = conv_cls(16, 16, 3, padding=1, bias=False, hp_dict=HPx, name="l1.conv.weight",
                        dense_w=dense["l1.conv.weight"].to(dev)).to(dev)
        refl = lin_cls(24, 48, bias=True, hp_dict=HPx, name="head.weight", dense_w=dense["head.weight"].to(dev),
                       dense_b=dense["head.bias"].to(dev)).to(dev)
        np.testing.assert_allclose(c1.to(dev)(x).detach().cpu().numpy(), ref1(x).detach().cpu().numpy(), atol=1e-5)
        np.testing.assert_allclose(ln.to(dev)(xl).detach().cpu().numpy(), refl(xl).detach().cpu().numpy(), atol=1e-5)
        assert c2.to(dev)(x).shape == (2, 32, 7, 7)
    if True:   # Tucker keys
        class HPk:
            ranks = {"l1.conv.weight": [10, 7], "head.weight": [6, 5]}
        sd = decompose.decompose_state_dict({k: v for k, v in dense.items() if k != "l2.conv.weight"}, HPk, "tk", "C",
                                            device=dev)
        assert sd["l1.conv.first_kernel"].shape == (7, 16, 1, 1) and sd["l1.conv.core_kernel"].shape == (10, 7, 3, 3)
        assert sd["l1.conv.last_kernel"].shape == (16, 10, 1, 1) and sd["head.core_tensor"].shape == (6, 5)


def test_admm_state_dict_roundtrip(dev):
    """Checkpoint/resume of the ADMM state (Z, U, rho, clamped ranks) -- missing in the reference."""
    from tadmm.admm import ADMM

    class HPs:
        pass

    def make():
        hp = HPs()
        hp.tt_shapes = {"a": [4, 4, 9, 4, 4], "b": [32, 1, 16]}
        hp.ranks = {"a": [1, 4, 12, 12, 4, 1], "b": [1, 12, 20, 1]}
        g = torch.Generator().manual_seed(4)

        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.pa = torch.nn.Parameter(torch.randn(16, 16, 3, 3, generator=g) * 0.1)
                self.pb = torch.nn.Parameter(torch.randn(32, 16, 1, 1, generator=g) * 0.1)

            def named_parameters(self, *a, **k):
                return iter([("a", self.pa), ("b", self.pb)])
        return M().to(dev), hp

    m1, hp1 = make()
    a1 = ADMM(m1, 1e-3, hp1, "tt", dev, log=True)
    a1.update(update_u=False)
    a1.update()
    a1.adjust_rho(9, 10)
    state = a1.state_dict()
    a1.update()
    m2, hp2 = make()
    a2 = ADMM(m2, 1e-3, hp2, "tt", dev, log=True)
    a2.load_state_dict(state)
    assert a2.rho == 5e-3 and hp2.ranks["b"] == [1, 12, 12, 1]
    a2.update()
    for k in a1.z:
        assert torch.equal(a1.z[k], a2.z[k]) and torch.equal(a1.u[k], a2.u[k])
    assert a1.logger == a2.logger
