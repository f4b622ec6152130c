"""One ADMM iteration (admm.py:42-78) on EVERY shipped rank table whose architecture tadmm/workloads.py re-derives
(ResNet-18/50, ResNet-32/56, DeiT-small/tiny, ViT-small, VGG-16 / VGG-16-BN, DenseNet-40/121/201, MobileNetV2 CIFAR / ImageNet: 36 of the 37 tables):
synthetic N(0, 2/fan_in) weights, `update(update_u=False)` then `update()` (VGG: one `update()`).  Checks that hold for any projection:
finite Z, U = W - Z, logged residual = ||W - Z||, the projection does not increase the norm (||Z|| <= ||W|| (1 + 1e-5)),
and idempotence on one layer per table.  Tucker entries: parity UNPINNED, as everywhere."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _keys():
    from tadmm import hp, workloads
    return [k for k in hp.table_keys() if workloads.shape_fn_for(k) is not None]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("key", _keys())
def test_one_admm_iteration_on_table(dev, key):
    from tadmm import hp, workloads
    from tadmm.admm import ADMM
    fn = workloads.shape_fn_for(key)
    table = hp.fresh_table(key)
    fmt = key.split("_")[0]
    shapes = {name: fn(name) for name in table.ranks}
    model = workloads.SyntheticModel(shapes, seed=1).to(dev)
    a = ADMM(model, 1e-3, table, fmt, dev, log=True)
    if "_vgg16" not in key:          # (a VGG update takes seconds: 4096-wide classifier Grams, streamed Jacobi pairs)
        a.update(update_u=False)
    a.update()
    for name, p in model.named_parameters():
        w, z, u = p.data, a.z[name], a.u[name]
        assert torch.isfinite(z).all(), name
        assert float((u - (w - z)).abs().max()) <= 1e-6, name
        nw, nz = float(w.norm()), float(z.norm())
        assert nz <= nw * (1 + 1e-5), (name, nz, nw)
        assert abs(a.logger[name][0] - float((w - z).double().norm())) <= 1e-4 * a.logger[name][0] + 1e-6 * nw, name
    # idempotence of the projection on the first layer: proj(Z) = Z
    name0 = next(iter(shapes))
    z0 = a.z[name0].clone()
    sub = hp.fresh_table(key)
    sub.ranks = {name0: sub.ranks[name0]}
    if hasattr(sub, "tt_shapes"):
        sub.tt_shapes = {name0: sub.tt_shapes[name0]}
    m2 = workloads.SyntheticModel({name0: shapes[name0]}, seed=2).to(dev)
    next(iter(m2.flat)).data.copy_(z0)
    b = ADMM(m2, 1e-3, sub, fmt, dev)
    b.update(update_u=False)
    rel = float((b.z[name0] - z0).norm() / max(float(z0.norm()), 1e-30))
    assert rel <= 5e-5, (name0, rel)
