"""Generate golden vectors G1..G6 from the REAL reference.

Runs ONLY in the build container: imports /root/reference/{ttd,admm,TTConv,
TTLinear}.py and records inputs + outputs as small .npz / .json fixtures.  The
reference never travels to the GPU box; these data files do.

``admm.py`` imports the absent third-party ``tensorly`` at module level.  A
stand-in module object whose entry points raise is registered first so the
import succeeds; the TT and SVD branches recorded here never touch it.  The
Tucker branch is NOT recorded (parity unpinned, see oracle/tt_oracle.py).
"""
import json, os, sys, types, io, contextlib
import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def _absent(*a, **k):
    raise RuntimeError("tensorly is not installed in this image")


tl = types.ModuleType("tensorly")
tl.set_backend = lambda *_a, **_k: None
tl.tucker_to_tensor = _absent
dec = types.ModuleType("tensorly.decomposition")
dec.partial_tucker = _absent
dec.parafac = _absent
tl.decomposition = dec
sys.modules["tensorly"] = tl
sys.modules["tensorly.decomposition"] = dec

import ttd            # noqa: E402
import admm as ref_admm   # noqa: E402
import TTConv as ref_ttconv  # noqa: E402
import TTLinear as ref_ttlinear  # noqa: E402


def decaying(shape, rng, rate=6.0):
    """Random matrix with an exponentially decaying spectrum, reshaped."""
    m = shape[0]
    n = int(np.prod(shape[1:]))
    k = min(m, n)
    q1, _ = np.linalg.qr(rng.standard_normal((m, k)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, k)))
    s = np.exp(-rate * np.arange(k) / k)
    return ((q1 * s) @ q2.T).reshape(shape).astype(np.float32)


# ---------------------------------------------------------------- G1
def g1():
    rng = np.random.default_rng(20211001)
    cases = {
        # name: (tensor shape fed to ten2tt, tt_shapes, tt_ranks)
        "conv3": ([16, 9, 16], [16, 9, 16], [1, 8, 8, 1]),
        "conv5": ([16, 9, 16], [4, 4, 9, 4, 4], [1, 4, 12, 12, 4, 1]),
        "conv1x1": ([32, 1, 64], [32, 1, 64], [1, 10, 10, 1]),
        "linear4": ([48, 24], [6, 8, 4, 6], [1, 5, 20, 5, 1]),
        "clamp": ([32, 1, 64], [32, 1, 64], [1, 10, 20, 1]),
        "fullrank": ([8, 4, 6], [8, 4, 6], [1, 8, 6, 1]),
        "tall": ([64, 1, 16], [64, 1, 16], [1, 12, 12, 1]),
    }
    out = {}
    meta = {}
    for name, (xshape, shapes, ranks) in cases.items():
        for kind in ("gauss", "decay"):
            x = rng.standard_normal(xshape).astype(np.float32) if kind == "gauss" else decaying(xshape, rng)
            r = list(ranks)
            cores = ttd.ten2tt(x.reshape(shapes), list(shapes), r)
            rec = ttd.tt2ten(cores, xshape)
            key = f"{name}_{kind}"
            out[key + "_x"] = x
            out[key + "_rec"] = rec
            for i, c in enumerate(cores):
                out[f"{key}_core{i}"] = c
            meta[key] = {"x_shape": xshape, "tt_shapes": shapes, "ranks_in": ranks, "ranks_out": r,
                         "n_cores": len(cores), "dtype": str(cores[0].dtype)}
    np.savez_compressed(os.path.join(HERE, "g1_ten2tt.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g1_ten2tt.json"), "w"), indent=1)
    print("G1", len(meta), "cases")


# ---------------------------------------------------------------- G2 / G3
class HP:
    pass


class Toy(torch.nn.Module):
    def __init__(self, shapes, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        for name, shp in shapes.items():
            mod = self
            parts = name.split(".")
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    setattr(mod, p, torch.nn.Module())
                mod = getattr(mod, p)
            fan_in = int(np.prod(shp[1:]))
            mod.register_parameter(parts[-1], torch.nn.Parameter(torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5))
        self.extra = torch.nn.Parameter(torch.randn(7, generator=g))  # not in the table


def g2_g3():
    shapes = {
        "layer1.conv2.weight": (16, 16, 3, 3),
        "layer2.conv1.weight": (32, 64, 1, 1),
        "layer2.conv3.weight": (32, 16, 1, 1),     # clamps [1,10,20,1] -> [1,10,16,1]? (10x16 unfolding)
        "head.fc.weight": (48, 24),
    }
    hp = HP()
    hp.tt_shapes = {
        "layer1.conv2.weight": [4, 4, 9, 4, 4],
        "layer2.conv1.weight": [32, 1, 64],
        "layer2.conv3.weight": [32, 1, 16],
        "head.fc.weight": (6, 8, 4, 6),
    }
    hp.ranks = {
        "layer1.conv2.weight": [1, 4, 12, 12, 4, 1],
        "layer2.conv1.weight": [1, 10, 10, 1],
        "layer2.conv3.weight": [1, 12, 20, 1],
        "head.fc.weight": (1, 5, 20, 5, 1),
    }
    model = Toy(shapes, seed=7)
    out = {"w__" + k: v.detach().numpy().copy() for k, v in model.named_parameters() if k in shapes}
    a = ref_admm.ADMM(model, 0.01, hp, "tt", "cpu", verbose=False, log=True)
    a.update(update_u=False)
    for k in shapes:
        out[f"z_init__{k}"] = a.z[k].numpy().copy()
    g = torch.Generator().manual_seed(99)
    for it in range(3):
        # emulate an optimiser step between projections
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k in shapes:
                    p.add_(0.02 * torch.randn(p.shape, generator=g) * p.std())
                    out[f"w_it{it}__{k}"] = p.detach().numpy().copy()
        a.update()
        for k in shapes:
            out[f"z_it{it}__{k}"] = a.z[k].numpy().copy()
            out[f"u_it{it}__{k}"] = a.u[k].numpy().copy()
    meta = {"shapes": {k: list(v) for k, v in shapes.items()},
            "tt_shapes": {k: list(v) for k, v in hp.tt_shapes.items()},
            "ranks_in": {"layer1.conv2.weight": [1, 4, 12, 12, 4, 1], "layer2.conv1.weight": [1, 10, 10, 1],
                         "layer2.conv3.weight": [1, 12, 20, 1], "head.fc.weight": [1, 5, 20, 5, 1]},
            "ranks_after": {k: list(v) for k, v in hp.ranks.items()},
            "logger": {k: [float(x) for x in v] for k, v in a.logger.items()},
            "rho": 0.01}
    # G3: penalty + gradient
    loss = torch.zeros((), dtype=torch.float32)
    for p in model.parameters():
        p.grad = None
    total = a.append_admm_loss(loss)
    total.backward()
    meta["penalty"] = float(total)
    for k, p in model.named_parameters():
        if k in shapes:
            out[f"pen_grad__{k}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g2_admm_tt.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g2_admm_tt.json"), "w"), indent=1)
    print("G2/G3 tt: ranks after", meta["ranks_after"])

    # SVD format: 1x1 convs + linear, int and list rank entries
    shapes = {"a.conv.weight": (24, 40, 1, 1), "b.conv.weight": (40, 24, 1, 1), "fc.weight": (20, 30)}
    hp = HP()
    hp.ranks = {"a.conv.weight": 6, "b.conv.weight": [9], "fc.weight": 5}
    model = Toy(shapes, seed=11)
    out = {"w__" + k: v.detach().numpy().copy() for k, v in model.named_parameters() if k in shapes}
    a = ref_admm.ADMM(model, 0.001, hp, "svd", "cpu", log=True)
    a.update(update_u=False)
    for it in range(2):
        a.update()
        for k in shapes:
            out[f"z_it{it}__{k}"] = a.z[k].numpy().copy()
            out[f"u_it{it}__{k}"] = a.u[k].numpy().copy()
    meta = {"shapes": {k: list(v) for k, v in shapes.items()}, "ranks": hp.ranks,
            "logger": {k: [float(x) for x in v] for k, v in a.logger.items()}}
    np.savez_compressed(os.path.join(HERE, "g2_admm_svd.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g2_admm_svd.json"), "w"), indent=1)
    print("G2 svd ok")


# ---------------------------------------------------------------- G4
def g4():
    torch.manual_seed(5)
    hp = HP()
    hp.tt_shapes = {"c5": [4, 4, 9, 4, 4], "c3": [16, 9, 8], "c1": [32, 1, 16],
                    "l4": (6, 8, 4, 6), "l3": (12, 4, 6)}
    hp.ranks = {"c5": [1, 4, 10, 10, 4, 1], "c3": [1, 8, 6, 1], "c1": [1, 9, 9, 1],
                "l4": (1, 5, 16, 5, 1), "l3": (1, 7, 5, 1)}
    out, meta = {}, {}

    def rec(key, mod, x, extra):
        y = mod(x)
        out[key + "_x"] = x.numpy().copy()
        out[key + "_y"] = y.detach().numpy().copy()
        for k, v in mod.state_dict().items():
            out[f"{key}_sd__{k}"] = v.numpy().copy()
        meta[key] = dict(extra, state_keys=list(mod.state_dict().keys()))

    g = torch.Generator().manual_seed(123)
    # conv cases: (name, O, I, k, stride, padding, bias)
    for name, o, i, k, stride, pad, bias in [("c5", 16, 16, 3, 1, 1, True), ("c5", 16, 16, 3, 2, 1, False),
                                              ("c3", 16, 8, 3, 1, 0, True), ("c1", 32, 16, 1, 1, 0, True)]:
        w = torch.randn(o, i, k, k, generator=g) * 0.2
        b = torch.randn(o, generator=g) if bias else None
        x = torch.randn(2, i, 9, 9, generator=g)
        for cls_name in ("TTConv2dM", "TTConv2dR"):
            cls = getattr(ref_ttconv, cls_name)
            if cls_name == "TTConv2dM" and bias:
                # reference quirk: TTConv.py:150-151 adds the (O,) bias to a (B,O,H,W)
                # tensor, i.e. broadcasts it along W and raises unless W == O.  The
                # M variant is therefore recorded without bias only.
                continue
            hp2 = HP(); hp2.tt_shapes = {n: list(v) for n, v in hp.tt_shapes.items()}
            hp2.ranks = {n: list(v) for n, v in hp.ranks.items()}
            mod = cls(i, o, k, stride=stride, padding=pad, bias=bias, hp_dict=hp2, name=name,
                      dense_w=w.clone(), dense_b=None if b is None else b.clone())
            key = f"{cls_name}_{name}_s{stride}_b{int(bias)}"
            out[key + "_w"] = w.numpy().copy()
            if b is not None:
                out[key + "_b"] = b.numpy().copy()
            rec(key, mod, x, dict(cls=cls_name, name=name, o=o, i=i, k=k, stride=stride, padding=pad, bias=bias,
                                  tt_shapes=list(hp.tt_shapes[name]), ranks=list(hp.ranks[name])))
    for name, o, i, bias in [("l4", 48, 24, True), ("l3", 48, 6, False)]:
        w = torch.randn(o, i, generator=g) * 0.3
        b = torch.randn(o, generator=g) if bias else None
        x = torch.randn(3, 5, i, generator=g)
        for cls_name in ("TTLinearM", "TTLinearR"):
            cls = getattr(ref_ttlinear, cls_name)
            mod = cls(i, o, bias=bias, hp_dict=hp, name=name, dense_w=w.clone(),
                      dense_b=None if b is None else b.clone())
            key = f"{cls_name}_{name}_b{int(bias)}"
            out[key + "_w"] = w.numpy().copy()
            if b is not None:
                out[key + "_b"] = b.numpy().copy()
            rec(key, mod, x, dict(cls=cls_name, name=name, o=o, i=i, bias=bias,
                                  tt_shapes=list(hp.tt_shapes[name]), ranks=list(hp.ranks[name])))
    np.savez_compressed(os.path.join(HERE, "g4_layers.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g4_layers.json"), "w"), indent=1)
    print("G4", len(meta), "layer cases")


# ---------------------------------------------------------------- G5
def g5():
    """Full-size BASELINE layers: too big to store -> seeds + summary statistics."""
    from hp_dicts.tt_resnet50_hp import HyperParamsDictGeneralRatio3x as R50
    from hp_dicts.tt_resnet18_hp import HyperParamsDictGeneralRatio2x as R18
    from hp_dicts.tt_deit_small_patch16_224_hp import HyperParamsDictRatio2x as DEIT
    picks = [
        ("r50", R50, "layer4.0.conv2.weight", (512, 512, 3, 3)),
        ("r50", R50, "layer4.0.conv3.weight", (2048, 512, 1, 1)),
        ("r50", R50, "layer3.1.conv1.weight", (256, 1024, 1, 1)),
        ("r50", R50, "layer2.0.conv2.weight", (128, 128, 3, 3)),
        ("r18", R18, "layer3.0.conv1.weight", (256, 128, 3, 3)),
        ("deit", DEIT, "blocks.1.attn.qkv.weight", (1152, 384)),
    ]
    meta = {}
    samples = {}
    for tag, hp, name, shape in picks:
        seed = 1000 + len(meta)
        g = torch.Generator().manual_seed(seed)
        fan_in = int(np.prod(shape[1:]))
        w = (torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5).numpy()
        shapes = list(hp.tt_shapes[name]); ranks = list(hp.ranks[name])
        for dt in (np.float32, np.float64):
            x = w.astype(dt)
            if len(shape) == 4:
                t = np.transpose(x.reshape(shape[0], shape[1], -1), (0, 2, 1))
            else:
                t = x.reshape(shapes)
            r = list(ranks)
            # singular values of every unfolding (instrumented restatement of the loop
            # around the reference's own svd call sequence; cores come from ttd.ten2tt)
            cores = ttd.ten2tt(t, list(shapes), r)
            svals = []
            tt = t
            for i in range(len(shapes) - 1):
                tt = tt.reshape(r[i] * shapes[i], -1)
                u, s, v = np.linalg.svd(tt, full_matrices=False)
                svals.append(s[: r[i + 1]].astype(np.float64).tolist())
                tt = np.dot(np.diag(s[: r[i + 1]]), v[: r[i + 1]])
            if len(shape) == 4:
                z = ttd.tt2ten(cores, (shape[0], shape[2] * shape[3], shape[1]))
                z = np.transpose(z, (0, 2, 1)).reshape(shape)
            else:
                z = ttd.tt2ten(cores, shape)
            idx = np.random.default_rng(seed).integers(0, z.size, 64)
            key = f"{tag}:{name}:{np.dtype(dt).name}"
            meta[key] = {"seed": seed, "shape": list(shape), "tt_shapes": shapes, "ranks": ranks, "ranks_out": r,
                         "norm_z": float(np.linalg.norm(z.astype(np.float64))),
                         "norm_w_minus_z": float(np.linalg.norm((x - z).astype(np.float64))),
                         "norm_w": float(np.linalg.norm(x.astype(np.float64))),
                         "svals": svals}
            samples[key + ":idx"] = idx
            samples[key + ":z"] = z.reshape(-1)[idx].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "g5_fullsize.npz"), **samples)
    json.dump(meta, open(os.path.join(HERE, "g5_fullsize.json"), "w"))
    print("G5", len(meta))


# ---------------------------------------------------------------- G6
def g6():
    res = {}
    for script in ("numeric_example1.py", "numeric_example2.py", "numeric_example3.py"):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            src = open(os.path.join(REF, script)).read()
            exec(compile(src, script, "exec"), {"__name__": "__main__"})
        res[script] = buf.getvalue().strip().splitlines()
    json.dump(res, open(os.path.join(HERE, "g6_numeric_examples.json"), "w"), indent=1)
    print("G6", res)


if __name__ == "__main__":
    g1(); g2_g3(); g4(); g5(); g6()
