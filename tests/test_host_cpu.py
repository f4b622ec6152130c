"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host logic (rank tables,
selector, rank clamp, scheduler, workloads, error behaviour).  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_header_symbol():
    from tadmm import _cabi
    lib = _cabi.load()
    assert lib.tadmm_version() >= 100
    hdr = open(os.path.join(ROOT, "include", "tadmm.h")).read()
    declared = set(re.findall(r"\b(tadmm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tadmm_status", "tadmm_kind"}
    assert declared, "header parse failed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tadmm.h but not exported"
    # and the Python binding table covers exactly the header
    assert set(_cabi.ABI) == declared


def test_struct_layout_matches_library():
    from tadmm import _cabi
    lib = _cabi.load()
    a, b = ctypes.c_int(), ctypes.c_int()
    assert lib.tadmm_abi_sizes(ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value == ctypes.sizeof(_cabi.LayerDesc) and b.value == ctypes.sizeof(_cabi.GemmDesc)


def test_missing_library_fails_loudly(monkeypatch):
    from tadmm import _cabi
    monkeypatch.setattr(_cabi, "_lib", None)
    monkeypatch.setattr(_cabi, "_LIB_NAME", "libdoes_not_exist.so")
    with pytest.raises(_cabi.TadmmLibraryError, match="no CPU fallback"):
        _cabi.load()


def test_no_device_fails_loudly():
    from tadmm import _cabi, ttd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_cabi.TadmmError):
        _cabi.Handle(0)
    with pytest.raises(_cabi.TadmmError, match="no CPU fallback"):
        ttd.ten2tt(np.zeros((4, 4), np.float32), [4, 4], [1, 2, 1])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dnn-compression-tensor-admm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|tt_oracle|importlib.*oracle", src, re.M), \
                    f"{f} imports the oracle"


@pytest.mark.parametrize("shapes,ranks,expect", [
    ([32, 1, 64], [1, 10, 20, 1], [1, 10, 10, 1]),
    ([32, 1, 16], [1, 12, 20, 1], [1, 12, 12, 1]),
    ([8, 8, 9, 8, 8], [1, 8, 64, 64, 8, 1], [1, 8, 64, 64, 8, 1]),
    ([4, 4], [1, 9, 1], [1, 4, 1]),
    ([2048, 1, 512], [1, 160, 70, 1], [1, 160, 70, 1]),
    ([512, 1, 2048], [1, 70, 160, 1], [1, 70, 70, 1]),       # ResNet-50 *special* table clamps
])
def test_rank_clamp_is_shape_only(shapes, ranks, expect):
    from tadmm import _cabi
    assert _cabi.clamp_ranks(shapes, ranks) == expect


def test_rank_clamp_matches_oracle_on_random_shapes():
    from oracle import tt_oracle as O
    from tadmm import _cabi
    rng = np.random.default_rng(0)
    for _ in range(25):
        d = int(rng.integers(2, 6))
        shapes = [int(rng.integers(1, 7)) for _ in range(d)]
        ranks = [1] + [int(rng.integers(1, 12)) for _ in range(d - 1)] + [1]
        r = list(ranks)
        O.ten2tt(rng.standard_normal(shapes).astype(np.float32), shapes, r)
        assert _cabi.clamp_ranks(shapes, ranks) == r, (shapes, ranks)


def test_hp_tables_and_selector():
    from tadmm import hp
    t = hp.get_hp_dict("resnet50", "3", "tt", "general")
    assert len(t.ranks) == 34 and t.tt_shapes["layer4.0.conv2.weight"] == [32, 16, 9, 16, 32]
    assert t.ranks["layer4.0.conv2.weight"] == [1, 30, 105, 105, 30, 1]
    assert hp.get_hp_dict("ttr_resnet50", "3") is t                   # prefix overrides format, cached object
    assert hp.get_hp_dict("resnet50", "3", "tt", "special") is not t
    d = hp.get_hp_dict("deit_small_patch16_224", "2", "tt")
    assert isinstance(d.ranks["blocks.0.attn.qkv.weight"], tuple)    # DeiT tables are tuples (immutable)
    assert hp.get_hp_dict("tkc_resnet32", "3").ranks["layer3.0.conv1.weight"] == [32, 25]   # tk_resnet32_hp.py:95
    assert hp.get_hp_dict("svd_mobilenetv2", "2").ranks["features.4.conv.0.weight"] == 20
    assert hp.get_hp_dict("resnet50", "3", "none") is None            # dense model -> None
    assert hp.get_hp_dict("unknown_net", "2", "tt") is None
    with pytest.raises(Exception, match="Unsupported compression ratio"):
        hp.get_hp_dict("resnet50", "7", "tt")
    with pytest.raises(ImportError):
        hp.get_hp_dict("resnet32", "5", "tt")                         # class named by the ladder, never defined
    # fresh copies do not share clamps
    f = hp.fresh_table("tt_resnet50_hp.HyperParamsDictSpecialRatio3x")
    f.ranks["layer4.1.conv1.weight"][2] = 1
    assert hp.fresh_table("tt_resnet50_hp.HyperParamsDictSpecialRatio3x").ranks["layer4.1.conv1.weight"][2] != 1


def test_hp_ladder_matches_reference_golden(golden_dir):
    """G7: every rung of utils.get_hp_dict (utils.py:258-400), recorded from the reference by
    tests/golden/make_hp_ladder.py -- returned table, None, the bare Exception or the ImportError."""
    import json
    import os
    from tadmm import hp
    gold = json.load(open(os.path.join(golden_dir, "g7_hp_ladder.json")))
    assert len(gold) > 2500
    for key, want in gold.items():
        name, ratio, fmt, ttt = key.split("|")
        try:
            r = hp.get_hp_dict(name, ratio, fmt, ttt)
            got = "None" if r is None else r.table
        except ImportError:
            got = "ImportError"
        except Exception as e:
            got = "Exception:" + str(e)
        assert got == want, (key, got, want)


def test_flop_model_matches_survey():
    from tadmm import sched, workloads
    from tadmm._cabi import KIND_TT_CONV, KIND_TT_LINEAR
    expect = {"resnet50_tt": (34, 20.10e6, 52.94), "resnet18_tt": (16, 10.99e6, 34.07),
              "deit_small_tt": (48, 21.23e6, 40.88)}
    for cfg, (n, numel, gflop) in expect.items():
        m, h, _ = workloads.build(cfg)
        tot = 0.0
        cnt = 0
        ne = 0
        for name, p in m.named_parameters():
            f = sched.layer_flops(KIND_TT_CONV if p.dim() == 4 else KIND_TT_LINEAR, list(p.shape),
                                  h.tt_shapes[name], list(h.ranks[name]))
            tot += f["svd"]
            ne += f["numel"]
            cnt += 1
        assert cnt == n and abs(ne - numel) < 0.01e6 and abs(tot / 1e9 - gflop) < 0.01


def test_lpt_partition_bounds():
    from tadmm import sched
    costs = [5, 3, 3, 2, 2, 2, 1]
    parts = sched.lpt_partition(costs, 3)
    assert sorted(sum(parts, [])) == list(range(7))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= (4.0 / 3.0) * 6 and sched.lpt_partition(costs, 3) == parts   # LPT bound, deterministic
    assert sched.lpt_partition(costs, 1) == [list(range(7))]
    assert [len(p) for p in sched.lpt_partition([1.0], 4)] == [1, 0, 0, 0]


def test_admm_error_behaviour_matches_reference():
    from tadmm.admm import ADMM

    class HP:
        ranks = {"w": [1, 2, 1]}
        tt_shapes = {"w": [2, 4]}

    class M(torch.nn.Module):
        def __init__(self, shape):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(shape))

    with pytest.raises(Exception, match="Tensor format should be specified"):      # admm.py:27-28
        ADMM(M((2, 4)), 1e-3, HP, "none", "cpu")
    a = ADMM(M((2, 2, 2)), 1e-3, HP, "tt", "cpu")
    with pytest.raises(Exception, match="unsupported layer in ADMM"):              # admm.py:69
        a.update()
    a = ADMM(M((2, 4)), 1e-3, HP, "tt", "cpu")
    assert set(a.u) == {"w"} and float(a.u["w"].abs().sum()) == 0 and a.rho == 1e-3
    a.adjust_rho(9, 10)
    assert a.rho == 5e-3                                                           # admm.py:87-89
    a.adjust_rho(1, 10)
    assert a.rho == 5e-3


def test_layer_constructors_shapes_and_keys():
    from tadmm import tk_layers, tt_layers

    class HP:
        tt_shapes = {"c": [4, 4, 9, 4, 4], "l": (6, 8, 4, 6)}
        ranks = {"c": [1, 4, 10, 10, 4, 1], "l": (1, 5, 16, 5, 1), "k": [6, 5]}

    m = tt_layers.TTConv2dM(16, 16, 3, padding=1, bias=False, hp_dict=HP, name="c")
    assert list(m.state_dict()) == ["core_kernel", "in_tt_cores.0", "in_tt_cores.1", "out_tt_cores.0", "out_tt_cores.1"]
    assert m.core_kernel.shape == (10, 10, 3, 3) and m.out_tt_order == 2 and m.in_tt_order == 2
    r = tt_layers.TTConv2dR(16, 16, 3, bias=True, hp_dict=HP, name="c")
    assert list(r.state_dict()) == ["conv_core", "bias", "out_tt_cores.0", "out_tt_cores.1", "in_tt_cores.0",
                                    "in_tt_cores.1"]
    assert r.conv_core.shape == (10, 9, 10)
    lin = tt_layers.TTLinearM(24, 48, hp_dict=HP, name="l")
    assert [tuple(c.shape) for c in lin.tt_cores] == [(1, 6, 5), (5, 8, 16), (16, 4, 5), (5, 6, 1)]
    with pytest.raises(ValueError, match="groups must be 1"):
        tt_layers.TTConv2dM(16, 16, 3, groups=2, hp_dict=HP, name="c")
    with pytest.raises(AssertionError):
        tt_layers.TTLinearM(25, 48, hp_dict=HP, name="l")
    k = tk_layers.TKConv2dC(12, 16, 3, hp_dict=HP, name="k")
    assert list(k.state_dict()) == ["first_kernel", "core_kernel", "last_kernel", "bias"]
    assert k.first_kernel.shape == (5, 12, 1, 1) and k.core_kernel.shape == (6, 5, 3, 3) and k.last_kernel.shape == (16, 6, 1, 1)
    kl = tk_layers.TKLinearM(12, 16, hp_dict=HP, name="k")
    assert list(kl.state_dict()) == ["first_factor", "core_tensor", "last_factor", "bias"]


def test_eigen_problem_size_limit_is_reported_not_hidden():
    """Eigen-problems up to N = 8160 are taken: rows that do not fit the LDS-resident pair kernels (N > 1152: the
    4096-wide `pre_logits.fc1/fc2` entries of the tk_vgg16 / tk_vgg16_bn tables) go through the streamed pair kernel.
    Beyond that the plan is refused with TADMM_ERR_UNSUPPORTED and a message -- no silent wrong answer, no crash."""
    import ctypes as C
    from tadmm import _cabi
    lib = _cabi.load()
    h = C.c_void_p()
    lib.tadmm_create(0, C.byref(h))          # no GPU here: the handle is still usable for host-side sizing
    assert h.value
    size = C.c_size_t()
    ok = (_cabi.LayerDesc * 1)(_cabi.make_layer_desc(_cabi.KIND_SVD, [1024, 1024], None, 64))
    assert lib.tadmm_plan_workspace_bytes(h, 1, ok, C.byref(size)) == 0 and size.value > 0
    big = (_cabi.LayerDesc * 1)(_cabi.make_layer_desc(_cabi.KIND_SVD, [4096, 4096], None, 512))
    assert lib.tadmm_plan_workspace_bytes(h, 1, big, C.byref(size)) == 0 and size.value > 4096 * 4096 * 8
    tk = (_cabi.LayerDesc * 1)(_cabi.make_layer_desc(_cabi.KIND_TUCKER2, [4096, 512, 7, 7], None, [512, 256]))
    assert lib.tadmm_tucker_workspace_bytes(h, 1, tk, C.byref(size)) == 0 and size.value > 0
    huge = (_cabi.LayerDesc * 1)(_cabi.make_layer_desc(_cabi.KIND_SVD, [16384, 9000], None, 512))
    rc = lib.tadmm_plan_workspace_bytes(h, 1, huge, C.byref(size))
    assert rc < 0
    msg = lib.tadmm_last_error(h).decode()
    assert "9000" in msg and "exceeds" in msg
    tk = (_cabi.LayerDesc * 1)(_cabi.make_layer_desc(_cabi.KIND_TUCKER2, [16384, 9000, 3, 3], None, [512, 256]))
    assert lib.tadmm_tucker_workspace_bytes(h, 1, tk, C.byref(size)) < 0
    assert "exceeds" in lib.tadmm_last_error(h).decode()
    lib.tadmm_destroy(h)


def test_chain_eligibility_rules_are_pure_host_logic():
    """Which forward path a layer takes is decided on the host from shapes alone (no GPU needed): the fused conv launch
    for output rows of <= 64 pixels whose halo and intermediates fit the LDS, the fused linear chain for middle ranks
    <= 256."""
    import torch
    from tadmm import functional as HF
    from tadmm import ops
    x = torch.zeros(2, 64, 8, 8)
    assert ops.conv_chain_fits(x, 23, 25, (3, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_chain_fits(x.to(torch.bfloat16), 220, 220, (3, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_chain_fits(x, 220, 220, (3, 3), (1, 1), (1, 1), (1, 1))              # three planes of 224: 32-pixel tiles
    assert not ops.conv_chain_fits(torch.zeros(2, 64, 14, 14), 250, 250, (5, 5), (1, 1), (2, 2), (1, 1))   # 5 halo rows of 256 ch
    assert ops.conv_chain_fits(torch.zeros(2, 64, 14, 14), 23, 25, (3, 3), (1, 1), (1, 1), (1, 1))       # row tiles
    assert not ops.conv_chain_fits(torch.zeros(2, 8, 112, 112), 8, 8, (3, 3), (1, 1), (1, 1), (1, 1))    # rows of 112 pixels
    assert not ops.conv_chain_fits(torch.zeros(2, 8, 56, 56), 8, 8, (7, 7), (1, 1), (3, 3), (1, 1))      # halo of 7 rows > 192 px
    assert not ops.conv_chain_fits(x, 300, 25, (3, 3), (1, 1), (1, 1), (1, 1))           # rank > 256
    assert not ops.conv_chain_fits(x.double(), 23, 25, (3, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_chain_fits(x, 23, 25, (3, 3), (2, 2), (1, 1), (1, 1))                 # stride 2: 8x8 -> 4x4
    assert not ops.conv_chain_fits(x, 23, 25, (9, 9), (1, 1), (0, 0), (1, 1))             # empty output plane
    # capability vs choice: fp32 with many row tiles fits but does not pay; bf16 always takes it
    big = torch.zeros(2, 64, 28, 28)
    assert ops.conv_chain_fits(big, 72, 72, (3, 3), (1, 1), (1, 1), (1, 1)) and not ops.conv_chain_pays(big, 72, 72, (3, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_chain_pays(big.to(torch.bfloat16), 72, 72, (3, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_chain_pays(x, 220, 220, (3, 3), (1, 1), (1, 1), (1, 1))              # 8x8 fp32: two workgroups per image
    assert HF.fused_rank_ok(256) and not HF.fused_rank_ok(257) and not HF.fused_rank_ok(0)


def test_lane_split_rule_on_the_benchmark_tables():
    """tadmm_lane_split is a pure host function of the layer descriptors (what tadmm_plan_create does): ResNet-50 puts the
    nine 3x3 kernels of layer3 / layer4 on the priority lane, DeiT-small (48 like layers) is halved, small tables and
    TADMM_LANES=1 stay in one lane."""
    import ctypes as C
    from tadmm import _cabi, workloads
    from tadmm._cabi import FLAG_SKIP_ROTATIONS, KIND_TT_CONV, KIND_TT_LINEAR
    lib = _cabi.load()

    def split(cfg, take=None):
        m, hp, _ = workloads.build(cfg)
        items = [(n, p) for n, p in m.named_parameters()][:take]
        descs = (_cabi.LayerDesc * len(items))()
        for i, (n, p) in enumerate(items):
            kind = KIND_TT_CONV if p.dim() == 4 else KIND_TT_LINEAR
            descs[i] = _cabi.make_layer_desc(kind, list(p.shape), hp.tt_shapes[n], list(hp.ranks[n]), FLAG_SKIP_ROTATIONS)
        out = (C.c_int32 * len(items))()
        lanes = lib.tadmm_lane_split(len(items), descs, out)
        return lanes, [n for n, _ in items], list(out)

    lanes, names, lane_of = split("resnet50_tt")
    assert lanes == 2
    long_chain = [n for n, l in zip(names, lane_of) if l == 0]
    assert len(long_chain) == 9 and all(".conv2." in n and n[:6] in ("layer3", "layer4") for n in long_chain)
    lanes, names, lane_of = split("deit_small_tt")
    assert lanes == 2 and lane_of == [0] * 24 + [1] * 24
    lanes, _, lane_of = split("resnet50_tt", take=3)
    assert lanes == 1 and set(lane_of) == {0}
    os.environ["TADMM_LANES"] = "1"
    try:
        assert split("resnet50_tt")[0] == 1
    finally:
        del os.environ["TADMM_LANES"]


def test_vgg16_workload_shapes_follow_the_tables():
    """The VGG-16 Tucker tables (tk_vgg16_hp / tk_vgg16_bn_hp) key the convolutions by their `features.N` index, which
    differs between the plain and the BN network; the synthetic workload maps the i-th table key to the i-th of
    convolutions 2..13 and the timm-style head to a 7x7 and a 1x1 convolution.  Host logic only (no tensors built)."""
    from tadmm import hp, workloads
    for cfg in ("vgg16_tk", "vgg16_bn_tk"):
        key, fmt, fn = workloads.CONFIGS[cfg]
        assert fmt == "tk"
        table = hp.table(key)
        shapes = {n: fn(n) for n in table.ranks}
        feats = [s for n, s in shapes.items() if n.startswith("features.")]
        assert len(feats) == 12 and feats[0] == (64, 64, 3, 3) and feats[-1] == (512, 512, 3, 3)
        assert all(a[0] <= b[0] for a, b in zip(feats, feats[1:]))                 # channel counts never shrink
        assert shapes["pre_logits.fc1.weight"] == (4096, 512, 7, 7)
        for n, s in shapes.items():
            r = table.ranks[n]
            if len(r) == 2:
                assert r[0] <= s[0] and r[1] <= s[1], (n, r, s)                     # Tucker ranks fit their modes
            else:
                assert s == (4096, 4096, 1, 1) and r[0] <= 4096                     # the SVD entry of fc2


def test_every_table_with_known_shapes_sizes_a_plan():
    """Every shipped rank table whose architecture is re-derived in tadmm/workloads.py (ResNet-18/50 ImageNet, ResNet-32/56
    CIFAR, DeiT-small/tiny, ViT-small, VGG-16 / VGG-16-BN, DenseNet-40/121/201, MobileNetV2 CIFAR / ImageNet: 36 of the 37 tables, TT, Tucker and SVD entries) passes the
    host-side plan sizing of the C ABI with the dispatch of admm.py:47-69 -- none is refused (the VGG tables were,
    before the streamed Jacobi pairs).  DenseNet-264 (a table whose entries are shapes, not ranks) is not.  Host logic only."""
    import ctypes as C
    from tadmm import _cabi, hp, workloads
    lib = _cabi.load()
    h = C.c_void_p()
    lib.tadmm_create(0, C.byref(h))
    assert h.value
    keys = hp.table_keys()
    assert len(keys) == 37
    covered = 0
    for key in keys:
        fn = workloads.shape_fn_for(key)
        if fn is None:
            assert "densenet264" in key, key
            continue
        covered += 1
        fmt = key.split("_")[0]
        table = hp.fresh_table(key)
        plan_descs, tk_descs = [], []
        for name, ranks in table.ranks.items():
            shape = fn(name)
            many = not isinstance(ranks, int) and len(ranks) > 1
            if fmt == "tk" and many:
                tk_descs.append(_cabi.make_layer_desc(_cabi.KIND_TUCKER2, shape, None, list(ranks)))
            elif fmt == "tt" and many:
                kind = _cabi.KIND_TT_CONV if len(shape) == 4 else _cabi.KIND_TT_LINEAR
                plan_descs.append(_cabi.make_layer_desc(kind, shape, list(table.tt_shapes[name]), list(ranks)))
            else:
                plan_descs.append(_cabi.make_layer_desc(_cabi.KIND_SVD, shape, None, ranks))
        size = C.c_size_t()
        if plan_descs:
            arr = (_cabi.LayerDesc * len(plan_descs))(*plan_descs)
            rc = lib.tadmm_plan_workspace_bytes(h, len(plan_descs), arr, C.byref(size))
            assert rc == 0 and size.value > 0, (key, lib.tadmm_last_error(h).decode())
        if tk_descs:
            arr = (_cabi.LayerDesc * len(tk_descs))(*tk_descs)
            rc = lib.tadmm_tucker_workspace_bytes(h, len(tk_descs), arr, C.byref(size))
            assert rc == 0 and size.value > 0, (key, lib.tadmm_last_error(h).decode())
    assert covered == 36, covered
    lib.tadmm_destroy(h)


# ---------------------------------------------------------------- round 3: host logic added this round
def test_filter_block_is_capped_not_refused():
    """csrc/filter_host.h and its mirror in sched.py: a preferred block wider than 256 columns is capped while it keeps at
    least 1.15 of oversampling (ResNet-18 layer4: r = 210 / 220), refused below that (r = 236) and by the 0.56 N rule."""
    from tadmm import sched
    assert sched.filter_block_size(512, 105) == 160          # ResNet-50 layer4: align32(1.45 r)
    assert sched.filter_block_size(512, 130) == 192
    assert sched.filter_block_size(480, 210) == 256 and sched.filter_block_size(512, 220) == 256
    assert sched.filter_block_size(480, 236) == 0             # 256 / 236 = 1.08
    assert sched.filter_block_size(384, 256) == 0 and sched.filter_block_size(288, 256) == 0     # DeiT-small: nothing
    assert sched.filter_block_size(240, 138) == 0             # 0.56 N rule


def test_launch_memo_is_a_small_lru():
    from tadmm import ops
    m = ops._LruMemo()
    for i in range(ops._LruMemo.CAP + 40):
        m.store(("k", i), i)
    assert len(m) == ops._LruMemo.CAP
    assert m.lookup(("k", 0)) is None and m.lookup(("k", 50)) == 50
    m.store(("k", "new"), 1)                                   # ("k", 50) was just used: it survives, the oldest goes
    assert m.lookup(("k", 50)) == 50 and m.lookup(("k", 40)) is None


def test_inference_caches_are_dropped_by_mode_switches_and_loads():
    import torch
    from tadmm import functional as HF

    class L(HF.InferenceCacheMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

    layer = L()
    for trigger in (lambda: layer.eval(), lambda: layer.train(), lambda: layer.double(), lambda: layer.load_state_dict(layer.state_dict()),
                    lambda: layer.invalidate_caches()):
        layer.__dict__["_chain_cache"] = {"key": 1}
        layer.__dict__["_plane_cache"] = {"x": 2}
        trigger()
        assert "_chain_cache" not in layer.__dict__ and "_plane_cache" not in layer.__dict__
    p = layer.w
    k0 = HF.param_key(p)
    p.data = torch.ones(3, dtype=p.dtype)                      # new storage, same version counter
    assert HF.param_key(p) != k0


def test_pmc_traffic_is_only_stamped_on_a_matching_launch_count():
    import bench
    pm = {"k": {"hbm_bytes_per_launch_corrected": 123.0, "launches_per_step": 100.0}, "old": {"hbm_bytes_per_launch_corrected": 5.0}}
    meta = {"file": "profiles/x.json", "commit": "abc1234"}
    t, src = bench.pmc_traffic_for(pm, meta, "k", 104.0, 1)
    assert t == 123.0 and "abc1234" in src
    t, src = bench.pmc_traffic_for(pm, meta, "k", 130.0, 1)
    assert t is None and "dropped" in src
    assert bench.pmc_traffic_for(pm, meta, "old", 100.0, 1)[0] is None         # a pass without provenance
    assert bench.pmc_traffic_for(pm, meta, "k", 100.0, 2)[0] is None           # multi-GPU: no PMC pass


def test_tucker_flop_model():
    import bench
    f = bench.tucker_flops((64, 64, 3, 3), [25, 23], 3)
    assert f["numel"] == 36864 and f["eig"] == 8.0 * 64 ** 3 * (2 + 2 * 3)
    hosvd = 2.0 * 576 * 64 * 64 * 2
    per = 2.0 * 64 * 9 * 64 * 23 + 2.0 * (9 * 23) * 64 * 64 + 2.0 * 64 * 9 * 64 * 25 + 2.0 * (9 * 25) * 64 * 64 + 2.0 * 25 * 9 * 64 * 23
    final = 2.0 * 25 * 9 * 23 * 64 + 2.0 * 64 * 25 * 9 * 64
    assert abs(f["mfma"] - (hosvd + 3 * per + final)) < 1.0
