"""The oracle (oracle/tt_oracle.py) against golden vectors recorded from the real
reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import tt_oracle as O


def _load(golden_dir, stem):
    return np.load(os.path.join(golden_dir, stem + ".npz")), json.load(open(os.path.join(golden_dir, stem + ".json")))


def test_g1_ten2tt_tt2ten(golden_dir):
    data, meta = _load(golden_dir, "g1_ten2tt")
    assert len(meta) == 14
    for key, m in meta.items():
        x = data[key + "_x"]
        ranks = list(m["ranks_in"])
        cores = O.ten2tt(x.reshape(m["tt_shapes"]), list(m["tt_shapes"]), ranks)
        assert ranks == m["ranks_out"], key              # bit-exact rank selection incl. clamp
        assert len(cores) == m["n_cores"]
        ref_cores = [data[f"{key}_core{i}"] for i in range(m["n_cores"])]
        cores = O.gauge_align_tt(cores, ref_cores)
        for c, rc in zip(cores, ref_cores):
            assert c.shape == rc.shape and c.dtype == rc.dtype
            np.testing.assert_allclose(c, rc, rtol=0, atol=2e-6 * max(1.0, np.abs(rc).max()))
        rec = O.tt2ten(cores, m["x_shape"])
        np.testing.assert_allclose(rec, data[key + "_rec"], rtol=0, atol=2e-6 * np.abs(x).max())


def test_g1_clamp_mutates_rank_list(golden_dir):
    _, meta = _load(golden_dir, "g1_ten2tt")
    assert meta["clamp_gauss"]["ranks_in"] == [1, 10, 20, 1]
    assert meta["clamp_gauss"]["ranks_out"] == [1, 10, 10, 1]


def test_g2_admm_update_tt(golden_dir):
    data, meta = _load(golden_dir, "g2_admm_tt")
    names = list(meta["shapes"])
    ranks = {k: list(v) for k, v in meta["ranks_in"].items()}
    ranks["head.fc.weight"] = tuple(ranks["head.fc.weight"])
    w = {k: data["w__" + k].copy() for k in names}
    u = {k: np.zeros_like(w[k]) for k in names}
    z, _ = O.admm_update(w, u, "tt", ranks, meta["tt_shapes"], update_u=False)
    for k in names:
        np.testing.assert_allclose(z[k], data["z_init__" + k], atol=3e-6)
    logger = {k: [] for k in names}
    for it in range(3):
        w = {k: data[f"w_it{it}__{k}"].copy() for k in names}
        z, resid = O.admm_update(w, u, "tt", ranks, meta["tt_shapes"])
        for k in names:
            np.testing.assert_allclose(z[k], data[f"z_it{it}__{k}"], atol=3e-6)
            np.testing.assert_allclose(u[k], data[f"u_it{it}__{k}"], atol=1e-5)
            logger[k].append(resid[k])
    # conv path clamps the table in place, linear path does not (admm.py:94 vs :105)
    assert {k: list(v) for k, v in ranks.items()} == meta["ranks_after"]
    for k in names:
        np.testing.assert_allclose(logger[k], meta["logger"][k], rtol=1e-5)
    # G3 penalty + gradient at the final state
    loss, grads = O.admm_penalty(w, z, u, meta["rho"])
    assert abs(loss - meta["penalty"]) <= 1e-5 * abs(meta["penalty"])
    for k in names:
        np.testing.assert_allclose(grads[k], data["pen_grad__" + k], rtol=1e-4, atol=1e-7)


def test_g2_admm_update_svd(golden_dir):
    data, meta = _load(golden_dir, "g2_admm_svd")
    names = list(meta["shapes"])
    w = {k: data["w__" + k].copy() for k in names}
    u = {k: np.zeros_like(w[k]) for k in names}
    O.admm_update(w, u, "svd", meta["ranks"], None, update_u=False)
    for it in range(2):
        z, resid = O.admm_update(w, u, "svd", meta["ranks"], None)
        for k in names:
            assert z[k].shape == tuple(meta["shapes"][k])
            np.testing.assert_allclose(z[k], data[f"z_it{it}__{k}"], atol=3e-6)
            np.testing.assert_allclose(u[k], data[f"u_it{it}__{k}"], atol=1e-5)
            assert abs(resid[k] - meta["logger"][k][it]) <= 1e-5 * meta["logger"][k][it]


def _cores_from_sd(data, key, keys, prefix):
    idx = sorted(int(k.split(".")[1]) for k in keys if k.startswith(prefix + "."))
    return [data[f"{key}_sd__{prefix}.{i}"] for i in idx]


def test_g4_layers(golden_dir):
    data, meta = _load(golden_dir, "g4_layers")
    for key, m in meta.items():
        x, y = data[key + "_x"], data[key + "_y"]
        w = data[key + "_w"]
        b = data[key + "_b"] if m["bias"] else None
        if m["cls"].startswith("TTLinear"):
            cores_ref = _cores_from_sd(data, key, m["state_keys"], "tt_cores")
            cores = O.ten2tt(w, list(m["tt_shapes"]), list(m["ranks"]))
            cores = O.gauge_align_tt(cores, cores_ref)
            for c, rc in zip(cores, cores_ref):
                np.testing.assert_allclose(c, rc, atol=3e-6)
            if m["cls"] == "TTLinearM":
                out = O.ttlinear_m_forward(x, cores_ref, m["o"], b)
            else:
                out = x @ O.tt_recover_weight(cores_ref, m["o"], m["i"]).T + (0 if b is None else b)
            np.testing.assert_allclose(out, y, atol=2e-5)
        elif m["cls"] == "TTConv2dM":
            outc = _cores_from_sd(data, key, m["state_keys"], "out_tt_cores")
            inc = _cores_from_sd(data, key, m["state_keys"], "in_tt_cores")
            ck = data[f"{key}_sd__core_kernel"]
            # decomposition parity: unfold + ten2tt + middle-core permute (TTConv.py:96-109)
            cores = O.ten2tt(O.conv_unfold(w), list(m["tt_shapes"]), list(m["ranks"]))
            mid = len(outc)
            ref_all = outc + [np.transpose(ck.reshape(ck.shape[0], ck.shape[1], -1), (0, 2, 1))] + inc
            cores = O.gauge_align_tt(cores, ref_all)
            for c, rc in zip(cores, ref_all):
                np.testing.assert_allclose(c, rc, atol=3e-6)
            assert len(cores) == mid + 1 + len(inc)
            s, p = m["stride"], m["padding"]
            out = O.ttconv2d_m_forward(x, inc, ck, outc, None, (s, s), (p, p))
            np.testing.assert_allclose(out, y, atol=3e-5)
            # structural identity: TT-M forward == dense conv with the projected kernel
            zk = O.prune_conv_rank_tt(w, list(m["tt_shapes"]), list(m["ranks"]))
            np.testing.assert_allclose(O.conv2d_nchw(x, zk, (s, s), (p, p)), y, atol=3e-5)
        else:  # TTConv2dR: flat-buffer TT (TTConv.py:285-288 has no (0,2,1) transpose)
            outc = _cores_from_sd(data, key, m["state_keys"], "out_tt_cores")
            inc = _cores_from_sd(data, key, m["state_keys"], "in_tt_cores")
            cc = data[f"{key}_sd__conv_core"]
            k2 = m["k"] * m["k"]
            wrec = O.tt2ten(outc + [cc] + inc, (m["o"], k2, m["i"])).reshape(m["o"], m["i"], m["k"], m["k"])
            s, p = m["stride"], m["padding"]
            out = O.conv2d_nchw(x, wrec, (s, s), (p, p))
            if b is not None:
                out = out + b.reshape(1, -1, 1, 1)
            np.testing.assert_allclose(out, y, atol=3e-5)


def test_g5_fullsize_layer_statistics(golden_dir):
    """One full-size BASELINE layer (ResNet-50 layer2.0.conv2) re-derived from its seed."""
    import torch
    meta = json.load(open(os.path.join(golden_dir, "g5_fullsize.json")))
    data = np.load(os.path.join(golden_dir, "g5_fullsize.npz"))
    key = "r50:layer2.0.conv2.weight:float32"
    m = meta[key]
    g = torch.Generator().manual_seed(m["seed"])
    shape = tuple(m["shape"])
    w = (torch.randn(shape, generator=g) * (2.0 / np.prod(shape[1:])) ** 0.5).numpy()
    ranks = list(m["ranks"])
    z = O.prune_conv_rank_tt(w, m["tt_shapes"], ranks)
    assert ranks == m["ranks_out"]
    assert abs(np.linalg.norm(z.astype(np.float64)) - m["norm_z"]) <= 2e-6 * m["norm_z"]
    assert abs(np.linalg.norm((w - z).astype(np.float64)) - m["norm_w_minus_z"]) <= 2e-6 * m["norm_w"]
    np.testing.assert_allclose(z.reshape(-1)[data[key + ":idx"]], data[key + ":z"], atol=3e-6 * np.abs(z).max())


def test_tucker_invariants_unpinned():
    """Tucker parity is UNPINNED (tensorly absent): invariants only."""
    rng = np.random.default_rng(3)
    w = rng.standard_normal((16, 12, 3, 3)).astype(np.float32)
    core, (uo, ui), errs = O.partial_tucker(w, [6, 5], return_trace=True)
    assert core.shape == (6, 5, 3, 3) and uo.shape == (16, 6) and ui.shape == (12, 5)
    np.testing.assert_allclose(uo.T @ uo, np.eye(6), atol=1e-5)
    np.testing.assert_allclose(ui.T @ ui, np.eye(5), atol=1e-5)
    assert all(errs[i + 1] <= errs[i] + 1e-6 for i in range(len(errs) - 1))
    z = O.tucker_to_tensor(core, [uo, ui])
    z2 = O.prune_rank_tk(z, [6, 5])
    np.testing.assert_allclose(z2, z, atol=1e-5)   # idempotent
    # linear (2-D) case is a two-sided subspace projection
    w2 = rng.standard_normal((20, 14)).astype(np.float32)
    z = O.prune_rank_tk(w2, [5, 4])
    assert np.linalg.matrix_rank(z.astype(np.float64), tol=1e-5) <= 4


def test_unsupported_ndim_raises():
    with pytest.raises(Exception, match="unsupported layer"):
        O.project_layer(np.zeros((2, 2, 2), np.float32), "tt", [1, 2, 1], [2, 2])
