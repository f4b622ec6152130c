"""The sharded ADMM on the device (reference context engines.py:152-155, 241-245).

Two fresh child processes (started with subprocess, not a re-exec of this one), both on cuda:0, `gloo` backend, the
REAL `ProjectionPlan` behind `ADMM(process_group=...)`: each rank projects only the layers it owns, `ADMM._exchange`
re-assembles Z / U / residuals on both.  Expected: both ranks end with the single-process device result bit for bit
(the projection of a layer does not depend on which other layers share its plan... to rounding: grouped sweeps -- so
the single-process comparison is to 2e-7 and the rank-to-rank one is exact) and with the reference's golden G2
sequence within 1e-5."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_device_match_single_process_and_golden(tmp_path, golden_dir):
    import torch
    assert torch.cuda.is_available(), "needs the MI355X"
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "_dist_gpu_worker.py")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    r0, r1 = (np.load(o) for o in outs)
    own0, own1 = json.loads(str(r0["owned_json"])), json.loads(str(r1["owned_json"]))
    data = np.load(os.path.join(golden_dir, "g2_admm_tt.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g2_admm_tt.json")))
    names = list(meta["shapes"])
    assert own0 and own1 and not set(own0) & set(own1) and set(own0) | set(own1) == set(names)
    # single process, same device, same code path without a group
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_gpu_worker as W
    single = W.replay(torch.device("cuda:0"))
    for key in r0.files:
        if key.endswith("_json"):
            continue
        np.testing.assert_array_equal(r0[key], r1[key], err_msg=key)          # both ranks hold the identical full state
    assert str(r0["ranks_json"]) == str(r1["ranks_json"]) == str(single["ranks_json"])
    assert json.loads(str(r0["ranks_json"])) == meta["ranks_after"]           # the clamp ran on every rank, owned or not
    for k in names:
        s = np.abs(data["w__" + k]).max()
        np.testing.assert_allclose(r0["zinit__" + k], data["z_init__" + k], atol=REL * s)
        for it in range(3):
            s = np.abs(data[f"w_it{it}__{k}"]).max()
            np.testing.assert_allclose(r0[f"z_it{it}__{k}"], single[f"z_it{it}__{k}"], atol=2e-7 * s, err_msg=k)
            np.testing.assert_allclose(r0[f"u_it{it}__{k}"], single[f"u_it{it}__{k}"], atol=6e-7 * s, err_msg=k)
            np.testing.assert_allclose(r0[f"z_it{it}__{k}"], data[f"z_it{it}__{k}"], atol=REL * s, err_msg=k)
            np.testing.assert_allclose(r0[f"u_it{it}__{k}"], data[f"u_it{it}__{k}"], atol=3 * REL * s, err_msg=k)
        np.testing.assert_allclose(r0["log__" + k], meta["logger"][k], rtol=1e-4)
        np.testing.assert_allclose(r0["log__" + k], single["log__" + k], rtol=1e-6)
