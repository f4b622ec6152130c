"""Child process of tests/test_gpu_dist.py: one rank of a two-rank ADMM run (gloo) on cuda:0 with the REAL device plan.

    python tests/_dist_gpu_worker.py <rank> <world> <port> <out.npz>

Replays the G2 sequence (tests/golden/g2_admm_tt: three ADMM.update iterations recorded from the reference) with
`ADMM(process_group=...)`: layers sharded over the ranks (sched.latency_partition), state re-assembled by
`ADMM._exchange` (gloo: list-form all-gather through host staging).  Writes Z, U, logger and the final rank table."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
sys.path.insert(0, ROOT)


def g2_model(dev):
    import torch
    gd = os.path.join(ROOT, "tests", "golden")
    data = np.load(os.path.join(gd, "g2_admm_tt.npz"))
    meta = json.load(open(os.path.join(gd, "g2_admm_tt.json")))
    names = list(meta["shapes"])

    class HP:
        pass

    hp = HP()
    hp.tt_shapes = {k: list(v) for k, v in meta["tt_shapes"].items()}
    hp.tt_shapes["head.fc.weight"] = tuple(hp.tt_shapes["head.fc.weight"])
    hp.ranks = {k: list(v) for k, v in meta["ranks_in"].items()}
    hp.ranks["head.fc.weight"] = tuple(hp.ranks["head.fc.weight"])

    class NamedParams(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.names = names
            self.flat = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.array(data["w__" + k])).to(dev))
                                                for k in names])

        def named_parameters(self, *a, **k):
            return iter(zip(self.names, self.flat))

    return NamedParams(), hp, data, meta, names


def replay(dev, process_group=None):
    import torch
    from tadmm.admm import ADMM
    model, hp, data, meta, names = g2_model(dev)
    a = ADMM(model, meta["rho"], hp, "tt", dev, log=True, process_group=process_group)
    a.update(update_u=False)
    out = {"zinit__" + k: a.z[k].cpu().numpy().copy() for k in names}
    for it in range(3):
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(torch.from_numpy(data[f"w_it{it}__{k}"]))
        a.update()
        for k in names:
            out[f"z_it{it}__{k}"] = a.z[k].cpu().numpy().copy()
            out[f"u_it{it}__{k}"] = a.u[k].cpu().numpy().copy()
    for k in names:
        out["log__" + k] = np.array(a.logger[k], dtype=np.float64)
    out["ranks_json"] = np.array(json.dumps({k: list(v) for k, v in hp.ranks.items()}))
    out["owned_json"] = np.array(json.dumps([a._names[i] for i in a._owned]))
    return out


def main():
    rank, world, port, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    out = replay(dev, dist.group.WORLD)
    np.savez(path, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
