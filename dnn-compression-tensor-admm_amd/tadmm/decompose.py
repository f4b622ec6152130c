"""The `--decompose` hand-off (reference main.py:82-90 -> model factories, e.g. resnet_inet_tt.py:444-483):
turn a dense `*_model.pt` state_dict (what the ADMM phase saves, engines.py:345-347) into the state_dict of
the factorised model, with exactly the keys the reference's layer classes register.

The reference does this inside every layer constructor on the CPU (`dense_w=dense_dict[w_name]`); here all
layers of the table are decomposed in ONE grouped device plan (TT / SVD) or by the device HOOI (Tucker), and
every tensor that is not in the rank table is copied through unchanged (resnet_inet_tt.py:444-449).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops, tucker
from ._cabi import KIND_TT_CONV, KIND_TT_LINEAR


def _prefix(name: str) -> str:
    assert name.endswith(".weight"), name
    return name[: -len("weight")]


def _split(tt_shapes, out_dim):
    prod = 1
    for i, n in enumerate(tt_shapes):
        prod *= n
        if prod == out_dim:
            return i + 1
    raise AssertionError("tt_shapes do not factor the output dimension")


def decompose_state_dict(dense: Dict[str, torch.Tensor], hp_dict, format: str, variant: str = "M",
                         device=None) -> Dict[str, torch.Tensor]:
    """format: 'tt' | 'tk'; variant: 'M' | 'R' | 'C' (TKConv2dC only) -- selects the layer class whose keys
    are emitted.  Returns a new state_dict on the CPU."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    out: Dict[str, torch.Tensor] = {}
    table = [k for k in dense if k in hp_dict.ranks]
    for k, v in dense.items():
        if k not in hp_dict.ranks:
            out[k] = v.detach().cpu().clone()
    if format == "tt":
        layers, meta = [], []
        for name in table:
            w = dense[name].detach().to(device, torch.float32).contiguous()
            shapes = list(hp_dict.tt_shapes[name])
            ranks = list(hp_dict.ranks[name])
            conv = w.dim() == 4
            if conv and variant == "R":
                # TTConv2dR quirk (TTConv.py:285-288): flat (O, I*k^2) buffer, modes out | k^2 | in, no transpose
                o = w.shape[0]
                n_out = _split(shapes, o)
                shapes = shapes[:n_out] + [w.shape[2] * w.shape[3]] + shapes[n_out + 1:]
                wk = w.reshape(o, -1)
                kind = KIND_TT_LINEAR
            else:
                wk, kind = w, (KIND_TT_CONV if conv else KIND_TT_LINEAR)
            layers.append(dict(kind=kind, W=wk, U=torch.zeros_like(wk), Z=torch.empty_like(wk), tt_shapes=shapes,
                               ranks=ranks))
            meta.append((name, conv, w.shape))
        if layers:
            plan = ops.ProjectionPlan(layers, want_cores=True)
            plan.run(update_u=False, use_u=False)
            for i, (name, conv, shape) in enumerate(meta):
                cores = [c.cpu().clone() for c in plan.core_tensors(i)]
                p = _prefix(name)
                n_out = _split(layers[i]["tt_shapes"], shape[0])
                if not conv:
                    for j, c in enumerate(cores):
                        out[f"{p}tt_cores.{j}"] = c
                else:
                    for j, c in enumerate(cores):
                        if j < n_out:
                            out[f"{p}out_tt_cores.{j}"] = c
                        elif j == n_out:
                            if variant == "R":
                                out[f"{p}conv_core"] = c
                            else:   # (r, k^2, r') -> (r, r', kh, kw)   TTConv.py:105-107
                                out[f"{p}core_kernel"] = c.permute(0, 2, 1).reshape(c.shape[0], c.shape[2], shape[2],
                                                                                    shape[3]).contiguous()
                        else:
                            out[f"{p}in_tt_cores.{j - n_out - 1}"] = c
            plan.close()
    elif format == "tk":
        ws = [dense[name].detach().to(device, torch.float32).contiguous() for name in table]
        res = tucker._plan_decompose(ws, [hp_dict.ranks[name] for name in table]) if table else []
        for name, w, (core, (u_out, u_in), _, _) in zip(table, ws, res):
            p = _prefix(name)
            first, last = u_in.t().contiguous().cpu(), u_out.contiguous().cpu()
            if w.dim() == 4 and variant == "C":
                out[f"{p}first_kernel"] = first[:, :, None, None]
                out[f"{p}core_kernel"] = core.cpu()
                out[f"{p}last_kernel"] = last[:, :, None, None]
            else:
                out[f"{p}first_factor"] = first
                out[f"{p}core_kernel" if (w.dim() == 4 and variant == "M") else f"{p}core_tensor"] = core.cpu()
                out[f"{p}last_factor"] = last
    else:
        raise Exception('ERROR: Tensor format should be specified!')
    return out
