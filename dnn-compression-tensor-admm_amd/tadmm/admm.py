"""ADMM controller with the reference's surface (admm.py:15-149), projection on the MI355X.

Differences a caller can observe (all deliberate):
  * the projection of ALL layers runs as one grouped plan on the device -- no D2H/H2D round trip
    (admm.py:50-67) and no per-layer Python loop;
  * `self.z[name]` is updated in place instead of being re-bound to a new tensor each call;
  * with `process_group` given, layers are sharded over the ranks (latency-aware partition, sched.py) and
    Z / U / the residuals are re-assembled with ONE flat all-gather; the reference repeats the full projection on
    every rank.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops, sched, ttd
from ._cabi import KIND_SVD, KIND_TT_CONV, KIND_TT_LINEAR, TadmmError


def _multi(rank_entry) -> bool:
    return (not isinstance(rank_entry, int)) and len(rank_entry) > 1


class _PenaltyFn(torch.autograd.Function):
    """0.5*rho*sum||W-Z+U||^2 with gradient rho*(W-Z+U) (admm.py:80-85), one fused launch."""

    @staticmethod
    def forward(ctx, admm, loss, *params):
        val, grads = admm._penalty_forward(params)
        ctx.grads = grads          # views of a buffer allocated for THIS call (a second forward cannot clobber them)
        ctx.n = len(params)
        return loss + val.to(loss.dtype)

    @staticmethod
    def backward(ctx, gout):
        gs = torch._foreach_mul(ctx.grads, gout.to(ctx.grads[0].dtype)) if ctx.grads else []
        return (None, gout) + tuple(gs)


class ADMM:
    def __init__(self, model, rho, hp_dict, format, device, verbose=False, log=False, process_group=None):
        self.model = model
        self.init_rho = rho
        self.hp_dict = hp_dict
        self.format = format
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            # the reference passes torch.device(args.device) with the default '--device cuda' (engines.py:69,
            # :242); parameters report 'cuda:<index>', so resolve the index once instead of failing the
            # device comparison in _build
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.verbose = verbose
        self.log = log
        self.process_group = process_group
        if self.log:
            self.logger = {}
        if format == 'none':
            raise Exception('ERROR: Tensor format should be specified!')      # admm.py:27-28
        self.rho = self.init_rho
        self.u: Dict[str, torch.Tensor] = {}
        self.z: Dict[str, torch.Tensor] = {}
        for name, param in self.model.named_parameters():                       # admm.py:35-40
            if name in self.hp_dict.ranks:
                self.u[name] = torch.zeros(param.shape, dtype=torch.float32, device=self.device)
                self.z[name] = param.data.detach().clone().to(self.device, torch.float32)
                if self.log:
                    self.logger[name] = []
        self._plan = None
        self._plan_key = None
        self._tk_names: List[str] = []
        self._pen = None

    # ------------------------------------------------------------------ plan construction
    def _kind_of(self, name, param):
        ranks = self.hp_dict.ranks[name]
        if param.dim() == 4:                                                    # admm.py:47-58
            if self.format == 'tk' and _multi(ranks):
                return 'tk'
            if self.format == 'tt' and _multi(ranks):
                return KIND_TT_CONV
            return KIND_SVD
        if param.dim() == 2:                                                    # admm.py:59-68
            if self.format == 'tk':
                return 'tk'
            if self.format == 'tt':
                return KIND_TT_LINEAR
            return KIND_SVD
        raise Exception('ERROR: unsupported layer in ADMM!')                    # admm.py:69

    def _named(self):
        # The table's parameters, in model order.  Cached: walking named_parameters() of a full network on every
        # training step (append_admm_loss) costs more host time than the fused penalty launch takes on the GPU;
        # the Parameter objects are stable (`.data` swaps are caught by the data_ptr keys).
        c = getattr(self, "_named_cache", None)
        if c is None:
            c = [(n, p) for n, p in self.model.named_parameters() if n in self.hp_dict.ranks]
            self._named_cache = c
        return c

    def _build(self):
        named = self._named()
        layers, names, self._tk_names = [], [], []
        for name, p in named:
            kind = self._kind_of(name, p)
            if p.device != self.device or p.dtype != torch.float32 or not p.data.is_contiguous():
                raise TadmmError(-1, f"{name}: parameters must be contiguous float32 on {self.device}")
            if kind == 'tk':
                self._tk_names.append(name)
                continue
            entry = dict(kind=kind, W=p.data, U=self.u[name], Z=self.z[name], ranks=self.hp_dict.ranks[name])
            if kind != KIND_SVD:
                entry["tt_shapes"] = list(self.hp_dict.tt_shapes[name])
                entry["ranks"] = list(self.hp_dict.ranks[name])
            layers.append(entry)
            names.append(name)
        self._names = names
        self._owned = list(range(len(names)))
        if self.process_group is not None and torch.distributed.get_world_size(self.process_group) > 1:
            ws = torch.distributed.get_world_size(self.process_group)
            rk = torch.distributed.get_rank(self.process_group)
            # latency-aware split: a layer's chain of eigen-solves is latency-bound and layers on one rank share
            # their launches, so ranks are balanced on the modelled level-by-level time, not on FLOPs (sched.py)
            prof = [sched.layer_latency_profile(L["kind"], list(L["W"].shape), L.get("tt_shapes"), L["ranks"])
                    for L in layers]
            parts = sched.latency_partition(prof, ws)
            self._parts = parts
            self._owned = parts[rk]
            # form of the state exchange, decided ONCE from the backend (never by catching an exception around a
            # collective: RCCL reports real failures as RuntimeError too, and ranks that disagree about which collective
            # comes next hang): "nccl" (= RCCL) gathers one flat device buffer; every other backend (gloo: no device
            # all-gather) gathers the list form through host staging
            self._xchg_flat = torch.distributed.get_backend(self.process_group) == "nccl"
        self._plan = ops.ProjectionPlan([layers[i] for i in self._owned]) if self._owned else None
        # rank clamp side effect: the conv TT path clamps the shared table in place (admm.py:94,97), the
        # linear path works on a copy (admm.py:105).  The clamp is a pure function of the shapes, so every
        # rank applies it to every layer, owned or not.
        from ._cabi import clamp_ranks
        for i, L in enumerate(layers):
            tbl = self.hp_dict.ranks[names[i]]
            if L["kind"] == KIND_TT_CONV and isinstance(tbl, list):
                tbl[:] = clamp_ranks(L["tt_shapes"], list(tbl))
        self._plan_key = tuple((n, p.data_ptr()) for n, p in named)
        if self._tk_names:
            from . import tucker
            self._tk = tucker.TuckerProjector(self, self._tk_names)

    # ------------------------------------------------------------------ reference API
    def update(self, update_u=True):
        named = self._named()
        key = tuple((n, p.data_ptr()) for n, p in named)
        if self._plan_key != key:
            self._build()
        resid = {}
        if self._plan is not None:
            r = self._plan.run(update_u=update_u, use_u=True)
            if update_u and (self.log or self.verbose):
                vals = r.sqrt().cpu().tolist()
                for j, i in enumerate(self._owned):
                    resid[self._names[i]] = vals[j]
        if self._tk_names:
            resid.update(self._tk.run(update_u))
        if self.process_group is not None and torch.distributed.get_world_size(self.process_group) > 1:
            resid = self._exchange(update_u, resid)
        if update_u:
            for name, _ in named:                                               # admm.py:75-78
                if name not in resid:
                    continue
                if self.log:
                    self.logger[name].append(float(resid[name]))
                if self.verbose:
                    print('*INFO: {} in ADMM, norm(w-z)={}'.format(name, resid[name]))

    def _exchange(self, update_u, resid):
        """Sharded mode: ONE all-gather re-assembles the state -- every rank packs the Z (and U) of the layers it
        owns plus their residuals into a flat buffer, the buffers are gathered, and every rank unpacks the layers
        it does not own.  Returns the residuals of ALL layers (so every rank logs what the reference logs)."""
        dist = torch.distributed
        ws = dist.get_world_size(self.process_group)
        rk = dist.get_rank(self.process_group)
        fields = [self.z, self.u] if update_u else [self.z]
        def seg_len(part):      # floats: tensors, then (hi, lo) float pairs of the float64 residuals
            return sum(self.z[self._names[i]].numel() for i in part) * len(fields) + 2 * len(part)
        mx = max(1, max(seg_len(p) for p in self._parts))
        send = torch.zeros(mx, dtype=torch.float32, device=self.device)
        off = 0
        mine = self._parts[rk]
        for f in fields:
            for i in mine:
                t = f[self._names[i]]
                send[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
        if mine:
            r64 = torch.tensor([float(resid.get(self._names[i], 0.0)) for i in mine], dtype=torch.float64)
            hi = r64.float()
            lo = (r64 - hi.double()).float()
            send[off:off + len(mine)].copy_(hi)
            send[off + len(mine):off + 2 * len(mine)].copy_(lo)
        if self._xchg_flat:
            recv = torch.empty(ws * mx, dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(recv, send, group=self.process_group)
        else:
            host = torch.empty(ws * mx, dtype=torch.float32)
            dist.all_gather(list(host.split(mx)), send.cpu(), group=self.process_group)
            recv = host.to(self.device)
        out = dict(resid)
        for owner, part in enumerate(self._parts):
            seg = recv[owner * mx:(owner + 1) * mx]
            off = 0
            for f in fields:
                for i in part:
                    t = f[self._names[i]]
                    if owner != rk:
                        t.copy_(seg[off:off + t.numel()].view_as(t))
                    off += t.numel()
            if part:         # (the owner too: every rank logs the same transported value, bit for bit)
                hi = seg[off:off + len(part)].double().cpu()
                lo = seg[off + len(part):off + 2 * len(part)].double().cpu()
                for j, i in enumerate(part):
                    out[self._names[i]] = float(hi[j] + lo[j])
        return out

    def append_admm_loss(self, loss):
        params = [p for _, p in self._named()]
        if not params:
            return loss
        return _PenaltyFn.apply(self, loss, *params)

    def adjust_rho(self, epoch, epochs, factor=5):
        if epoch > int(0.85 * epochs):                                          # admm.py:87-89
            self.rho = factor * self.init_rho

    # ------------------------------------------------------------------ checkpointing (beyond the reference)
    def state_dict(self):
        """ADMM state for checkpoint/resume.  The reference never saves Z, U or rho (engines.py:333-347), so a
        resumed --admm run silently restarts from U=0, Z=proj(W) (engines.py:241-245); this closes that gap."""
        return {"rho": self.rho, "init_rho": self.init_rho, "format": self.format,
                "z": {k: v.detach().cpu() for k, v in self.z.items()},
                "u": {k: v.detach().cpu() for k, v in self.u.items()},
                "ranks": {k: (list(v) if not isinstance(v, int) else v) for k, v in self.hp_dict.ranks.items()},
                "logger": {k: list(v) for k, v in self.logger.items()} if self.log else None}

    def load_state_dict(self, state):
        if state["format"] != self.format:
            raise ValueError(f"checkpoint format {state['format']!r} != {self.format!r}")
        if set(state["z"]) != set(self.z):
            raise KeyError("checkpoint layers differ from the rank table")
        self.rho, self.init_rho = state["rho"], state["init_rho"]
        for k in self.z:                       # in place: the projection plan holds these pointers
            self.z[k].copy_(state["z"][k])
            self.u[k].copy_(state["u"][k])
        for k, r in state["ranks"].items():    # re-apply clamps recorded at save time
            tbl = self.hp_dict.ranks.get(k)
            if isinstance(tbl, list):
                tbl[:] = r
        if self.log and state.get("logger"):
            self.logger = {k: list(v) for k, v in state["logger"].items()}

    # ------------------------------------------------------------------ fused penalty
    def _penalty_forward(self, params):
        import ctypes as C
        names = [n for n, _ in self._named()]
        n = len(names)
        dev = self.device
        key = tuple(p.data_ptr() for p in params)
        if self._pen is None or self._pen["key"] != key:
            numel = [p.numel() for p in params]
            # gradient slots: byte offsets inside one flat buffer that is allocated per call (64-byte aligned)
            goff, o = [], 0
            for k in numel:
                goff.append(o)
                o += (k * 4 + 63) // 64 * 64
            ptrs = [p.data_ptr() for p in params] + [self.z[k].data_ptr() for k in names] + \
                   [self.u[k].data_ptr() for k in names] + [0] * n
            h = ops.Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
            self._pen = dict(key=key, h=h, gbytes=o, goff=goff, shapes=[tuple(p.shape) for p in params], nel=numel,
                             goff_dev=torch.tensor(goff, dtype=torch.int64).to(dev),
                             ptrs=torch.tensor(ptrs, dtype=torch.int64).to(dev),
                             numel=torch.tensor(numel, dtype=torch.int64).to(dev), total=int(sum(numel)),
                             partial=torch.empty(h.lib.tadmm_penalty_scratch_doubles(), dtype=torch.float64,
                                                 device=dev))
        pen = self._pen
        # pointers of z may be unchanged (in-place update) but re-check u/z identity cheaply
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        # fresh gradient storage per forward: two penalties alive in one step (or a changed rho) keep their own
        flat = torch.empty(pen["gbytes"], dtype=torch.uint8, device=dev)
        torch.add(pen["goff_dev"], flat.data_ptr(), out=pen["ptrs"][3 * n:])
        grads = [flat[o:o + k * 4].view(torch.float32).view(shp)
                 for o, k, shp in zip(pen["goff"], pen["nel"], pen["shapes"])]
        h = pen["h"]
        h.check(h.lib.tadmm_penalty(h.ptr, n, pen["ptrs"].data_ptr(), pen["numel"].data_ptr(), pen["total"],
                                    float(self.rho), float(self.rho), loss.data_ptr(), pen["partial"].data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream))
        return loss[0], grads

    # ------------------------------------------------------------------ per-layer projections (numpy in/out)
    def _project_np(self, z, kind, tt_shapes, ranks):
        # The reference-API `prune_*` methods are called layer by layer: keep one single-layer plan (with its
        # W/U/Z staging tensors and workspace) per distinct (kind, shape, modes, ranks) instead of building and
        # destroying a plan per call.
        z = np.ascontiguousarray(z, dtype=np.float32)
        rk = tuple(ranks) if not isinstance(ranks, int) else ranks
        key = (kind, z.shape, tuple(tt_shapes) if tt_shapes is not None else None, rk)
        cache = self.__dict__.setdefault("_np_plans", {})
        ent = cache.get(key)
        if ent is None:
            w = torch.empty(z.shape, dtype=torch.float32, device=self.device)
            entry = dict(kind=kind, W=w, U=torch.zeros_like(w), Z=torch.empty_like(w), ranks=ranks)
            if tt_shapes is not None:
                entry["tt_shapes"] = list(tt_shapes)
            if len(cache) >= 64:                      # bounded: drop the oldest entry
                cache.pop(next(iter(cache))).get("plan").close()
            ent = cache[key] = dict(plan=ops.ProjectionPlan([entry]), W=w, Z=entry["Z"])
        ent["W"].copy_(torch.from_numpy(z))
        ent["plan"].run(update_u=False, use_u=False)
        return ent["Z"].cpu().numpy(), ent["plan"].ranks[0]

    def prune_conv_rank_tt(self, z, name):                                      # admm.py:91-101
        tbl = self.hp_dict.ranks[name]
        out, ranks = self._project_np(z, KIND_TT_CONV, self.hp_dict.tt_shapes[name], list(tbl))
        if isinstance(tbl, list):
            tbl[:] = ranks
        return out

    def prune_linear_rank_tt(self, z, name):                                    # admm.py:103-111
        out, _ = self._project_np(z, KIND_TT_LINEAR, self.hp_dict.tt_shapes[name], list(self.hp_dict.ranks[name]))
        return out

    def prune_conv_rank_svd(self, z, name):                                     # admm.py:129-139
        zz = z.detach().cpu().numpy() if torch.is_tensor(z) else z
        out, _ = self._project_np(zz, KIND_SVD, None, self.hp_dict.ranks[name])
        return out.reshape(out.shape[0], out.shape[1], 1, 1)

    def prune_linear_rank_svd(self, z, name):                                   # admm.py:141-149
        zz = z.detach().cpu().numpy() if torch.is_tensor(z) else z
        out, _ = self._project_np(zz, KIND_SVD, None, self.hp_dict.ranks[name])
        return out

    def prune_conv_rank_tk(self, z, name):                                      # admm.py:113-119
        from . import tucker
        return tucker.project_numpy(z, self.hp_dict.ranks[name], self.device)

    def prune_linear_rank_tk(self, z, name):                                    # admm.py:121-127
        from . import tucker
        return tucker.project_numpy(z, self.hp_dict.ranks[name], self.device)
