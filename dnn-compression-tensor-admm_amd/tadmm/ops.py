"""Torch-facing wrappers over the C ABI: device memory, streams and pointer plumbing only.

Nothing here computes; every function hands raw device pointers of torch tensors
to libtadmm_hip.so on the current HIP stream.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _cabi
from ._cabi import (FLAG_SKIP_ROTATIONS, KIND_SVD, KIND_TT_CONV, KIND_TT_LINEAR, KIND_TUCKER2, GemmDesc, Handle,
                    LayerDesc, TadmmError, make_layer_desc)


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise TadmmError(-1, f"{what} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32 and what != "G":
        raise TadmmError(-1, f"{what} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise TadmmError(-1, f"{what} must be contiguous")


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class ProjectionPlan:
    """Grouped TT/SVD projection of a set of layers (ADMM.update, admm.py:42-78).

    layers: sequence of dicts with keys
        kind       : KIND_TT_CONV | KIND_TT_LINEAR | KIND_SVD
        W, U, Z    : float32 device tensors of identical shape (captured by pointer)
        tt_shapes  : list[int]   (TT kinds)
        ranks      : list[int] (TT) | int / [int] (SVD)
    """

    def __init__(self, layers: Sequence[dict], want_cores: bool = False, skip_rotations: bool = True):
        if not layers:
            raise ValueError("empty plan")
        dev = layers[0]["W"].device
        self.device = dev
        self.h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
        lib = self.h.lib
        n = len(layers)
        self.n = n
        flags = 0 if want_cores or not skip_rotations else FLAG_SKIP_ROTATIONS
        self._descs = (LayerDesc * n)()
        self._keep = []
        Wp = (C.c_void_p * n)()
        Up = (C.c_void_p * n)()
        Zp = (C.c_void_p * n)()
        Cp = (C.c_void_p * n)()
        self.cores: List[Optional[torch.Tensor]] = [None] * n
        self._core_shapes = []
        for i, L in enumerate(layers):
            for key in ("W", "U", "Z"):
                _require_cuda(L[key], key)
                if L[key].shape != L["W"].shape:
                    raise TadmmError(-1, f"{key} shape differs from W")
            self._descs[i] = make_layer_desc(L["kind"], list(L["W"].shape), L.get("tt_shapes"), L["ranks"], flags)
            Wp[i], Up[i], Zp[i] = L["W"].data_ptr(), L["U"].data_ptr(), L["Z"].data_ptr()
            self._keep.append((L["W"], L["U"], L["Z"]))
        size = C.c_size_t()
        self.h.check(lib.tadmm_plan_workspace_bytes(self.h.ptr, n, self._descs, C.byref(size)))
        # clamped ranks are a pure function of the shapes: compute them to size the core buffers
        self.ranks = []
        for i, L in enumerate(layers):
            d = self._descs[i]
            if L["kind"] == KIND_SVD:
                o, k = int(L["W"].shape[0]), int(L["W"].shape[1])
                r = min(int(d.ranks[0]), o, k)
                shapes, ranks = [o, k], [1, r, 1]
            else:
                shapes = [int(d.tt_shapes[j]) for j in range(d.d)]
                ranks = _cabi.clamp_ranks(shapes, [int(d.ranks[j]) for j in range(d.d + 1)])
            self.ranks.append(ranks)
            self._core_shapes.append([(ranks[j], shapes[j], ranks[j + 1]) for j in range(len(shapes))])
            if want_cores:
                tot = sum(a * b * c for a, b, c in self._core_shapes[i])
                self.cores[i] = torch.empty(tot, dtype=torch.float32, device=dev)
                Cp[i] = self.cores[i].data_ptr()
            else:
                Cp[i] = None
        self.workspace_bytes = int(size.value)
        self.workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=dev)
        self.resid_sq = torch.zeros(n, dtype=torch.float64, device=dev)
        self._plan = C.c_void_p()
        self.h.check(lib.tadmm_plan_create(self.h.ptr, n, self._descs, Wp, Up, Zp, Cp, self.workspace.data_ptr(),
                                           self.workspace_bytes, C.byref(self._plan)))

    def run(self, update_u: bool = True, use_u: bool = True) -> torch.Tensor:
        """One projection; returns the device tensor of per-layer ||W-Z||^2 (float64)."""
        self.h.check(self.h.lib.tadmm_plan_run(self._plan, int(update_u), int(use_u), self.resid_sq.data_ptr(),
                                               _stream(self.device)))
        return self.resid_sq

    def set_jacobi(self, tol=0.0, inner_sweeps=0, max_sweeps=0):
        self.h.check(self.h.lib.tadmm_plan_set_jacobi(self._plan, float(tol), int(inner_sweeps), int(max_sweeps)))

    def enable_timing(self, on=True):
        self.h.check(self.h.lib.tadmm_plan_enable_timing(self._plan, int(on)))

    def last_timing(self):
        out = (C.c_double * 8)()
        self.h.check(self.h.lib.tadmm_plan_last_timing(self._plan, out))
        keys = ["unfold_ms", "gram_ms", "eig_ms", "project_ms", "reconstruct_ms", "fold_update_ms", "jacobi_sweeps"]
        return {k: float(out[i]) for i, k in enumerate(keys)}

    def lanes(self) -> List[int]:
        """Lane of every layer: all 0 for a one-lane plan; long chains 0 / the rest 1 when the plan runs as two
        concurrent sub-plans (include/tadmm.h, tadmm_plan_lanes)."""
        out = (C.c_int32 * self.n)()
        self.h.lib.tadmm_plan_lanes(self._plan, out)
        return [int(x) for x in out]

    def filter_stats(self) -> dict:
        """Filtered eigen-solver counters: eligible problems of the plan; of the last run: solves, fallbacks, stages."""
        out = (C.c_int32 * 4)()
        self.h.check(self.h.lib.tadmm_plan_filter_stats(self._plan, out))
        return dict(eligible=int(out[0]), solves=int(out[1]), fallbacks=int(out[2]), stages=int(out[3]))

    def filter_timing(self) -> dict:
        """Instrumented runs: summed duration, count and executed FLOPs of the filter's fp64 GEMM launches."""
        out = (C.c_double * 4)()
        self.h.check(self.h.lib.tadmm_plan_filter_timing(self._plan, out))
        return dict(gemm_ms=float(out[0]), gemm_launches=int(out[1]), gemm_flops=float(out[2]))

    def filter_timing_fast(self) -> dict:
        """Instrumented runs: the filter products that ran at fp32 accuracy on the bf16 matrix cores (dgemm3_kernel)."""
        out = (C.c_double * 4)()
        self.h.check(self.h.lib.tadmm_plan_filter_timing_fast(self._plan, out))
        return dict(ms=float(out[0]), launches=int(out[1]), flops=float(out[2]))

    def jacobi_timing(self) -> dict:
        """Instrumented runs: summed duration, count, executed matrix-core FLOPs and workgroups of the jacobi_tick3 launches."""
        out = (C.c_double * 4)()
        self.h.check(self.h.lib.tadmm_plan_jacobi_timing(self._plan, out))
        return dict(tick_ms=float(out[0]), tick_launches=int(out[1]), tick_flops=float(out[2]), tick_wgs=float(out[3]))

    def singular_values(self, layer: int, step: int) -> np.ndarray:
        r = self.ranks[layer][step + 1]
        out = (C.c_double * r)()
        self.h.check(self.h.lib.tadmm_plan_singular_values(self._plan, layer, step, out, _stream(self.device)))
        return np.array(out[:], dtype=np.float64)

    def core_tensors(self, layer: int) -> List[torch.Tensor]:
        """Views (r_j, n_j, r_{j+1}) into the core buffer of a layer (want_cores=True)."""
        buf = self.cores[layer]
        if buf is None:
            raise TadmmError(-1, "plan was created without cores")
        out, off = [], 0
        for a, b, c in self._core_shapes[layer]:
            out.append(buf[off:off + a * b * c].view(a, b, c))
            off += a * b * c
        return out

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan:
            self.h.lib.tadmm_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ grouped GEMM
def gemm_desc(A, B, Cout, M, N, K, a_strides, b_strides, c_strides, alpha=1.0, beta=0.0, bias_n=None, bias_m=None):
    g = GemmDesc()
    g.A, g.B, g.C = A, B, Cout
    g.M, g.N, g.K = M, N, K
    g.a_rs, g.a_cs = a_strides
    g.b_rs, g.b_cs = b_strides
    g.c_rs, g.c_cs = c_strides
    g.alpha, g.beta = alpha, beta
    g.bias_n = bias_n
    g.bias_m = bias_m
    return g


class TuckerPlan:
    """Grouped Tucker-2 projection of a set of layers (the 'tk' branches of ADMM.update, admm.py:47-50, :59-62).

    layers: sequence of dicts with keys W, U, Z (float32 device tensors, 4-D (O,I,kh,kw) or 2-D (out,in)) and
    ranks = [r_out, r_in].  HOSVD + HOOI for all layers in lock-step on the device (csrc/tucker_plan.hip).
    """

    def __init__(self, layers: Sequence[dict], n_iter_max: int = 100, tol: float = 1e-4):
        if not layers:
            raise ValueError("empty plan")
        dev = layers[0]["W"].device
        self.device = dev
        self.h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
        lib = self.h.lib
        n = len(layers)
        self.n = n
        self._descs = (LayerDesc * n)()
        self._keep = []
        self._shapes = []
        Wp = (C.c_void_p * n)()
        Up = (C.c_void_p * n)()
        Zp = (C.c_void_p * n)()
        for i, L in enumerate(layers):
            for key in ("W", "U", "Z"):
                _require_cuda(L[key], key)
                if L[key].shape != L["W"].shape:
                    raise TadmmError(-1, f"{key} shape differs from W")
            shape = list(L["W"].shape)
            if len(shape) not in (2, 4):
                raise TadmmError(-1, "Tucker layers are 2-D or 4-D")
            self._descs[i] = make_layer_desc(KIND_TUCKER2, shape, None, L["ranks"], 0, n_iter_max, tol)
            Wp[i], Up[i], Zp[i] = L["W"].data_ptr(), L["U"].data_ptr(), L["Z"].data_ptr()
            self._keep.append((L["W"], L["U"], L["Z"]))
            self._shapes.append((shape, int(L["ranks"][0]), int(L["ranks"][1])))
        size = C.c_size_t()
        self.h.check(lib.tadmm_tucker_workspace_bytes(self.h.ptr, n, self._descs, C.byref(size)))
        self.workspace_bytes = int(size.value)
        self.workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=dev)
        self.resid_sq = torch.zeros(n, dtype=torch.float64, device=dev)
        self._plan = C.c_void_p()
        self.h.check(lib.tadmm_tucker_create(self.h.ptr, n, self._descs, Wp, Up, Zp, self.workspace.data_ptr(),
                                             self.workspace_bytes, C.byref(self._plan)))

    def run(self, update_u: bool = True, use_u: bool = True) -> torch.Tensor:
        self.h.check(self.h.lib.tadmm_tucker_run(self._plan, int(update_u), int(use_u), self.resid_sq.data_ptr(),
                                                 _stream(self.device)))
        return self.resid_sq

    def _view(self, ptr: int, numel: int) -> torch.Tensor:
        off = ptr - self.workspace.data_ptr()
        return self.workspace[off:off + 4 * numel].view(torch.float32)

    def factors(self, layer: int):
        """(core, U_out, U_in) of the last run as tensors shaped like tensorly's partial_tucker output:
        core (r_out, r_in, kh, kw) | (r_out, r_in), U_out (O, r_out), U_in (I, r_in).  Copies."""
        shape, ro, ri = self._shapes[layer]
        c, a, b = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.h.check(self.h.lib.tadmm_tucker_factors(self._plan, layer, C.byref(c), C.byref(a), C.byref(b)))
        k2 = shape[2] * shape[3] if len(shape) == 4 else 1
        core = self._view(c.value, ro * k2 * ri).view(ro, k2, ri).permute(0, 2, 1).contiguous()
        core = core.view(ro, ri, shape[2], shape[3]) if len(shape) == 4 else core.view(ro, ri)
        u_out = self._view(a.value, shape[0] * ro).view(shape[0], ro).clone()
        u_in = self._view(b.value, shape[1] * ri).view(shape[1], ri).clone()
        return core, u_out, u_in

    def iterations(self):
        """(HOOI sweeps per layer, final relative reconstruction error per layer) of the last run."""
        it = (C.c_int32 * self.n)()
        err = (C.c_double * self.n)()
        self.h.check(self.h.lib.tadmm_tucker_iterations(self._plan, it, err, _stream(self.device)))
        return list(it), list(err)

    def jacobi_sweeps(self) -> int:
        """Jacobi sweeps summed over all eigen-solve groups of the last run."""
        return int(self.h.lib.tadmm_tucker_jacobi_sweeps(self._plan))

    def enable_timing(self, on: bool = True):
        self.h.check(self.h.lib.tadmm_tucker_enable_timing(self._plan, int(on)))

    def last_timing(self) -> dict:
        """Instrumented run: eigen-solver launches timed one by one with HIP events on the launch stream."""
        out = (C.c_double * 8)()
        self.h.check(self.h.lib.tadmm_tucker_last_timing(self._plan, out))
        return dict(eig_ms=out[0], eig_launches=int(out[1]), eig_model_flops=out[2], total_ms=out[3], hooi_sweeps=int(out[4]))

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan:
            self.h.lib.tadmm_tucker_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GemmBatch:
    """A packed group of strided GEMMs; `run()` is one kernel launch."""

    def __init__(self, descs: Sequence[GemmDesc], device):
        self.device = device
        self.h = Handle.get(device.index if device.index is not None else torch.cuda.current_device())
        lib = self.h.lib
        n = len(descs)
        arr = (GemmDesc * n)(*descs)
        nbytes = lib.tadmm_gemm_pack_bytes(n, arr)
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else \
            torch.empty(nbytes, dtype=torch.uint8)
        nblocks = C.c_int()
        rc = lib.tadmm_gemm_pack(n, arr, host.data_ptr(), nbytes, C.byref(nblocks))
        if rc < 0:
            raise TadmmError(rc, "tadmm_gemm_pack: invalid GEMM descriptor (each operand needs one unit stride)")
        self.blob = host.to(device, non_blocking=True)
        self._host = host
        self.n, self.nblocks = n, int(nblocks.value)

    def run(self):
        self.h.check(self.h.lib.tadmm_gemm_run(self.h.ptr, self.blob.data_ptr(), self.n, self.nblocks,
                                               _stream(self.device)))


def mm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, alpha=1.0, bias_n=None, bias_m=None):
    """out = alpha * a @ b (+bias) for 2-D float32 device tensors with arbitrary (one unit) strides."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0]
    M, K = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    d = gemm_desc(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(), b.stride(), out.stride(), alpha, 0.0,
                  None if bias_n is None else bias_n.data_ptr(), None if bias_m is None else bias_m.data_ptr())
    dev = a.device
    h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
    h.check(h.lib.tadmm_gemm(h.ptr, C.byref(d), _stream(dev)))       # descriptor by value: one launch, no upload
    return out


def mm_nt_bf16(a: torch.Tensor, bt: torch.Tensor, bias_n: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a (M,K) @ bt (N,K)^T [+ bias over N] in bf16 with fp32 accumulation; rows contiguous along K."""
    assert a.dim() == 2 and bt.dim() == 2 and a.shape[1] == bt.shape[1]
    for t, what in ((a, "a"), (bt, "bt")):
        if not t.is_cuda:
            raise TadmmError(-1, f"{what} must live on a HIP device; there is no CPU path")
        if t.dtype != torch.bfloat16:
            raise TadmmError(-1, f"{what} must be bfloat16")
        if t.stride(1) != 1:
            raise TadmmError(-1, f"{what} must be contiguous along K")
    M, K = a.shape
    N = bt.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    if bias_n is not None:
        bias_n = bias_n.float().contiguous()
    dev = a.device
    h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
    h.check(h.lib.tadmm_gemm_bf16_nt(h.ptr, a.data_ptr(), bt.data_ptr(), out.data_ptr(), M, N, K, a.stride(0),
                                     bt.stride(0), N, None if bias_n is None else bias_n.data_ptr(), _stream(dev)))
    return out


# ------------------------------------------------------------------ forward chains (csrc/chain.hip)
def weight_planes(w: torch.Tensor, planes: int, pad_rows: int = 16, pad_cols: int = 32) -> torch.Tensor:
    """(N, K) weight -> (planes, Np/16, Kp/32, 64, 8) bfloat16 in the fragment-major order of csrc/chain.hip
    (Kp = K rounded up to `pad_cols`, a multiple of 32; Np = N rounded up to `pad_rows`, a multiple of 16; zero
    padded).  The fused chain wants its middle rank padded to 64: `pad_rows=64` for Win, `pad_cols=64` for Wout.
    planes == 3: the exact three-term split w = w1 + w2 + w3 of a float32 weight; planes == 1: its bf16 rounding."""
    assert w.dim() == 2 and planes in (1, 3) and pad_rows % 16 == 0 and pad_cols % 32 == 0
    if not w.is_cuda:
        raise TadmmError(-1, "weights must live on a HIP device; there is no CPU path")
    N, K = w.shape
    Np, Kp = -(-N // pad_rows) * pad_rows, -(-K // pad_cols) * pad_cols
    flat = torch.zeros(planes, Np, Kp, dtype=torch.bfloat16, device=w.device)
    r = w.detach().float()
    for p in range(planes):
        t = r.to(torch.bfloat16)
        flat[p, :N, :K] = t
        if p + 1 < planes:
            r = r - t.float()
    # (p, n/16, n%16, k/32, (k%32)/8, k%8) -> (p, n/16, k/32, (k%32)/8, n%16, k%8)
    return flat.view(planes, Np // 16, 16, Kp // 32, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(
        planes, Np // 16, Kp // 32, 64, 8)


def unpack_planes(wp: torch.Tensor) -> torch.Tensor:
    """Inverse of `weight_planes`: (planes, Np, Kp) bfloat16 row-major (tests, debugging)."""
    P, nt, ks = wp.shape[:3]
    return wp.view(P, nt, ks, 4, 16, 8).permute(0, 1, 4, 2, 3, 5).reshape(P, nt * 16, ks * 32)


class _LruMemo(collections.OrderedDict):
    """Validated descriptors of recent chain launches, keyed on geometry + weight-plane addresses.  An entry pins its
    plane tensors (that is what keeps the addresses meaningful), so the memo is a small LRU: callers whose planes are
    short-lived (the autograd Functions pack fresh planes every call) pass `memo=False` and never enter it."""
    CAP = 128

    def lookup(self, key):
        hit = self.get(key)
        if hit is not None:
            self.move_to_end(key)
        return hit

    def store(self, key, value):
        self[key] = value
        self.move_to_end(key)
        while len(self) > self.CAP:
            self.popitem(last=False)


_CHAIN_MEMO = _LruMemo()


def _chain_call(entry: str, x: torch.Tensor, win: torch.Tensor, wout, bias, kin: int, n1: int, n_out: int,
                image_out: bool, tile_tokens: int, prepare_only: bool = False, use_memo: bool = True):
    if not x.is_cuda:
        raise TadmmError(-1, "x must live on a HIP device; there is no CPU path")
    if x.dim() == 4:                                   # (B, C, H, W) read in place
        if not x.is_contiguous():
            x = x.contiguous()
    elif x.stride(1) != 1 or (x.stride(0) * x.element_size()) % 16 or x.data_ptr() % 16:
        x = x.contiguous()
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.detach().float().contiguous()
        bias_key = None                                 # a converted copy: nothing to memoise
    else:
        bias_key = 0 if bias is None else bias.data_ptr()
    if not use_memo:
        bias_key = None                                 # short-lived planes (training): build, launch, forget
    # geometry + weight identity -> validated descriptor; only the activation pointers change between calls
    key = (entry, tuple(x.shape), x.stride(0), x.dtype, x.device, win.data_ptr(), 0 if wout is None else wout.data_ptr(),
           bias_key, kin, n1, n_out, image_out, tile_tokens)
    memo = _CHAIN_MEMO.lookup(key) if bias_key is not None else None
    if memo is None:
        if x.dtype == torch.float32:
            dtype, planes = _cabi.CHAIN_F32, 3
        elif x.dtype == torch.bfloat16:
            dtype, planes = _cabi.CHAIN_BF16, 1
        else:
            raise TadmmError(-1, f"chain: unsupported dtype {x.dtype}")
        if win.dtype != torch.bfloat16 or win.dim() != 5 or win.shape[0] != planes or not win.is_contiguous():
            raise TadmmError(-1, f"chain: weights must be {planes} contiguous bf16 plane(s) (ops.weight_planes)")
        d = _cabi.ChainDesc()
        if x.dim() == 4:
            B, Cc, H, W = x.shape
            assert Cc == kin
            T, hw = B * H * W, H * W
            d.x_hw, d.ldx = hw, 0
        else:
            assert x.dim() == 2 and x.shape[1] == kin
            T, hw = x.shape[0], 0
            d.x_hw, d.ldx = 0, x.stride(0)
        feat = n_out if wout is not None else n1
        if image_out:
            assert hw > 0
            yshape = (x.shape[0], feat, x.shape[2], x.shape[3])
            d.y_hw, d.ldy = hw, 0
        else:
            yshape = (T, feat)
            d.y_hw, d.ldy = 0, feat
        d.Win = win.data_ptr()
        d.Wout = None if wout is None else wout.data_ptr()
        d.bias = None if bias is None else bias.data_ptr()
        d.T, d.Kin, d.R, d.Nout = T, kin, n1, n_out if wout is not None else 0
        if win.shape[2] != -(-kin // 32) or win.shape[1] * 16 < n1:
            raise TadmmError(-1, "chain: weight planes do not match the operand shape")
        d.win_plane = win[0].numel()
        if wout is not None:
            if wout.dtype != torch.bfloat16 or wout.dim() != 5 or wout.shape[0] != planes or not wout.is_contiguous():
                raise TadmmError(-1, f"chain: weights must be {planes} contiguous bf16 plane(s) (ops.weight_planes)")
            if wout.shape[2] * 32 != n1 or wout.shape[1] * 16 < n_out:
                raise TadmmError(-1, "chain: output weight planes do not match the middle rank / output size")
            d.wout_plane = wout[0].numel()
        d.dtype, d.tile_tokens = dtype, tile_tokens
        dev = x.device
        h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
        memo = (d, getattr(h.lib, entry), h, yshape, T, (win, wout, bias))     # the tuple keeps the weights alive
        if bias_key is not None:
            _CHAIN_MEMO.store(key, memo)
    d, fn, h, yshape, T, _ = memo
    y = torch.empty(yshape, dtype=x.dtype, device=x.device)

    def launch():
        d.X, d.Y = x.data_ptr(), y.data_ptr()
        if T > 0:
            h.check(fn(h.ptr, C.byref(d), _stream(x.device)))
        return y

    if prepare_only:
        return launch
    return launch()


def chain_fused(x, win_planes, wout_planes, bias, n_out: int, entry: str = "tadmm_ttlinear_fwd", tile_tokens: int = 0,
                prepare_only: bool = False, memo: bool = True):
    """y (T, n_out) = (x (T, Kin) @ Win^T) @ Wout^T + bias in one launch (TTLinear.py:75-93).  `win_planes` (rows
    padded to the middle rank R, a multiple of 64, <= 256) and `wout_planes` come from `weight_planes`."""
    return _chain_call(entry, x, win_planes, wout_planes, bias, x.shape[-1], win_planes.shape[1] * 16, n_out, False,
                       tile_tokens, prepare_only, memo)


def chain_single(x, w_planes, bias, n_out: int, entry: str = "tadmm_ttconv_chain_in", image_out: bool = False,
                 tile_tokens: int = 0, prepare_only: bool = False, memo: bool = True):
    """y = x @ W^T + bias for token rows (T, Kin) or, in place, a channels-first image (B, Kin, H, W) ->
    (B, n_out, H, W) when `image_out` (TTConv.py:131-137 / :141-151, TKConv.py:93-98).  `prepare_only` returns a
    zero-argument launcher over the same buffers (benchmarks: no per-call descriptor building)."""
    kin = x.shape[1]
    return _chain_call(entry, x, w_planes, None, bias, kin, n_out, 0, image_out, tile_tokens, prepare_only, memo)


def _conv_chain_plan(x: torch.Tensor, r1: int, r2: int, kernel_size, stride, padding, dilation):
    """(pixels per workgroup, output rows per workgroup, halo tiles, workgroups per image) of the one-launch factorised
    convolution, or None when it does not apply -- the rule tadmm_ttconv_fused applies."""
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        return None
    H, W = x.shape[2], x.shape[3]
    ho = (H + 2 * padding[0] - dilation[0] * (kernel_size[0] - 1) - 1) // stride[0] + 1
    wo = (W + 2 * padding[1] - dilation[1] * (kernel_size[1] - 1) - 1) // stride[1] + 1
    if ho <= 0 or wo <= 0 or wo > 64:
        return None
    r1p, r2p = -(-r1 // 32) * 32, -(-r2 // 32) * 32
    if r1p > 256 or r2p > 256:
        return None
    planes, kc = (3, 64) if x.dtype == torch.float32 else (1, 128)
    for tm in (64, 32):                                  # pixels per workgroup: 64, or 32 when 64 does not fit the LDS
        if wo > tm:
            continue
        tr, nt = min(ho, tm // wo), 0
        while tr >= 1:
            irows = min(H, (tr - 1) * stride[0] + (kernel_size[0] - 1) * dilation[0] + 1)
            nt = -(-(irows * W) // tm)
            if nt <= 3:
                break
            tr -= 1
        if tr < 1:
            continue
        lds = (2 * planes * tm * (kc + 8) + planes * tm * nt * (r1p + 8) + planes * tm * (r2p + 8)) * 2
        if lds <= 160 * 1024:
            return tm, tr, nt, -(-ho // tr)
    return None


def conv_chain_fits(x: torch.Tensor, r1: int, r2: int, kernel_size, stride, padding, dilation) -> bool:
    """True when the one-launch factorised convolution (csrc/convchain.hip) applies: output rows of at most 64 pixels,
    a tile of output rows whose halo is at most three pixel tiles, ranks at most 256 and both intermediates inside the
    160 KiB of LDS."""
    return _conv_chain_plan(x, r1, r2, kernel_size, stride, padding, dilation) is not None


def conv_chain_pays(x: torch.Tensor, r1: int, r2: int, kernel_size, stride, padding, dilation) -> bool:
    """... and is the faster path (measured, scripts/bench_conv_layers.py): always in bf16; in fp32 (three planes, six
    MFMA products per product) only while an image is at most two workgroups -- with more row tiles the recomputed halos
    cost more than the two launches and the round trip of the intermediates they save."""
    plan = _conv_chain_plan(x, r1, r2, kernel_size, stride, padding, dilation)
    return plan is not None and (x.dtype == torch.bfloat16 or plan[3] <= 2)


def conv_core_planes(core: torch.Tensor, planes: int) -> torch.Tensor:
    """(r2, r1, kh, kw) core kernel -> fragment-major planes of the (r2 x kh*kw*r1p) tap-major matrix convchain.hip
    multiplies with (r1p = r1 rounded up to 32, rows rounded up to 32)."""
    r2, r1, kh, kw = core.shape
    r1p = -(-r1 // 32) * 32
    m = torch.zeros(r2, kh * kw, r1p, dtype=torch.float32, device=core.device)
    m[:, :, :r1] = core.detach().float().permute(0, 2, 3, 1).reshape(r2, kh * kw, r1)
    return weight_planes(m.reshape(r2, kh * kw * r1p), planes, pad_rows=32)


def conv_chain(x: torch.Tensor, w1p: torch.Tensor, w2p: torch.Tensor, w3p: torch.Tensor, bias, n_out: int, kernel_size,
               stride, padding, dilation) -> torch.Tensor:
    """y (B, n_out, Ho, Wo) = W3 conv_kxk(W1 x; Wc) + bias for NCHW images in one launch (`tadmm_ttconv_fused`; see
    `conv_chain_fits` for what is eligible).
    w1p = weight_planes(W1, P, pad_rows=32), w2p = conv_core_planes(core, P), w3p = weight_planes(W3, P)."""
    if not x.is_cuda:
        raise TadmmError(-1, "x must live on a HIP device; there is no CPU path")
    if not x.is_contiguous():
        x = x.contiguous()
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.detach().float().contiguous()
        bias_key = None
    else:
        bias_key = 0 if bias is None else bias.data_ptr()
    key = ("conv", tuple(x.shape), x.dtype, x.device, w1p.data_ptr(), w2p.data_ptr(), w3p.data_ptr(), bias_key, n_out,
           tuple(kernel_size), tuple(stride), tuple(padding), tuple(dilation))
    memo = _CHAIN_MEMO.lookup(key) if bias_key is not None else None
    if memo is None:
        B, Cc, H, W = x.shape
        d = _cabi.ConvChainDesc()
        d.dtype = _cabi.CHAIN_F32 if x.dtype == torch.float32 else _cabi.CHAIN_BF16
        ho = (H + 2 * padding[0] - dilation[0] * (kernel_size[0] - 1) - 1) // stride[0] + 1
        wo = (W + 2 * padding[1] - dilation[1] * (kernel_size[1] - 1) - 1) // stride[1] + 1
        d.W1, d.W2, d.W3 = w1p.data_ptr(), w2p.data_ptr(), w3p.data_ptr()
        d.bias = None if bias is None else bias.data_ptr()
        d.w1_plane, d.w2_plane, d.w3_plane = w1p[0].numel(), w2p[0].numel(), w3p[0].numel()
        d.B, d.C, d.R1, d.R2, d.Nout = B, Cc, w1p.shape[1] * 16, w2p.shape[1] * 16, n_out
        d.H, d.W, d.Ho, d.Wo, d.kh, d.kw = H, W, ho, wo, kernel_size[0], kernel_size[1]
        d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w = (stride[0], stride[1], padding[0], padding[1],
                                                                      dilation[0], dilation[1])
        if (w2p.shape[2] * 32 != kernel_size[0] * kernel_size[1] * d.R1 or w3p.shape[2] * 32 != d.R2
                or w1p.shape[2] != -(-Cc // 32)):
            raise TadmmError(-1, "conv chain: weight planes do not match each other")
        dev = x.device
        h = Handle.get(dev.index if dev.index is not None else torch.cuda.current_device())
        memo = (d, h.lib.tadmm_ttconv_fused, h, (B, n_out, ho, wo), (w1p, w2p, w3p, bias))
        if bias_key is not None:
            _CHAIN_MEMO.store(key, memo)
    d, fn, h, yshape, _ = memo
    y = torch.empty(yshape, dtype=x.dtype, device=x.device)
    d.X, d.Y = x.data_ptr(), y.data_ptr()
    if yshape[0] > 0:
        h.check(fn(h.ptr, C.byref(d), _stream(x.device)))
    return y


# ------------------------------------------------------------------ Gram / eigh (tests, Tucker)
def gram(a: torch.Tensor):
    """fp64 Gram of a float32 (m,n) matrix: A A^T if m<=n else A^T A.  Returns (N,N) float64."""
    _require_cuda(a, "A")
    h = Handle.get(a.device.index)
    lib = h.lib
    m, n = a.shape
    npad, ld = C.c_int(), C.c_int()
    N = lib.tadmm_gram_ld(m, n, C.byref(npad), C.byref(ld))
    G = torch.empty(npad.value, ld.value, dtype=torch.float64, device=a.device)
    sb = lib.tadmm_gram_scratch_bytes(m, n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device)
    h.check(lib.tadmm_gram_f64(h.ptr, a.data_ptr(), m, n, G.data_ptr(), ld.value, scratch.data_ptr(), sb,
                               _stream(a.device)))
    return G[:N, :N]


def eigh(G: torch.Tensor):
    """Symmetric PSD eigen-decomposition (descending).  Returns (evals (N,), evecs (N,N) rows, sweeps)."""
    assert G.dtype == torch.float64 and G.is_cuda and G.dim() == 2 and G.shape[0] == G.shape[1]
    G = G.contiguous()
    h = Handle.get(G.device.index)
    lib = h.lib
    N = G.shape[0]
    ev = torch.empty(N, dtype=torch.float64, device=G.device)
    vec = torch.empty(N, N, dtype=torch.float64, device=G.device)
    sb = lib.tadmm_eigh_scratch_bytes(N)
    scratch = torch.empty(sb, dtype=torch.uint8, device=G.device)
    sweeps = C.c_int()
    h.check(lib.tadmm_eigh_f64(h.ptr, G.data_ptr(), N, ev.data_ptr(), vec.data_ptr(), scratch.data_ptr(), sb,
                               C.byref(sweeps), _stream(G.device)))
    return ev, vec, int(sweeps.value)


# ------------------------------------------------------------------ filtered eigen-solver building blocks (tests)
def dgemm(a: torch.Tensor, b: torch.Tensor, b_transposed: bool = True) -> torch.Tensor:
    """fp64 matrix-core GEMM: a (M,K) @ b^T with b (N,K) (b_transposed) or a @ b with b (K,N); row-major float64."""
    assert a.dtype == torch.float64 and b.dtype == torch.float64 and a.is_cuda and b.is_cuda
    a, b = a.contiguous(), b.contiguous()
    M, K = a.shape
    N = b.shape[0] if b_transposed else b.shape[1]
    out = torch.empty(M, N, dtype=torch.float64, device=a.device)
    h = Handle.get(a.device.index)
    sb = h.lib.tadmm_dgemm_scratch_bytes(M, N)
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device)
    h.check(h.lib.tadmm_dgemm_f64(h.ptr, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0),
                                  N, int(b_transposed), scratch.data_ptr(), sb, _stream(a.device)))
    return out


def dgemm3(a: torch.Tensor, g: torch.Tensor, repeats: int = 1) -> torch.Tensor:
    """a (M, N) @ g (N, N)^T at fp32 accuracy on the bf16 matrix cores (float64 in / out): the product the early
    stages of the filtered eigen-solver use (csrc/dgemm3.hip)."""
    assert a.dtype == torch.float64 and g.dtype == torch.float64 and a.is_cuda and g.is_cuda
    assert a.is_contiguous() and g.is_contiguous() and g.shape[0] == g.shape[1] == a.shape[1]
    M, N = a.shape
    h = Handle.get(a.device.index)
    out = torch.empty(M, N, dtype=torch.float64, device=a.device)
    sb = h.lib.tadmm_dgemm3_scratch_bytes(M, N)
    scratch = torch.empty(sb, dtype=torch.uint8, device=a.device)
    h.check(h.lib.tadmm_dgemm3_f64(h.ptr, a.data_ptr(), g.data_ptr(), out.data_ptr(), M, N, N, N, N, repeats,
                                   scratch.data_ptr(), sb, _stream(a.device)))
    return out


def cholqr_(yt: torch.Tensor) -> bool:
    """In-place Cholesky QR of the block whose columns are the ROWS of yt (n, ncols) float64.  Returns False when a
    pivot broke down (numerically rank-deficient block)."""
    assert yt.dtype == torch.float64 and yt.is_cuda and yt.is_contiguous()
    n, ncols = yt.shape
    h = Handle.get(yt.device.index)
    sb = h.lib.tadmm_cholqr_scratch_bytes(n, ncols)
    scratch = torch.empty(sb, dtype=torch.uint8, device=yt.device)
    bad = C.c_int(0)
    h.check(h.lib.tadmm_cholqr_f64(h.ptr, yt.data_ptr(), n, ncols, yt.stride(0), scratch.data_ptr(), sb, C.byref(bad),
                                   _stream(yt.device)))
    return bad.value == 0
