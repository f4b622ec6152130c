"""Differentiable building blocks on top of the HIP grouped GEMM.

`mm(a, b)` is the one contraction primitive of the factorised layers: forward and both gradients run
on the fp32 matrix cores through `tadmm_gemm_run`, with transposes expressed as operand strides (no
materialised `.t()`), replacing the `torch.mm` / `F.linear` calls of TTLinear.py:79-86,
TTConv.py:133-147, TKConv.py:210-214 and TKLinear.py:66-71.
"""
from __future__ import annotations

import torch

from . import ops
from ._cabi import TadmmError


def _as_gemm_operand(t: torch.Tensor) -> torch.Tensor:
    """Return a 2-D float32 device view with one unit stride (copying only if there is none)."""
    if t.dim() != 2:
        raise ValueError("mm expects 2-D operands")
    if not t.is_cuda:
        raise TadmmError(-1, "mm operands must live on the HIP device (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    if 1 not in t.stride():
        t = t.contiguous()
    return t


class _Mm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias_n):
        a_, b_ = _as_gemm_operand(a), _as_gemm_operand(b)
        ctx.save_for_backward(a_, b_)
        ctx.has_bias = bias_n is not None
        return ops.mm(a_, b_, bias_n=bias_n)

    @staticmethod
    def backward(ctx, g):
        a_, b_ = ctx.saved_tensors
        g_ = _as_gemm_operand(g)
        ga = gb = gbias = None
        if ctx.needs_input_grad[0]:
            ga = ops.mm(g_, b_.t())            # dA = dC B^T   (B^T is a stride swap)
        if ctx.needs_input_grad[1]:
            gb = ops.mm(a_.t(), g_)            # dB = A^T dC
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gbias = g.sum(0)
        return ga, gb, gbias


def mm(a: torch.Tensor, b: torch.Tensor, bias_n: torch.Tensor = None) -> torch.Tensor:
    """(M,K) @ (K,N) [+ bias over N] on the MI355X matrix cores; differentiable."""
    return _Mm.apply(a, b, bias_n)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    """F.linear semantics: x (..., in) @ weight(out, in)^T + bias."""
    lead = x.shape[:-1]
    y = mm(x.reshape(-1, x.shape[-1]), weight.t(), bias)
    return y.reshape(*lead, weight.shape[0])


# ---------------------------------------------------------------------------------------------------------------
# Forward chains on the bf16 matrix cores (csrc/chain.hip): one launch per chain, the per-token intermediate of the
# fused TT-linear chain stays in LDS.  float32 tensors go through the exact three-plane bf16 split (fp32-GEMM
# accuracy), bfloat16 tensors through one plane.
# ---------------------------------------------------------------------------------------------------------------
def planes_of(w: torch.Tensor, nplanes: int, pad_rows: int = 16, pad_cols: int = 32, transpose: bool = False,
              cache: dict = None, tag=None):
    """Fragment-major bf16 planes of a 2-D weight (`ops.weight_planes`).  With `cache` (a dict owned by the layer) the
    packing is reused until the weight's version counter or storage address changes -- inference packs a layer once
    (`param_key` for what that cannot see)."""
    key = None
    if cache is not None:
        key = (tag, (w._version, w.data_ptr()), tuple(w.shape), nplanes, pad_rows, pad_cols, transpose)
        hit = cache.get(key)
        if hit is not None:
            return hit
    src = w.detach()
    wp = ops.weight_planes(src.t() if transpose else src, nplanes, pad_rows, pad_cols)
    if cache is not None:
        for k in [k for k in cache if k[0] == tag and k[3:] == key[3:]]:
            del cache[k]                                            # older versions of the same weight
        cache[key] = wp
    return wp


def param_key(*params):
    """Identity of a set of parameters for the inference caches: version counter AND storage address.  `p.data = t`
    re-points the storage without touching the counter; an in-place write THROUGH `.data` (`p.data.copy_(t)`,
    `p.data.mul_()`) changes neither -- nothing observable from the outside does -- so the layers also drop their
    caches on `train()` / `eval()`, `_apply` (`.to()`, `.half()`) and `load_state_dict`, and expose
    `invalidate_caches()` for callers that write through `.data` between two inference calls."""
    return tuple((p._version, p.data_ptr()) for p in params if p is not None)


class InferenceCacheMixin:
    """Inference-only caches of the factorised layers (contracted factors, packed bf16 planes): see `param_key`."""
    _CACHE_ATTRS = ("_chain_cache", "_plane_cache", "_fused_cache")

    def invalidate_caches(self):
        for a in self._CACHE_ATTRS:
            self.__dict__.pop(a, None)

    def train(self, mode: bool = True):
        self.invalidate_caches()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_caches()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_caches()
        return super()._load_from_state_dict(*args, **kwargs)


def _nplanes(x: torch.Tensor) -> int:
    if x.dtype == torch.float32:
        return 3
    if x.dtype == torch.bfloat16:
        return 1
    raise TadmmError(-1, f"chain kernels take float32 or bfloat16 activations (got {x.dtype})")


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def fused_rank_ok(r: int) -> bool:
    """The fused chain keeps a token's middle-rank vector in LDS: ranks up to 256 (padded to a multiple of 64)."""
    return 0 < r <= 256


class _ChainSingle(torch.autograd.Function):
    """y = x W^T + bias on token rows (T, K) or, in place, on channels-first images (B, K, H, W) -> (B, N, H, W)."""

    @staticmethod
    def forward(ctx, x, w, bias, entry, wp):
        if not x.is_cuda:
            raise TadmmError(-1, "chain operands must live on the HIP device (no CPU fallback)")
        image = x.dim() == 4
        fresh = wp is None                          # planes packed for this call only: keep them out of the launch memo
        if fresh:
            wp = planes_of(w, _nplanes(x))
        y = ops.chain_single(x, wp, bias, w.shape[0], entry=entry, image_out=image, memo=not fresh)
        ctx.save_for_backward(x, w)
        ctx.entry, ctx.has_bias, ctx.image = entry, bias is not None, image
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:                 # dX = dY W : the same kernel with the transposed weight
            gx = ops.chain_single(g, planes_of(w, _nplanes(g), transpose=True), None, w.shape[1], entry=ctx.entry,
                                  image_out=ctx.image, memo=False)
        if ctx.needs_input_grad[1]:                 # dW = dY^T X  (N x K), plain fp32 product
            if ctx.image:
                g2 = g.permute(1, 0, 2, 3).reshape(g.shape[1], -1)
                x2 = x.permute(1, 0, 2, 3).reshape(x.shape[1], -1)
                gw = ops.mm(_as_gemm_operand(g2.float()), _as_gemm_operand(x2.float()).t())
            else:
                gw = ops.mm(_as_gemm_operand(g.float()).t(), _as_gemm_operand(x.float()))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum((0, 2, 3)) if ctx.image else g.sum(0)
        return gx, gw, gb, None, None


def pointwise(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None, entry: str = "tadmm_tucker_1x1",
              planes: torch.Tensor = None):
    """Per-token / per-pixel linear map `x W^T + bias` with W (N, K): rows (..., K) or an image (B, K, H, W) in place
    (the 1x1 convolutions of TKConv.py:93-98 and the core chains of TTConv.py:131-151).  Differentiable.  `planes`:
    prebuilt `planes_of(w, ...)` (inference caches)."""
    if not _needs_grad(x, w, bias):                   # inference: straight to the C ABI, no autograd node
        fresh = planes is None
        if fresh:
            planes = planes_of(w, _nplanes(x))
        if x.dim() == 4:
            return ops.chain_single(x, planes, bias, w.shape[0], entry=entry, image_out=True, memo=not fresh)
        lead = x.shape[:-1]
        return ops.chain_single(x.reshape(-1, x.shape[-1]), planes, bias, w.shape[0], entry=entry,
                                memo=not fresh).reshape(*lead, w.shape[0])
    if x.dim() == 4:
        return _ChainSingle.apply(x.contiguous(), w, bias, entry, planes)
    lead = x.shape[:-1]
    return _ChainSingle.apply(x.reshape(-1, x.shape[-1]), w, bias, entry, planes).reshape(*lead, w.shape[0])


class _ChainFused(torch.autograd.Function):
    """y = (x Win^T) Wout^T + bias in one launch; Win (R, K), Wout (N, R), R <= 256."""

    @staticmethod
    def forward(ctx, x, w_in, w_out, bias, planes):
        if not x.is_cuda:
            raise TadmmError(-1, "chain operands must live on the HIP device (no CPU fallback)")
        n = _nplanes(x)
        fresh = planes is None
        if fresh:
            planes = (planes_of(w_in, n, pad_rows=64), planes_of(w_out, n, pad_cols=64))
        y = ops.chain_fused(x, planes[0], planes[1], bias, w_out.shape[0], memo=not fresh)
        ctx.save_for_backward(x, w_in, w_out)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w_in, w_out = ctx.saved_tensors
        g = g.contiguous()
        n = _nplanes(g)
        gx = gwi = gwo = gb = None
        if ctx.needs_input_grad[0]:                 # dX = (dY Wout) Win: the fused kernel on the transposed factors
            gx = ops.chain_fused(g, planes_of(w_out, n, pad_rows=64, transpose=True),
                                 planes_of(w_in, n, pad_cols=64, transpose=True), None, w_in.shape[1],
                                 entry="tadmm_ttlinear_bwd", memo=False)
        if ctx.needs_input_grad[1]:                 # dWin = (dY Wout)^T X
            gr = ops.chain_single(g, planes_of(w_out, n, transpose=True), None, w_out.shape[1], memo=False)
            gwi = ops.mm(_as_gemm_operand(gr.float()).t(), _as_gemm_operand(x.float()))
        if ctx.needs_input_grad[2]:                 # dWout = dY^T (X Win^T)
            h = ops.chain_single(x, planes_of(w_in, _nplanes(x)), None, w_in.shape[0], memo=False)
            gwo = ops.mm(_as_gemm_operand(g.float()).t(), _as_gemm_operand(h.float()))
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = g.sum(0)
        return gx, gwi, gwo, gb, None


def linear_chain(x: torch.Tensor, w_in: torch.Tensor, w_out: torch.Tensor, bias: torch.Tensor = None, planes=None):
    """(..., K) -> (..., N): x Win^T Wout^T + bias through the fused chain kernel (TTLinear.py:75-93 with the input
    cores contracted into Win and the output cores into Wout).  Differentiable.  `planes`: prebuilt
    (planes_of(w_in, n, pad_rows=64), planes_of(w_out, n, pad_cols=64))."""
    lead = x.shape[:-1]
    if not _needs_grad(x, w_in, w_out, bias):         # inference: straight to the C ABI, no autograd node
        fresh = planes is None
        if fresh:
            n = _nplanes(x)
            planes = (planes_of(w_in, n, pad_rows=64), planes_of(w_out, n, pad_cols=64))
        return ops.chain_fused(x.reshape(-1, x.shape[-1]), planes[0], planes[1], bias, w_out.shape[0],
                               memo=not fresh).reshape(*lead, w_out.shape[0])
    return _ChainFused.apply(x.reshape(-1, x.shape[-1]), w_in, w_out, bias, planes).reshape(*lead, w_out.shape[0])
