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
