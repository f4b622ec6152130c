"""Tensor-train factorised layers with the reference's constructor signatures and state_dict keys
(TTConv.py:23-333, TTLinear.py:23-160) so `*_model.pt` checkpoints interchange.

  TTConv2dM : in-core GEMM chain -> small k x k conv -> out-core GEMM chain     (keys in_tt_cores.i,
              core_kernel, out_tt_cores.i, bias)
  TTConv2dR : rebuild the dense kernel from the cores each forward + conv2d     (out_tt_cores.i,
              conv_core, in_tt_cores.i, bias)
  TTLinearM : GEMM chain                                                        (tt_cores.i, bias)
  TTLinearR : rebuild the dense weight + linear                                 (tt_cores.i, bias)

Every contraction runs on the fp32 matrix cores through `functional.mm`; the one-shot decomposition of
`dense_w` (the --decompose hand-off) runs the device TT-SVD of libtadmm_hip.so.  The k x k core conv
is the stock conv2d of the device library, exactly as in the reference.

Deliberate deviation: the reference's TTConv2dM adds its (O,) bias to a (B,O,H,W) tensor
(TTConv.py:150-151), i.e. broadcasts it along W and raises unless W == O.  Here the bias is added per
output channel.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn
from torch.nn import init
from torch.nn.modules.utils import _pair, _reverse_repeat_tuple

from . import functional as HF
from . import ops
from . import ttd
from ._cabi import KIND_TT_CONV, KIND_TT_LINEAR


def _split_modes(tt_shapes, out_dim, has_kernel_mode):
    """Leading modes whose product reaches `out_dim` are output modes (TTConv.py:49-59, TTLinear.py:31-40)."""
    prod = 1
    n_out = None
    for i, n in enumerate(tt_shapes):
        prod *= n
        if prod == out_dim:
            n_out = i + 1
            break
    if n_out is None:
        raise AssertionError("tt_shapes do not factor the output dimension")
    n_in = len(tt_shapes) - n_out - (1 if has_kernel_mode else 0)
    return n_out, n_in


def _empty(*shape):
    return nn.Parameter(torch.empty(*shape))


def _zero_bias(n: int):
    """Deliberate deviation: the reference allocates the bias with `torch.Tensor(n)` (TTLinear.py:53, TTConv.py:254) and
    the M variants (and every layer built from `dense_w` without `dense_b`) never initialise it -- it holds whatever the
    allocator hands out, NaNs included.  Zeros are one of the values that memory can hold; `reset_parameters` of the R
    variants overwrites them as the reference does."""
    return nn.Parameter(torch.zeros(n))


def _decompose(dense_w: Tensor, tt_shapes, tt_ranks, kind):
    """Device TT-SVD of a dense weight (TTConv.py:96-100, TTLinear.py:61-63).  Clamps `tt_ranks` in place
    like ttd.ten2tt does."""
    dev = dense_w.device if dense_w.is_cuda else torch.device("cuda", torch.cuda.current_device())
    cores, ranks, _, _ = ttd.tt_decompose(dense_w.detach().to(dev), tt_shapes, tt_ranks, kind)
    tt_ranks[:] = ranks
    return cores


class _TTConvBase(HF.InferenceCacheMixin, nn.Module):
    def _setup(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
               hp_dict, name):
        self.tt_shapes = list(hp_dict.tt_shapes[name])
        self.tt_order = len(self.tt_shapes)
        self.out_tt_order, self.in_tt_order = _split_modes(self.tt_shapes, out_channels, True)
        self.out_tt_shapes = self.tt_shapes[:self.out_tt_order]
        self.in_tt_shapes = self.tt_shapes[self.out_tt_order + 1:]
        assert in_channels == int(np.prod(self.in_tt_shapes))
        assert out_channels == int(np.prod(self.out_tt_shapes))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.tt_ranks = list(hp_dict.ranks[name])
        # sliced before any clamp, like the reference (TTConv.py:66-68 / :233-235)
        self.out_tt_ranks = self.tt_ranks[:self.out_tt_order + 1]
        self.in_tt_ranks = self.tt_ranks[self.out_tt_order + 1:]
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.transposed = False
        self.output_padding = _pair(0)
        self.groups = groups
        self.padding_mode = padding_mode

    def get_ranks(self):
        return ', '.join(str(r) for r in self.tt_ranks)


class TTConv2dM(_TTConvBase):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = 'zeros', hp_dict=None, name: str = None,
                 dense_w: Tensor = None, dense_b: Tensor = None):
        if groups != 1:
            raise ValueError("groups must be 1 in this mode")
        if padding_mode != 'zeros':
            raise ValueError("padding_mode must be zero in this mode")
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                    hp_dict, name)
        self.in_tt_cores = nn.ParameterList(
            [_empty(self.in_tt_ranks[i], self.in_tt_shapes[i], self.in_tt_ranks[i + 1]) for i in range(self.in_tt_order)])
        self.core_kernel = _empty(self.out_tt_ranks[-1], self.in_tt_ranks[0], *self.kernel_size)
        self.out_tt_cores = nn.ParameterList(
            [_empty(self.out_tt_ranks[i], self.out_tt_shapes[i], self.out_tt_ranks[i + 1])
             for i in range(self.out_tt_order)])
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.out_channels))
            if dense_b is not None:
                self.bias.data = dense_b
        else:
            self.register_parameter('bias', None)
        if dense_w is not None:
            cores = _decompose(dense_w, self.tt_shapes, self.tt_ranks, KIND_TT_CONV)
            kq = self.out_tt_order
            for i, c in enumerate(cores):
                if i < kq:
                    self.out_tt_cores[i].data = c
                elif i == kq:     # (r, k^2, r') -> (r, r', kh, kw)   TTConv.py:105-107
                    self.core_kernel.data = c.permute(0, 2, 1).reshape(c.shape[0], c.shape[2], *self.kernel_size).contiguous()
                else:
                    self.in_tt_cores[i - kq - 1].data = c
        else:
            self.reset_parameters()

    def reset_parameters(self) -> None:
        for p in list(self.out_tt_cores) + list(self.in_tt_cores) + [self.core_kernel]:
            init.xavier_uniform_(p)

    def _chains(self, x, count_flops=False):
        flops = 0.0
        b, _, h, w = x.shape
        out = x.permute(0, 2, 3, 1)
        for i in range(self.in_tt_order - 1, -1, -1):            # TTConv.py:133-136
            k = self.in_tt_shapes[i] * self.in_tt_ranks[i + 1]
            rows = out.reshape(-1, k)
            out = HF.mm(rows, self.in_tt_cores[i].reshape(self.in_tt_ranks[i], k).t())
            flops += self.in_tt_ranks[i] * k * rows.shape[0] / 1e6
        out = out.reshape(b, h, w, self.in_tt_ranks[0]).permute(0, 3, 1, 2)
        out = F.conv2d(out, self.core_kernel, None, self.stride, self.padding, self.dilation, self.groups)
        _, _, h2, w2 = out.shape
        flops += h2 * w2 * self.core_kernel.numel() / 1e6
        out = out.permute(0, 2, 3, 1)
        for i in range(self.out_tt_order - 1, -1, -1):           # TTConv.py:143-147
            r1 = self.out_tt_ranks[i + 1]
            a = self.out_tt_cores[i].reshape(self.out_tt_ranks[i] * self.out_tt_shapes[i], r1)
            rows = out.reshape(-1, r1)
            out = HF.mm(a, rows.t())
            flops += a.shape[0] * a.shape[1] * rows.shape[0] / 1e6
            out = out.reshape(self.out_tt_ranks[i], -1).t()
        out = out.reshape(self.out_channels, b, h2, w2).permute(1, 0, 2, 3)
        if self.bias is not None:
            out = out + self.bias.view(1, -1, 1, 1)
        return out, flops, (h2, w2)

    def _factors(self):
        """Input cores contracted into (r, C), output cores into (O, r'): the two per-pixel maps around the core conv."""
        w_in = _chain_recover(list(self.in_tt_cores)).reshape(self.in_tt_ranks[0], self.in_channels)
        w_out = _chain_recover(list(self.out_tt_cores)).reshape(self.out_channels, self.out_tt_ranks[-1])
        return w_in, w_out

    def forward(self, x):
        """TTConv.py:130-153 as three launches: `tadmm_ttconv_chain_in` on the NCHW input in place (input cores
        contracted), the k x k core convolution, `tadmm_ttconv_chain_out` + bias (output cores contracted)."""
        if x.dtype not in (torch.float32, torch.bfloat16) or not self.in_tt_order or not self.out_tt_order:
            return self._chains(x)[0]
        params = list(self.in_tt_cores) + list(self.out_tt_cores)
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        n = 1 if x.dtype == torch.bfloat16 else 3
        if grad:
            w_in, w_out = self._factors()
            p_in = p_out = None
        else:
            cache = self.__dict__.setdefault("_chain_cache", {})
            key = (n, x.device, HF.param_key(*params, self.core_kernel))
            if cache.get("key") != key:
                with torch.no_grad():
                    w_in, w_out = self._factors()
                cache.update(key=key, w=(w_in, w_out), planes=(HF.planes_of(w_in, n), HF.planes_of(w_out, n)))
            (w_in, w_out), (p_in, p_out) = cache["w"], cache["planes"]
        # the one-launch path builds no autograd node: only when NOTHING it reads wants a gradient (frozen cores with a
        # trainable input / core kernel / bias keep the three differentiable launches below)
        if (not grad and not HF._needs_grad(x, self.core_kernel, self.bias) and self.groups == 1
                and ops.conv_chain_pays(x, w_in.shape[0], w_out.shape[1], self.kernel_size, self.stride, self.padding,
                                        self.dilation)):
            # small planes (<= 64 pixels): the whole layer in ONE launch, both intermediates in LDS (csrc/convchain.hip)
            fkey = (key, "fused")
            if cache.get("fkey") != fkey:
                cache.update(fkey=fkey, fplanes=(ops.weight_planes(w_in, n, pad_rows=32), ops.conv_core_planes(self.core_kernel, n),
                                                 ops.weight_planes(w_out, n)))
            f1, f2, f3 = cache["fplanes"]
            return ops.conv_chain(x, f1, f2, f3, self.bias, self.out_channels, self.kernel_size, self.stride, self.padding,
                                  self.dilation)
        out = HF.pointwise(x, w_in, None, "tadmm_ttconv_chain_in", p_in)
        core = self.core_kernel if x.dtype == self.core_kernel.dtype else self.core_kernel.to(x.dtype)
        out = F.conv2d(out, core, None, self.stride, self.padding, self.dilation, self.groups)
        return HF.pointwise(out, w_out, self.bias, "tadmm_ttconv_chain_out", p_out)

    def forward_flops(self, x):                                   # TTConv.py:155-195
        out, tt_flops, (h2, w2) = self._chains(x, True)
        kh, kw = self.kernel_size
        base_flops = h2 * w2 * kh * kw * self.in_channels * self.out_channels / 1e6
        tt_params = sum(p.numel() for p in self.in_tt_cores) + sum(p.numel() for p in self.out_tt_cores) \
            + self.core_kernel.numel()
        base_params = kh * kw * self.in_channels * self.out_channels
        print('baseline # params: {:.2f}K, tt # params: {:.2f}K'.format(base_params / 1000, tt_params / 1000))
        print('baseline # flops: {:.2f}M, tt # flops: {:.2f}M'.format(base_flops, tt_flops))
        return out, base_flops, tt_flops


def _chain_recover(cores: List[Tensor]) -> Tensor:
    """core_0 (core_1 (...)) left to right (TTConv.py:313-319, TTLinear.py:151-155, ttd.py:39-40)."""
    w = cores[0].reshape(-1, cores[0].shape[-1])
    for c in cores[1:]:
        w = HF.mm(w.reshape(-1, c.shape[0]), c.reshape(c.shape[0], -1))
    return w


class TTConv2dR(_TTConvBase):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = 'zeros', hp_dict=None, name=str,
                 dense_w: Tensor = None, dense_b: Tensor = None):
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                    hp_dict, name)
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        valid_padding_modes = {'zeros', 'reflect', 'replicate', 'circular'}
        if padding_mode not in valid_padding_modes:
            raise ValueError("padding_mode must be one of {}, but got padding_mode='{}'".format(
                valid_padding_modes, padding_mode))
        self._reversed_padding_repeated_twice = _reverse_repeat_tuple(self.padding, 2)
        self.kernel_shape = [out_channels, in_channels // groups, *self.kernel_size]
        self.filter_dim = int(self.kernel_shape[2] * self.kernel_shape[3])
        self.out_tt_cores = nn.ParameterList(
            [_empty(self.out_tt_ranks[i], self.out_tt_shapes[i], self.out_tt_ranks[i + 1])
             for i in range(self.out_tt_order)])
        self.conv_core = _empty(self.out_tt_ranks[-1], self.filter_dim, self.in_tt_ranks[0])
        self.in_tt_cores = nn.ParameterList(
            [_empty(self.in_tt_ranks[i], self.in_tt_shapes[i], self.in_tt_ranks[i + 1]) for i in range(self.in_tt_order)])
        if bias:
            self.bias = _zero_bias(self.out_channels)
            if dense_b is not None:
                self.bias.data = dense_b
        else:
            self.register_parameter('bias', None)
        if dense_w is not None:
            # reference quirk kept on purpose (TTConv.py:285-288): the (O,I,k^2) buffer is TT-decomposed
            # WITHOUT the (0,2,1) transpose, i.e. as a flat re-interpretation with modes out|k^2|in.
            flat = dense_w.detach().reshape(self.out_channels, -1)
            shapes = self.out_tt_shapes + [self.filter_dim] + self.in_tt_shapes
            cores = _decompose(flat, shapes, self.tt_ranks, KIND_TT_LINEAR)
            kq = self.out_tt_order
            for i, c in enumerate(cores):
                if i < kq:
                    self.out_tt_cores[i].data = c
                elif i == kq:
                    self.conv_core.data = c
                else:
                    self.in_tt_cores[i - kq - 1].data = c
        else:
            self.reset_parameters()

    def reset_parameters(self) -> None:
        for p in list(self.out_tt_cores) + [self.conv_core] + list(self.in_tt_cores):
            init.xavier_uniform_(p)
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(torch.empty(self.kernel_shape))
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _recover_weight(self):
        w = _chain_recover(list(self.out_tt_cores) + [self.conv_core] + list(self.in_tt_cores))
        return w.reshape(self.out_channels, self.filter_dim, self.in_channels).reshape(self.kernel_shape)

    def _conv_forward(self, x, weight):
        if self.padding_mode != 'zeros':
            return F.conv2d(F.pad(x, self._reversed_padding_repeated_twice, mode=self.padding_mode), weight, self.bias,
                            self.stride, _pair(0), self.dilation, self.groups)
        return F.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, x: Tensor) -> Tensor:
        return self._conv_forward(x, self._recover_weight())


class _TTLinearBase(HF.InferenceCacheMixin, nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, hp_dict=None, name: str = None,
                 dense_w: Tensor = None, dense_b: Tensor = None):
        super().__init__()
        self.tt_shapes = list(hp_dict.tt_shapes[name])
        self.tt_order = len(self.tt_shapes)
        self.out_tt_order, self.in_tt_order = _split_modes(self.tt_shapes, out_features, False)
        self.out_tt_shapes = self.tt_shapes[:self.out_tt_order]
        self.in_tt_shapes = self.tt_shapes[self.out_tt_order:]
        assert in_features == int(np.prod(self.in_tt_shapes))
        assert out_features == int(np.prod(self.out_tt_shapes))
        self.in_features, self.out_features = in_features, out_features
        self.tt_ranks = list(hp_dict.ranks[name])
        self.tt_cores = nn.ParameterList(
            [_empty(self.tt_ranks[i], self.tt_shapes[i], self.tt_ranks[i + 1]) for i in range(self.tt_order)])
        if bias:
            self.bias = _zero_bias(self.out_features)
            if dense_b is not None:
                self.bias.data = dense_b
        else:
            self.register_parameter('bias', None)
        if dense_w is not None:
            cores = _decompose(dense_w, self.tt_shapes, self.tt_ranks, KIND_TT_LINEAR)   # TTLinear.py:61-66
            for i, c in enumerate(cores):
                self.tt_cores[i].data = c
        else:
            self.reset_parameters()

    def _init_cores(self):
        for p in self.tt_cores:
            init.xavier_uniform_(p)


class TTLinearM(_TTLinearBase):
    def reset_parameters(self):
        self._init_cores()

    def _factors(self):
        """Input cores contracted into Win (r_q, in_features), output cores into Wout (out_features, r_q)."""
        q = self.out_tt_order
        w_out = _chain_recover(list(self.tt_cores[:q])).reshape(self.out_features, self.tt_ranks[q])
        w_in = _chain_recover(list(self.tt_cores[q:])).reshape(self.tt_ranks[q], self.in_features)
        return w_in, w_out

    def _fused_ok(self, x):
        q = self.out_tt_order
        align = 8 if x.dtype == torch.bfloat16 else 4
        if not (x.dtype in (torch.float32, torch.bfloat16) and 0 < q < self.tt_order
                and HF.fused_rank_ok(self.tt_ranks[q]) and self.in_features % align == 0):
            return False
        # the backward runs the same kernels with the gradient as X (row length out_features): a head whose width is
        # not 16-byte aligned (10 classes) trains through the per-core chain instead
        return self.out_features % align == 0 or not HF._needs_grad(x, self.bias, *self.tt_cores)

    def _dense_pays(self, x, r_q: int) -> bool:
        """bf16 inference only.  The contracted chain costs r_q (in + out) multiply-adds per token, the dense layer in * out:
        the recovered weight is used when the chain saves nothing (ratio >= 1: DeiT-small `proj`, 384 -> 384 through rank
        256 -- 12.5 us against 10.5 for the dense product under graph replay) or saves little over a long reduction
        (ratio >= 0.8 and in_features >= 1024: `fc2`).  `qkv` / `fc1` (ratio 0.89 / 0.83, K = 384) keep the chain kernel:
        1.4 - 1.65x the dense product."""
        if x.dtype != torch.bfloat16:
            return False
        chain, dense = r_q * (self.in_features + self.out_features), self.in_features * self.out_features
        return chain >= dense or (self.in_features >= 1024 and chain >= 0.8 * dense)

    def forward(self, x):
        """TTLinear.py:75-93.  One launch (`tadmm_ttlinear_fwd`) when the middle rank fits the fused kernel: the
        input cores are contracted into Win, the output cores into Wout, and y = Wout (Win x) + bias with the
        rank-r_q vector of a token held in LDS.  Otherwise the per-core GEMM chain below."""
        if not torch.is_grad_enabled():
            # inference fast path: the launch closure of the previous call, valid while no parameter changed (version
            # counters and storage addresses).  The full dispatch below costs more host time than a 20 us kernel takes;
            # the parameters are read from the modules' own dictionaries (indexing the ParameterList costs ~1 us an entry).
            c = self.__dict__.get("_chain_cache")
            if c is not None and c.get("fast_dtype") is x.dtype and c.get("fast_dev") == x.device \
                    and c.get("fast_key") == HF.param_key(*self.tt_cores._parameters.values(), self._parameters.get("bias")):
                return c["fast"](x)
        if self._fused_ok(x):
            grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.tt_cores)
            if grad and x.dtype == torch.float32:
                w_in, w_out = self._factors()
                return HF.linear_chain(x, w_in, w_out, self.bias)
            if not grad:
                n = 1 if x.dtype == torch.bfloat16 else 3
                cache = self.__dict__.setdefault("_chain_cache", {})
                key = (n, x.device, HF.param_key(*self.tt_cores))
                if cache.get("key") != key:
                    with torch.no_grad():
                        w_in, w_out = self._factors()
                    cache.update(key=key, w=(w_in, w_out), planes=(HF.planes_of(w_in, n, pad_rows=64),
                                                                   HF.planes_of(w_out, n, pad_cols=64)), dense=None)
                w_in, w_out = cache["w"]
                if self._dense_pays(x, w_in.shape[0]) and not HF._needs_grad(x, self.bias):
                    # long-K bf16 layers whose chain saves (almost) no arithmetic (DeiT-small fc2: 983 k of 1 180 k flop per
                    # token): every 64-token tile of the chain kernel pulls all of W_in through its CU, the library's dense
                    # kernel reuses a weight tile across 256 tokens -- 0.027 vs 0.020 ms.  Inference therefore multiplies by
                    # the recovered weight (cached with the other inference state), exactly what TTLinearR does every call
                    # (TTLinear.py:151-160).
                    bkey = None if self.bias is None else HF.param_key(self.bias)
                    if cache.get("dense") is None or cache.get("dense_bkey") != bkey:
                        with torch.no_grad():
                            cache["dense"] = HF.mm(w_out, w_in).to(x.dtype).contiguous()
                            cache["dense_bias"] = None if self.bias is None else self.bias.detach().to(x.dtype)
                            cache["dense_bkey"] = bkey
                    dense, dbias = cache["dense"], cache["dense_bias"]
                    fast = lambda t: F.linear(t, dense, dbias)
                else:
                    bias, planes = self.bias, cache["planes"]
                    fast = lambda t: HF.linear_chain(t, w_in, w_out, bias, planes)
                if not torch.is_grad_enabled():
                    cache.update(fast=fast, fast_dtype=x.dtype, fast_dev=x.device,
                                 fast_key=HF.param_key(*self.tt_cores._parameters.values(), self._parameters.get("bias")))
                return fast(x)
        if x.dtype == torch.bfloat16 and not (torch.is_grad_enabled() and (x.requires_grad or self.tt_cores[0].requires_grad)):
            return self._forward_bf16(x)
        return self._forward_chain(x)

    def _forward_chain(self, x):
        out_shape = list(x.shape)
        out_shape[-1] = self.out_features
        q = self.out_tt_order
        out = x
        for i in range(self.in_tt_order - 1, -1, -1):            # TTLinear.py:79-82
            k = self.in_tt_shapes[i] * self.tt_ranks[i + q + 1]
            out = HF.mm(out.reshape(-1, k), self.tt_cores[i + q].reshape(-1, k).t())
        for i in range(q - 1, -1, -1):                           # TTLinear.py:84-86
            r1 = self.tt_ranks[i + 1]
            out = HF.mm(self.tt_cores[i].reshape(-1, r1), out.reshape(-1, r1).t())
            out = out.reshape(self.tt_ranks[i], -1).t()
        out = out.reshape(self.out_features, -1).t().reshape(out_shape)
        if self.bias is not None:
            out = out + self.bias
        return out

    def _forward_bf16(self, x):
        """Inference in bfloat16 (the reference reaches this through AMP autocast, engines.py:285-289): the same
        chain, every product `A * Bt^T` with both operands contiguous along K, on the bf16 matrix cores with fp32
        accumulation (`tadmm_gemm_bf16_nt`); intermediate activations stay in bf16."""
        from . import ops
        out_shape = list(x.shape)
        out_shape[-1] = self.out_features
        q = self.out_tt_order
        cores = [c.detach().to(torch.bfloat16) for c in self.tt_cores]
        out = x
        for i in range(self.in_tt_order - 1, -1, -1):
            k = self.in_tt_shapes[i] * self.tt_ranks[i + q + 1]
            out = ops.mm_nt_bf16(out.reshape(-1, k).contiguous(), cores[i + q].reshape(-1, k))
        for i in range(q - 1, -1, -1):
            r1 = self.tt_ranks[i + 1]
            out = ops.mm_nt_bf16(cores[i].reshape(-1, r1), out.reshape(-1, r1).contiguous())
            out = out.reshape(self.tt_ranks[i], -1).t()
        out = out.reshape(self.out_features, -1).t().reshape(out_shape)
        if self.bias is not None:
            out = out + self.bias.to(torch.bfloat16)
        return out


class TTLinearR(_TTLinearBase):
    def reset_parameters(self):
        self._init_cores()
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            init.uniform_(self.bias, -bound, bound)

    def _recover_weight(self):
        return _chain_recover(list(self.tt_cores)).reshape(self.out_features, self.in_features)

    def forward(self, x):
        return HF.linear(x, self._recover_weight(), self.bias)
