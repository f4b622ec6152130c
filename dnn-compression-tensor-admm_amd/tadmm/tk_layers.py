"""Tucker-2 factorised layers with the reference's constructor signatures and state_dict keys
(TKConv.py:26-325, TKLinear.py:23-122).

  TKConv2dC : 1x1 conv -> k x k conv -> 1x1 conv          (first_kernel, core_kernel, last_kernel, bias)
  TKConv2dM : linear on NHWC -> k x k conv -> linear      (first_factor, core_kernel, last_factor, bias)
  TKConv2dR : rebuild dense kernel + conv2d               (first_factor, core_tensor, last_factor, bias)
  TKLinearM : three chained linears                       (first_factor, core_tensor, last_factor, bias)
  TKLinearR : rebuild dense weight + linear               (same keys)

`dense_w` is decomposed by the device HOSVD+HOOI of `tadmm.tucker` (parity unpinned, see there):
first = U_in^T, last = U_out, core = W x_0 U_out^T x_1 U_in^T  (TKConv.py:79-83).  The channel-mixing
contractions run through `functional.mm` / `functional.linear` on the fp32 matrix cores; the k x k core
conv is the device library's conv2d as in the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor, nn
from torch.nn import init
from torch.nn.modules.utils import _pair, _reverse_repeat_tuple

from . import functional as HF
from . import ops
from . import tucker


def _empty(*shape):
    return nn.Parameter(torch.empty(*shape))


def _tucker_factors(dense_w: Tensor, out_rank: int, in_rank: int):
    dev = dense_w.device if dense_w.is_cuda else torch.device("cuda", torch.cuda.current_device())
    core, (u_out, u_in), _ = tucker.partial_tucker(dense_w.detach().to(dev), [out_rank, in_rank])
    return core, u_out, u_in


def _recover(core: Tensor, last_factor: Tensor, first_factor: Tensor) -> Tensor:
    """tucker_to_tensor((core, (last_factor, first_factor^T)))  (TKConv.py:313-314, TKLinear.py:117-119)."""
    o, ri = last_factor.shape[0], first_factor.shape[0]
    t = HF.mm(last_factor, core.reshape(core.shape[0], -1)).reshape(o, ri, -1)          # (O, r_in, rest)
    rest = t.shape[2]
    t = HF.mm(t.permute(0, 2, 1).reshape(-1, ri), first_factor)                          # (O*rest, I)
    return t.reshape(o, rest, -1).permute(0, 2, 1).reshape(o, first_factor.shape[1], *core.shape[2:])


class _TKConvBase(HF.InferenceCacheMixin, nn.Module):
    def _setup(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode, ranks):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.ranks = ranks
        self.in_rank, self.out_rank = self.ranks[1], self.ranks[0]
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.transposed = False
        self.output_padding = _pair(0)
        self.groups = groups
        self.padding_mode = padding_mode

    def _make_bias(self, bias, dense_b):
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.out_channels))
            if dense_b is not None:
                self.bias.data = dense_b
        else:
            self.register_parameter('bias', None)


def _check_mode(groups, padding_mode):
    if groups != 1:
        raise ValueError("groups must be 1 in this mode")
    if padding_mode != 'zeros':
        raise ValueError("padding_mode must be zero in this mode")


class TKConv2dC(_TKConvBase):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = 'zeros', hp_dict=None, name: str = None,
                 dense_w: Tensor = None, dense_b: Tensor = None):
        _check_mode(groups, padding_mode)
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                    hp_dict.ranks[name])
        self.first_kernel = _empty(self.in_rank, self.in_channels, 1, 1)
        self.core_kernel = _empty(self.out_rank, self.in_rank, *self.kernel_size)
        self.last_kernel = _empty(self.out_channels, self.out_rank, 1, 1)
        self._make_bias(bias, dense_b)
        if dense_w is not None:
            core, u_out, u_in = _tucker_factors(dense_w, self.out_rank, self.in_rank)
            self.first_kernel.data = u_in.t().contiguous()[:, :, None, None]
            self.last_kernel.data = u_out.contiguous()[:, :, None, None]
            self.core_kernel.data = core
        else:
            self.reset_parameters()

    def reset_parameters(self) -> None:
        for p in (self.first_kernel, self.core_kernel, self.last_kernel):
            init.xavier_uniform_(p)

    def _stages(self, x):
        # a 1x1 conv is a per-pixel channel mix: one chain-kernel launch on the NCHW tensor as it stands
        # (`tadmm_tucker_1x1`: no NHWC copies, bias in the epilogue)
        grad = torch.is_grad_enabled()
        cache = None if grad else self.__dict__.setdefault("_plane_cache", {})
        w1 = self.first_kernel.reshape(self.in_rank, self.in_channels)
        w3 = self.last_kernel.reshape(self.out_channels, self.out_rank)
        n = 1 if x.dtype == torch.bfloat16 else 3
        p1 = None if grad else HF.planes_of(self.first_kernel.reshape(self.in_rank, self.in_channels), n, cache=cache,
                                            tag="first")
        f1 = HF.pointwise(x, w1, None, "tadmm_tucker_1x1", p1)
        core = self.core_kernel if x.dtype == self.core_kernel.dtype else self.core_kernel.to(x.dtype)
        f2 = F.conv2d(f1, core, None, self.stride, self.padding, self.dilation, self.groups)
        p3 = None if grad else HF.planes_of(w3, n, cache=cache, tag="last")
        f3 = HF.pointwise(f2, w3, self.bias, "tadmm_tucker_1x1", p3)
        return f1, f2, f3

    def _fused(self, x, w1, core, w3):
        """The whole layer in one launch when the planes are small (csrc/convchain.hip); None when it does not apply."""
        if torch.is_grad_enabled() or self.groups != 1 or not ops.conv_chain_pays(
                x, w1.shape[0], w3.shape[1], self.kernel_size, self.stride, self.padding, self.dilation):
            return None
        n = 1 if x.dtype == torch.bfloat16 else 3
        cache = self.__dict__.setdefault("_fused_cache", {})
        key = (n, x.device, HF.param_key(w1, core, w3))
        if cache.get("key") != key:
            cache.update(key=key, planes=(ops.weight_planes(w1.detach(), n, pad_rows=32), ops.conv_core_planes(core, n),
                                          ops.weight_planes(w3.detach(), n)))
        p1, p2, p3 = cache["planes"]
        return ops.conv_chain(x, p1, p2, p3, self.bias, self.out_channels, self.kernel_size, self.stride, self.padding,
                              self.dilation)

    def forward(self, x):
        y = self._fused(x, self.first_kernel.reshape(self.in_rank, self.in_channels), self.core_kernel,
                        self.last_kernel.reshape(self.out_channels, self.out_rank))
        return y if y is not None else self._stages(x)[2]

    def forward_features(self, x):                                # TKConv.py:100-109
        f1, f2, f3 = self._stages(x)
        return f3, [f1, f2, f3]

    def forward_flops(self, x):                                   # TKConv.py:111-134
        f1, f2, out = self._stages(x)
        compr_params = (self.first_kernel.numel() + self.core_kernel.numel() + self.last_kernel.numel()) / 1000
        compr_flops = f1.shape[2] * f1.shape[3] * self.first_kernel.numel() / 1e6
        compr_flops += f2.shape[2] * f2.shape[3] * self.core_kernel.numel() / 1e6
        h2, w2 = out.shape[2], out.shape[3]
        compr_flops += h2 * w2 * self.last_kernel.numel() / 1e6
        kh, kw = self.kernel_size
        base_params = kh * kw * self.in_channels * self.out_channels / 1000
        base_flops = h2 * w2 * kh * kw * self.in_channels * self.out_channels / 1e6
        print('baseline # params: {:.2f}K\t compressed # params: {:.2f}K\t '
              'baseline # flops: {:.2f}M\t compressed # flops: {:.2f}M'.format(base_params, compr_params, base_flops,
                                                                               compr_flops))
        return out, base_flops, compr_flops

    def extra_repr(self):
        return ('first_conv(in={}, out={}, kernel_size=(1, 1), bias=False), '
                'core_conv(in={}, out={}, kernel_size={}, stride={}, padding={}, bias={}), '
                'last_conv(in={}, out={}, kernel_size=(1, 1), bias=False)').format(
            self.in_channels, self.in_rank, self.in_rank, self.out_rank, self.kernel_size, self.stride, self.padding,
            self.bias is None, self.out_rank, self.out_channels)


class TKConv2dM(_TKConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode='zeros', hp_dict=None, name=str, dense_w=None, dense_b=None):
        _check_mode(groups, padding_mode)
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                    hp_dict.ranks[name])
        self.first_factor = _empty(self.in_rank, in_channels)
        self.core_kernel = _empty(self.out_rank, self.in_rank, *self.kernel_size)
        self.last_factor = _empty(out_channels, self.out_rank)
        self._make_bias(bias, dense_b)
        if dense_w is not None:
            core, u_out, u_in = _tucker_factors(dense_w, self.out_rank, self.in_rank)
            self.first_factor.data = u_in.t().contiguous()
            self.last_factor.data = u_out.contiguous()
            self.core_kernel.data = core
        else:
            self.reset_parameters()

    def reset_parameters(self):
        for p in (self.first_factor, self.last_factor, self.core_kernel):
            init.xavier_uniform_(p)

    def forward(self, x: Tensor) -> Tensor:                       # TKConv.py:210-214
        y = TKConv2dC._fused(self, x, self.first_factor, self.core_kernel, self.last_factor)
        if y is not None:
            return y
        grad = torch.is_grad_enabled()
        cache = None if grad else self.__dict__.setdefault("_plane_cache", {})
        n = 1 if x.dtype == torch.bfloat16 else 3
        p1 = None if grad else HF.planes_of(self.first_factor, n, cache=cache, tag="first")
        out = HF.pointwise(x, self.first_factor, None, "tadmm_tucker_1x1", p1)
        core = self.core_kernel if x.dtype == self.core_kernel.dtype else self.core_kernel.to(x.dtype)
        out = F.conv2d(out, core, None, self.stride, self.padding, self.dilation, self.groups)
        p3 = None if grad else HF.planes_of(self.last_factor, n, cache=cache, tag="last")
        return HF.pointwise(out, self.last_factor, self.bias, "tadmm_tucker_1x1", p3)


class TKConv2dR(_TKConvBase):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = 'zeros', hp_dict=None, name: str = None,
                 dense_w: Tensor = None, dense_b: Tensor = None):
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                    list(hp_dict.ranks[name]))
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
        self.first_factor = _empty(self.in_rank, in_channels)
        self.core_tensor = _empty(self.out_rank, self.in_rank, self.kernel_shape[2], self.kernel_shape[3])
        self.last_factor = _empty(out_channels, self.out_rank)
        self._make_bias(bias, dense_b)
        if dense_w is not None:
            core, u_out, u_in = _tucker_factors(dense_w, self.out_rank, self.in_rank)
            self.first_factor.data = u_in.t().contiguous()
            self.last_factor.data = u_out.contiguous()
            self.core_tensor.data = core
        else:
            self.reset_parameters()

    def reset_parameters(self):
        for p in (self.first_factor, self.core_tensor, self.last_factor):
            init.xavier_uniform_(p)
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(torch.empty(self.kernel_shape))
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _recover_weight(self):
        return _recover(self.core_tensor, self.last_factor, self.first_factor)

    def _conv_forward(self, x, weight):
        if self.padding_mode != 'zeros':
            return F.conv2d(F.pad(x, self._reversed_padding_repeated_twice, mode=self.padding_mode), weight, self.bias,
                            self.stride, _pair(0), self.dilation, self.groups)
        return F.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, x: Tensor) -> Tensor:
        return self._conv_forward(x, self._recover_weight())


class _TKLinearBase(HF.InferenceCacheMixin, nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, hp_dict=None, name: str = None,
                 dense_w: Tensor = None, dense_b: Tensor = None) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.ranks = hp_dict.ranks[name]
        self.in_rank, self.out_rank = self.ranks[1], self.ranks[0]
        self.first_factor = _empty(self.in_rank, self.in_features)
        self.core_tensor = _empty(self.out_rank, self.in_rank)
        self.last_factor = _empty(self.out_features, self.out_rank)
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.out_features))   # reference: uninitialised (TKLinear.py); see tt_layers._zero_bias
            if dense_b is not None:
                self.bias.data = dense_b
        else:
            self.register_parameter('bias', None)
        if dense_w is not None:
            core, u_out, u_in = _tucker_factors(dense_w, self.out_rank, self.in_rank)
            self.first_factor.data = u_in.t().contiguous()
            self.last_factor.data = u_out.contiguous()
            self.core_tensor.data = core
        else:
            self.reset_parameters()

    def reset_parameters(self) -> None:
        for p in (self.first_factor, self.core_tensor, self.last_factor):
            init.kaiming_uniform_(p, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            init.uniform_(self.bias, -bound, bound)


class TKLinearM(_TKLinearBase):
    def forward(self, x: Tensor) -> Tensor:                       # TKLinear.py:66-71
        """Three products in the reference; here the small core is contracted into the input factor and the layer is one
        launch of the fused chain (`tadmm_ttlinear_fwd`: y = last (core first) x + bias, the out_rank-vector of a token in
        LDS) whenever out_rank fits it; otherwise three strided GEMMs."""
        align = 8 if x.dtype == torch.bfloat16 else 4
        params = (self.first_factor, self.core_tensor, self.last_factor)
        # (the backward runs the fused kernel with the gradient as X: its row length out_features must be aligned too)
        if (x.dtype in (torch.float32, torch.bfloat16) and HF.fused_rank_ok(self.out_rank) and x.is_cuda
                and self.in_features % align == 0
                and (self.out_features % align == 0 or not HF._needs_grad(x, self.bias, *params))):
            grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            if grad and x.dtype == torch.float32:
                w_in = HF.mm(self.core_tensor, self.first_factor)             # (out_rank, in_features), differentiable
                return HF.linear_chain(x, w_in, self.last_factor, self.bias)
            if not grad:
                n = 1 if x.dtype == torch.bfloat16 else 3
                cache = self.__dict__.setdefault("_chain_cache", {})
                key = (n, x.device, HF.param_key(*params))
                if cache.get("key") != key:
                    with torch.no_grad():
                        w_in = HF.mm(self.core_tensor, self.first_factor)
                    cache.update(key=key, w_in=w_in, planes=(HF.planes_of(w_in, n, pad_rows=64),
                                                             HF.planes_of(self.last_factor, n, pad_cols=64)))
                return HF.linear_chain(x, cache["w_in"], self.last_factor, self.bias, cache["planes"])
        out = HF.linear(x, self.first_factor)
        out = HF.linear(out, self.core_tensor)
        return HF.linear(out, self.last_factor, self.bias)


class TKLinearR(_TKLinearBase):
    def _recover_weight(self):
        return _recover(self.core_tensor, self.last_factor, self.first_factor)

    def forward(self, x: Tensor) -> Tensor:
        return HF.linear(x, self._recover_weight(), self.bias)
