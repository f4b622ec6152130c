"""Synthetic stand-ins for the BASELINE.json configurations.

torchvision / timm are absent, so the parameter shapes of the compressed layers are re-derived
from the architectures (ResNet-50/18 ImageNet, ResNet-32 CIFAR: resnet_cifar.py:36-38,83-85,
DeiT-small).  `SyntheticModel` exposes `named_parameters()` with exactly the state_dict names the
rank tables are keyed by, in table order, initialised N(0, 2/fan_in) from a seeded generator.
"""
from __future__ import annotations

import re
from typing import Dict, Tuple

import torch

from . import hp


def resnet50_shape(name: str) -> Tuple[int, ...]:
    m = re.match(r"layer(\d)\.(\d+)\.conv(\d)\.weight", name)
    L, b, c = int(m.group(1)), int(m.group(2)), int(m.group(3))
    p = 64 * 2 ** (L - 1)
    if c == 1:
        inp = (64 if L == 1 else 2 * p) if b == 0 else 4 * p
        return (p, inp, 1, 1)
    if c == 2:
        return (p, p, 3, 3)
    return (4 * p, p, 1, 1)


def resnet18_shape(name: str) -> Tuple[int, ...]:
    m = re.match(r"layer(\d)\.(\d+)\.conv(\d)\.weight", name)
    L, b, c = int(m.group(1)), int(m.group(2)), int(m.group(3))
    p = 64 * 2 ** (L - 1)
    inp = p // 2 if (c == 1 and b == 0 and L > 1) else p
    return (p, inp, 3, 3)


def resnet_cifar_shape(name: str) -> Tuple[int, ...]:
    m = re.match(r"layer(\d)\.(\d+)\.conv(\d)\.weight", name)
    L, b, c = int(m.group(1)), int(m.group(2)), int(m.group(3))
    p = 16 * 2 ** (L - 1)
    inp = p // 2 if (c == 1 and b == 0 and L > 1) else p
    return (p, inp, 3, 3)


def deit_small_shape(name: str, e: int = 384) -> Tuple[int, ...]:
    if name.endswith("attn.qkv.weight"):
        return (3 * e, e)
    if name.endswith("attn.proj.weight"):
        return (e, e)
    if name.endswith("mlp.fc1.weight"):
        return (4 * e, e)
    if name.endswith("mlp.fc2.weight"):
        return (e, 4 * e)
    raise KeyError(name)


# VGG-16 convolutions 2..13 in order (the first one, 3 -> 64, is in no table); the timm-style head the tables are keyed
# by (`pre_logits.fc1` = 7x7 convolution 512 -> 4096, `pre_logits.fc2` = 1x1 convolution 4096 -> 4096)
_VGG16_CONVS = [(64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (256, 256), (512, 256), (512, 512), (512, 512),
                (512, 512), (512, 512), (512, 512)]


def _vgg16_shape_fn(table_key: str):
    feats = [n for n in hp.table(table_key).ranks if n.startswith("features.")]

    def fn(name: str) -> Tuple[int, ...]:
        if name == "pre_logits.fc1.weight":
            return (4096, 512, 7, 7)
        if name == "pre_logits.fc2.weight":
            return (4096, 4096, 1, 1)
        o, i = _VGG16_CONVS[feats.index(name)]
        return (o, i, 3, 3)
    return fn


def densenet_inet_shape(name: str) -> Tuple[int, ...]:
    """DenseNet-121 / -201 (growth 32, bottleneck 4 x 32, first block at 64 channels, transitions halve): the tables hold the
    3x3 `conv2` of the dense layers, and (121) a few 1x1 `conv1` / transition kernels."""
    if name.endswith("conv2.weight"):
        return (32, 128, 3, 3)
    m = re.match(r"features\.transition(\d)\.conv\.weight", name)
    if m:
        c = 128 * 2 ** int(m.group(1))              # channels entering transition 1, 2, 3 of DenseNet-121
        return (c // 2, c, 1, 1)
    m = re.match(r"features\.denseblock(\d)\.denselayer(\d+)\.conv1\.weight", name)
    start = {1: 64, 2: 128, 3: 256, 4: 512}[int(m.group(1))]
    return (128, start + 32 * (int(m.group(2)) - 1), 1, 1)


# mobilenetv2_cifar.py:62-79: (input channels, output channels) of the 17 bottlenecks, expansion 6 (1 for the first)
_MBV2_CIFAR = [(32, 16), (16, 24), (24, 24), (24, 32), (32, 32), (32, 32), (32, 64), (64, 64), (64, 64), (64, 64), (64, 96),
               (96, 96), (96, 96), (96, 160), (160, 160), (160, 160), (160, 320)]


def mobilenetv2_cifar_shape(name: str) -> Tuple[int, ...]:
    """The 1x1 kernels of mobilenetv2_cifar.py: `bottlenecks.i.conv1` (expansion), `.conv3` (projection), final `conv1`."""
    if name == "conv1.weight":
        return (1280, 320, 1, 1)
    m = re.match(r"bottlenecks\.(\d+)\.conv([13])\.weight", name)
    i = int(m.group(1))
    cin, cout = _MBV2_CIFAR[i]
    c = (1 if i == 0 else 6) * cin
    return (c, cin, 1, 1) if m.group(2) == "1" else (cout, c, 1, 1)


def mobilenetv2_inet_shape(name: str) -> Tuple[int, ...]:
    """ImageNet MobileNetV2 under the flat-Sequential names of the SVD table: `features.k.conv.0` (expansion) and
    `.conv.6` (projection) of inverted residual k (same channel plan as the CIFAR model), `conv.0` = the 1280-wide head."""
    if name == "conv.0.weight":
        return (1280, 320, 1, 1)
    m = re.match(r"features\.(\d+)\.conv\.([06])\.weight", name)
    cin, cout = _MBV2_CIFAR[int(m.group(1)) - 1]
    return (6 * cin, cin, 1, 1) if m.group(2) == "0" else (cout, 6 * cin, 1, 1)


def densenet40_shape(name: str) -> Tuple[int, ...]:
    """densenet_cifar.py:81-106,141-142: depth 40, growth 16, basic blocks (one 3x3 conv per layer, 12 per block),
    32 channels into block 1, transitions with reduction 0.5."""
    m = re.match(r"block(\d)\.layer\.(\d+)\.conv1\.weight", name)
    start = {1: 32, 2: 112, 3: 152}
    if m:
        return (16, start[int(m.group(1))] + 16 * int(m.group(2)), 3, 3)
    m = re.match(r"trans(\d)\.conv1\.weight", name)
    cin = {1: 224, 2: 304}[int(m.group(1))]
    return (cin // 2, cin, 1, 1)


CONFIGS = {
    # name: (hp table key, format, shape function)
    "resnet50_tt": ("tt_resnet50_hp.HyperParamsDictGeneralRatio3x", "tt", resnet50_shape),
    "resnet18_tt": ("tt_resnet18_hp.HyperParamsDictGeneralRatio2x", "tt", resnet18_shape),
    "deit_small_tt": ("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x", "tt", deit_small_shape),
    "resnet32_tk": ("tk_resnet32_hp.HyperParamsDictRatio3x", "tk", resnet_cifar_shape),
    "resnet32_tt": ("tt_resnet32_hp.HyperParamsDictRatio3x", "tt", resnet_cifar_shape),
    # 4096-wide classifier Grams: streamed Jacobi pairs (csrc/jacobi.hip: jacobi_tick_stream_kernel)
    "vgg16_tk": ("tk_vgg16_hp.HyperParamsDictRatio2x", "tk", _vgg16_shape_fn("tk_vgg16_hp.HyperParamsDictRatio2x")),
    "vgg16_bn_tk": ("tk_vgg16_bn_hp.HyperParamsDictRatio10x", "tk",
                    _vgg16_shape_fn("tk_vgg16_bn_hp.HyperParamsDictRatio10x")),
}


def shape_fn_for(table_key: str):
    """Shape function of the architecture a rank table (key of data/hp_dicts.json, e.g. `tk_resnet18_hp.HyperParamsDict3x`)
    belongs to, or None for the families whose layer shapes are not re-derived here (DenseNet-264)."""
    mod = table_key.split(".")[0]
    if "_vgg16" in mod:
        return _vgg16_shape_fn(table_key)
    if "_resnet18_" in mod:
        return resnet18_shape
    if "_resnet50_" in mod:
        return resnet50_shape
    if "_resnet32_" in mod or "_resnet56_" in mod:
        return resnet_cifar_shape
    if "_deit_small_" in mod or "_vit_small_" in mod:
        return deit_small_shape
    if "_deit_tiny_" in mod:
        return lambda name: deit_small_shape(name, 192)
    if "_densenet201_" in mod or "_densenet121_" in mod:
        return densenet_inet_shape
    if "_densenet40_" in mod:
        return densenet40_shape
    if "_mobilenetv2_cifar_" in mod:
        return mobilenetv2_cifar_shape
    if mod in ("tt_mobilenetv2_hp", "tk_mobilenetv2_hp"):
        # the TT table states every 1x1 kernel as tt_shapes [O, 1, I]; the Tucker table (timm names) lists the same 33
        # kernels in the same order
        tts = list(hp.table("tt_mobilenetv2_hp.HyperParamsDictRatio2x").tt_shapes.values())
        names = list(hp.table(table_key).ranks)
        return lambda name: (tts[names.index(name)][0], tts[names.index(name)][2], 1, 1)
    if mod == "svd_mobilenetv2_hp":
        return mobilenetv2_inet_shape
    return None


class SyntheticModel(torch.nn.Module):
    """Bare parameters under the reference's state_dict names (nested modules for the dots)."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.flat = torch.nn.ParameterList()
        for name, shp in shapes.items():
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            self.flat.append(torch.nn.Parameter(torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5))
        self._names = list(shapes.keys())

    def named_parameters(self, *a, **k):
        # flat storage (table order = draw order of the generator), reference state_dict names
        for name, p in zip(self._names, self.flat):
            yield name, p


def build(config: str, seed: int = 0, fresh_table: bool = True):
    """Returns (model on CPU, hp_dict, format)."""
    key, fmt, fn = CONFIGS[config]
    table = hp.fresh_table(key) if fresh_table else hp.table(key)
    shapes = {name: fn(name) for name in table.ranks}
    return SyntheticModel(shapes, seed), table, fmt
