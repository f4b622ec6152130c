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


def deit_small_shape(name: str) -> Tuple[int, ...]:
    e = 384
    if name.endswith("attn.qkv.weight"):
        return (3 * e, e)
    if name.endswith("attn.proj.weight"):
        return (e, e)
    if name.endswith("mlp.fc1.weight"):
        return (4 * e, e)
    if name.endswith("mlp.fc2.weight"):
        return (e, 4 * e)
    raise KeyError(name)


CONFIGS = {
    # name: (hp table key, format, shape function)
    "resnet50_tt": ("tt_resnet50_hp.HyperParamsDictGeneralRatio3x", "tt", resnet50_shape),
    "resnet18_tt": ("tt_resnet18_hp.HyperParamsDictGeneralRatio2x", "tt", resnet18_shape),
    "deit_small_tt": ("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x", "tt", deit_small_shape),
    "resnet32_tk": ("tk_resnet32_hp.HyperParamsDictRatio3x", "tk", resnet_cifar_shape),
    "resnet32_tt": ("tt_resnet32_hp.HyperParamsDictRatio3x", "tt", resnet_cifar_shape),
}


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
