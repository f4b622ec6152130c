"""Rank tables (reference hp_dicts/*.py) and their selector (reference utils.py:258-400).

The numbers are shipped as JSON data (tadmm/data/hp_dicts.json, exported from the reference's
classes by tests/golden/make_hp_json.py).  ``get_hp_dict`` returns ONE cached object per table and
process holding *mutable* lists, mirroring the reference where every caller shares the class
attributes -- so the in-place rank clamp of the conv TT path (ttd.py:18-19 via admm.py:94) persists
exactly as it does there.  DeiT tables keep tuples (immutable), as in the reference.
"""
from __future__ import annotations

import json
import os
import re
from functools import lru_cache

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hp_dicts.json")


@lru_cache(maxsize=1)
def _tables():
    with open(_DATA) as f:
        return json.load(f)


class HyperParams:
    """Object with `.ranks` (and `.tt_shapes` for TT tables) keyed by state_dict parameter names."""

    def __init__(self, key, entry):
        self.table = key
        self.ranks = {}
        tup = entry.get("ranks_is_tuple", {})
        for k, v in entry["ranks"].items():
            self.ranks[k] = tuple(v) if tup.get(k) else (list(v) if isinstance(v, list) else v)
        if "tt_shapes" in entry:
            tup = entry.get("tt_shapes_is_tuple", {})
            self.tt_shapes = {k: (tuple(v) if tup.get(k) else list(v)) for k, v in entry["tt_shapes"].items()}

    def __repr__(self):
        return f"<HyperParams {self.table}: {len(self.ranks)} layers>"


_cache = {}


def table(key: str) -> HyperParams:
    """`key` = '<hp file stem>.<class name>', e.g. 'tt_resnet50_hp.HyperParamsDictGeneralRatio3x'."""
    if key not in _cache:
        t = _tables()
        if key not in t:
            # the reference's ladder names a few classes its hp files never define -> ImportError there too
            raise ImportError(f"cannot import name {key.split('.')[-1]!r} from 'hp_dicts.{key.split('.')[0]}'")
        _cache[key] = HyperParams(key, t[key])
    return _cache[key]


def table_keys():
    """Keys of every shipped rank table ('<hp file stem>.<class name>')."""
    return sorted(_tables())


def fresh_table(key: str) -> HyperParams:
    """An un-shared copy (tests / benchmarks that must not see earlier clamps)."""
    return HyperParams(key, _tables()[key])


# (format, model) -> {ratio or (ratio, tt_type): class name}; file stem is '<format>_<model>_hp'
_LADDER = {
    ("tk", "deit_tiny_patch16_224"): {"2": "HyperParamsDictRatio2x"},
    ("tt", "deit_tiny_patch16_224"): {"2": "HyperParamsDictRatio2x"},
    ("tt", "deit_small_patch16_224"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "resnet32"): {"1.5": "HyperParamsDictRatio1p5x", "2": "HyperParamsDictRatio2x",
                         "3": "HyperParamsDictRatio3x", "5": "HyperParamsDictRatio5x"},
    ("tt", "resnet32"): {"3": "HyperParamsDictRatio3x", "5": "HyperParamsDictRatio5x"},
    ("tk", "resnet56"): {"2": "HyperParamsDictRatio2x", "3": "HyperParamsDictRatio3x"},
    ("tt", "resnet56"): {"2": "HyperParamsDictRatio2x", "3": "HyperParamsDictRatio3x"},
    ("tk", "resnet18"): {"2": "HyperParamsDictRatio2x", "sc": "HyperParamsDictSC", "3": "HyperParamsDict3x",
                         "4": "HyperParamsDict4x", "5": "HyperParamsDict5x"},
    ("tt", "resnet18"): {("2", "general"): "HyperParamsDictGeneralRatio2x",
                         ("2", "special"): "HyperParamsDictSpecialRatio2x"},
    ("tt", "resnet50"): {("3", "general"): "HyperParamsDictGeneralRatio3x",
                         ("3", "special"): "HyperParamsDictSpecialRatio3x"},
    ("tk", "resnet50"): {"3": "HyperParamsDictRatio3x", "sc": "HyperParamsDictSC", "10": "HyperParamsDictRatio10x"},
    ("tk", "mobilenetv2"): {"2": "HyperParamsDictRatio2x"},
    ("tt", "mobilenetv2"): {"2": "HyperParamsDictRatio2x"},
    ("svd", "mobilenetv2"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "mobilenetv2_cifar"): {"2": "HyperParamsDictRatio2x"},
    ("svd", "mobilenetv2_cifar"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "densenet40"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "densenet121"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "densenet201"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "vgg16"): {"2": "HyperParamsDictRatio2x"},
    ("tk", "vgg16_bn"): {"2": "HyperParamsDictRatio2x", "10": "HyperParamsDictRatio10x"},
}


def get_hp_dict(model_name, ratio, format="none", tt_type="general"):
    """Same selection rule as the reference (utils.py:258-400): the model-name prefix
    `(tk|tt|svd)?[crm]?_?` overrides `format`; unknown (format, model) -> None; known model with an
    unsupported ratio -> Exception('ERROR: Unsupported compression ratio!')."""
    match = re.match(r"(tk|tt|svd)?[crm]?_?(.+)", model_name)
    if match.group(1):
        format, model_name = match.group(1), match.group(2)
    rungs = _LADDER.get((format, model_name))
    if rungs is None:
        return None
    cls = rungs.get(ratio)
    if cls is None:
        cls = rungs.get((ratio, tt_type))
    if cls is None:
        raise Exception("ERROR: Unsupported compression ratio!")
    return table(f"{format}_{model_name}_hp.{cls}")
