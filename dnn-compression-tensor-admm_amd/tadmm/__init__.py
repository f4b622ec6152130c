"""tadmm -- MI355X-native ADMM low-rank projection path (host-side mirror of the reference API).

Public surface mirrors the reference modules (admm.py, ttd.py, TTConv.py, TTLinear.py, TKConv.py,
TKLinear.py, utils.get_hp_dict); arithmetic runs in libtadmm_hip.so (csrc/) through the C ABI of
include/tadmm.h.  Importing the package does not need a GPU; computing does, and there is no CPU
fallback: a missing library or device raises.
"""
from . import _cabi  # noqa: F401
from .hp import get_hp_dict  # noqa: F401


def __getattr__(name):
    # heavier sub-modules (they import torch.nn) are resolved lazily
    import importlib
    table = {
        "ADMM": ("admm", "ADMM"), "ten2tt": ("ttd", "ten2tt"), "tt2ten": ("ttd", "tt2ten"),
        "TTConv2dM": ("tt_layers", "TTConv2dM"), "TTConv2dR": ("tt_layers", "TTConv2dR"),
        "TTLinearM": ("tt_layers", "TTLinearM"), "TTLinearR": ("tt_layers", "TTLinearR"),
        "TKConv2dC": ("tk_layers", "TKConv2dC"), "TKConv2dM": ("tk_layers", "TKConv2dM"),
        "TKConv2dR": ("tk_layers", "TKConv2dR"), "TKLinearM": ("tk_layers", "TKLinearM"),
        "TKLinearR": ("tk_layers", "TKLinearR"),
    }
    if name in table:
        mod, attr = table[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
