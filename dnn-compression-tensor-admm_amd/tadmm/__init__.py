"""tadmm -- MI355X-native ADMM low-rank projection path (host-side mirror of the reference API).

Public surface mirrors the reference modules (admm.py, ttd.py, TTConv.py, TTLinear.py, TKConv.py,
TKLinear.py, utils.get_hp_dict); arithmetic runs in libtadmm_hip.so (csrc/) through the C ABI of
include/tadmm.h.  Importing the package does not need a GPU; computing does.
"""
from . import _cabi  # noqa: F401

__all__ = ["_cabi"]
