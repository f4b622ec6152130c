"""Drop-in alias of the reference module `TKLinear.py`: re-exports the MI355X-native implementation."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from tadmm.tk_layers import TKLinearM, TKLinearR  # noqa: E402,F401
