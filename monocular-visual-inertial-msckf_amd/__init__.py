"""MI355X-native MSCKF measurement-update engine (host side).

NumPy + ctypes only.  The compute path is the HIP library built from `csrc/`
(C-ABI in `include/msckf_mi355x.h`); there is no CPU fallback -- creating an
engine without the library (or without a GPU) raises.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
