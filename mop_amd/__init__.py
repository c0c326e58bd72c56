"""mop_amd -- MI355X-native Mixture-of-Products attention (drop-in for Eran-BA/MoP's attention path).

Host side: PyTorch-ROCm modules with the reference's constructor / forward / state_dict
surface (`mop_amd.nn`).  Device side: hand-written gfx950 HIP kernels behind the C ABI in
include/mopk.h, loaded with ctypes (`mop_amd._lib`).  There is no CPU fallback.
"""
from . import nn  # noqa: F401
from .ops import get_precision, set_precision  # noqa: F401

__version__ = "0.1.0"
