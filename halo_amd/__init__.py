"""halo_amd -- MI355X-native (gfx950) implementation of HALO's acquisition-scoring path.

Host side: Python mirroring the reference's plugin surface (halo_amd.core.*), calling
hand-written HIP kernels through the C ABI in include/halo_hip.h (halo_amd/csrc/libhalo_hip.so).
There is no CPU fallback: every op raises if the HIP library or a ROCm device is missing.
"""
__version__ = "0.3.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)
from ._install import install, uninstall  # noqa: F401
