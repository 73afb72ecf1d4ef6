"""halo_amd -- MI355X-native (gfx950) implementation of HALO's acquisition-scoring path.

Host side: Python mirroring the reference's plugin surface (halo_amd.core.*), calling
hand-written HIP kernels through the C ABI in include/halo_hip.h (halo_amd/csrc/libhalo_hip.so).
There is no CPU fallback: every op raises if the HIP library or a ROCm device is missing.
"""
__version__ = "0.4.0"

import os as _os
import sys as _sys


HW_QUEUES = "2"


def _configure_hw_queues():
    """ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The acquisition keeps up to six streams busy
    (RegionSelection: 4 side streams + the caller's; bench.py: 1 + 3 + 1), and it runs FASTEST on TWO queues: kernels of
    different streams still overlap inside a queue, while with three or more queues the short kernels that follow a long one on
    the scoring stream start 40 us late and run up to 3x slower (k_box3_minmax 127 vs 43 us; bench.py, interleaved: 1355-1392
    images/s on 1-2 queues, 1337-1368 on 3-8; --source lowres 5900 vs 5260, --branch ripu 10 590 vs 9 680;
    profiles/r03_hw_queues.txt).  The variable is read when the HIP runtime initialises, so it is set here -- at import,
    normally the first lines of train.py -- unless the user chose a value or the runtime is already up (then RegionSelection
    warns once)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    torch = _sys.modules.get("torch")
    if torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized():
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = HW_QUEUES


_configure_hw_queues()

from . import _lib  # noqa: F401,E402  (does not load the .so until first use)
from ._install import install, uninstall  # noqa: F401,E402


def release_workspaces():
    """Drop the cached side streams' scratch buffers (scorer + selector workspaces, ~16 MB per 1024x2048 image and stream
    for mask radius 5; bench-sized 16-image select slots hold ~250 MB each).  They are re-created on demand; call between
    acquisition rounds when the training iterations need the memory.  Work already enqueued keeps its buffers alive
    through the caching allocator's stream ordering, so this is safe while kernels are still running."""
    from .core.active import build, floating_region
    floating_region._WS.clear()
    for slot_list in build._SIDE.values():
        del slot_list[:]
    build._SIDE.clear()
