"""halo_amd -- MI355X-native (gfx950) implementation of HALO's acquisition-scoring path.

Host side: Python mirroring the reference's plugin surface (halo_amd.core.*), calling
hand-written HIP kernels through the C ABI in include/halo_hip.h (halo_amd/csrc/libhalo_hip.so).
There is no CPU fallback: every op raises if the HIP library or a ROCm device is missing.
"""
__version__ = "0.5.0"

import os as _os
import sys as _sys


def configure(hw_queues=None):
    """Explicit, opt-in process settings for the acquisition.  `import halo_amd` itself changes NOTHING in the process
    environment (round 3 set GPU_MAX_HW_QUEUES at import; a library must not rewrite a runtime-wide setting of the training
    process it is imported into).

    hw_queues: ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and the variable is read once, when the
    HIP runtime initialises.  Pool scoring in 16-image steps (bench.py: 1 + 3 + 1 streams) measures fastest on TWO queues (kernels
    of different streams still overlap inside a queue; with three or more, the short kernels that follow a long one on the scoring
    stream start late: bench.py +2 %, --source lowres +10 %, --branch ripu +6 %).  RegionSelection at the reference's loader batch
    of one does NOT: an image's GPU time is mostly one single-workgroup selection sweep, a queue holds one of them, and the driver
    runs 0.47-0.51 ms/image on ROCm's default of four against 0.58 on two (profiles/r06_hw_queues.txt) -- leave the default alone
    for it.  The setting also governs the training iterations' streams (DDP / RCCL communication, H2D copies): the caller's decision.
    Call this before the first HIP call of the process (bench.py and tools/ do); returns the value in force, or raises
    RuntimeError when the runtime is already up with a different one."""
    if hw_queues is None:
        return _os.environ.get("GPU_MAX_HW_QUEUES")
    want = str(int(hw_queues))
    have = _os.environ.get("GPU_MAX_HW_QUEUES")
    if have == want:
        return want
    torch = _sys.modules.get("torch")
    if torch is not None and getattr(torch, "cuda", None) is not None and torch.cuda.is_initialized():
        raise RuntimeError("halo_amd.configure(hw_queues=%s): the HIP runtime is already initialised (GPU_MAX_HW_QUEUES=%s); "
                           "call configure() before the first HIP call" % (want, have or "unset: ROCm's default, 4"))
    _os.environ["GPU_MAX_HW_QUEUES"] = want
    return want


from . import _lib  # noqa: F401,E402  (does not load the .so until first use)
from ._host import usable_cpus, host_threads_per_rank  # noqa: F401,E402
from ._install import install, uninstall  # noqa: F401,E402


def release_workspaces():
    """Drop the cached side streams' scratch buffers (scorer + selector workspaces, ~16 MB per 1024x2048 image and stream
    for mask radius 5; bench-sized 16-image select slots hold ~250 MB each).  They are re-created on demand; call between
    acquisition rounds when the training iterations need the memory.  Work already enqueued keeps its buffers alive
    through the caching allocator's stream ordering, so this is safe while kernels are still running."""
    from .core.active import build, floating_region
    floating_region._WS.clear()
    for slot_list in list(build._SIDE.values()) + list(build._SLOTS.values()):
        del slot_list[:]
    build._SIDE.clear()
    build._SLOTS.clear()
