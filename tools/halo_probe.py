"""Measurement aids (NOT product code, not in the C ABI of include/halo_hip.h): tools/libhalo_probe.so.

    from tools import halo_probe
    halo_probe.flat_read_gbps(tensor)              # this box's own ceiling for a streaming read of `tensor`
    halo_probe.alloc_contiguous / probe_streaming / contiguous_memory_stats    # round 3's HBM placement study (NOTES.md)

bench.py loads it for `roofline.flat_read` when the library is present; nothing in halo_amd/ does.  The library is built in-tree
by build() (hipcc cross-compiles for gfx950 without a GPU; __graft_entry__.build() calls it) and travels to the GPU box.
"""
import ctypes as C
import math
import os
import shutil
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "halo_probe.hip")
SO = os.path.join(HERE, "libhalo_probe.so")
_handle = None


def build(force=False):
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(SRC):
        return SO
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = SO + ".tmp.%d" % os.getpid()
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", SRC, "-o", tmp],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on tools/halo_probe.hip:\n" + r.stdout)
    os.replace(tmp, SO)
    return SO


def lib():
    """The probe library, typed; loads torch's HIP runtime first (one runtime per process, as halo_amd/_lib.py does)."""
    global _handle
    if _handle is None:
        from halo_amd import _lib
        _lib._preload_torch_hip_runtime()
        h = C.CDLL(build())
        vp, sz = C.c_void_p, C.c_size_t
        h.halo_probe_last_error.restype = C.c_char_p
        h.halo_hbm_read_probe.restype = C.c_int
        h.halo_hbm_read_probe.argtypes = [vp, sz, vp, C.c_int, vp]
        h.halo_hbm_walk_probe.restype = C.c_int
        h.halo_hbm_walk_probe.argtypes = [vp, sz, sz, C.c_int, vp, vp]
        h.halo_pool_alloc.restype = vp
        h.halo_pool_alloc.argtypes = [sz, C.c_int, vp]
        h.halo_pool_free.restype = None
        h.halo_pool_free.argtypes = [vp, sz, C.c_int, vp]
        h.halo_pool_alloc_stats.restype = C.c_int
        h.halo_pool_alloc_stats.argtypes = [C.POINTER(C.c_uint64)]
        _handle = h
    return _handle


class ProbeError(RuntimeError):
    pass


def check(rc, what="probe"):
    if rc != 0:
        raise ProbeError("%s failed (%d): %s" % (what, rc, lib().halo_probe_last_error().decode("utf-8", "replace")))


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def read_probe(t, nbytes=None, sink=None, offset=0):
    """Enqueue one flat non-temporal streaming read of `t`'s bytes on the current stream."""
    nb = t.numel() * t.element_size() if nbytes is None else int(nbytes)
    check(lib().halo_hbm_read_probe(t.data_ptr() + offset, nb, None if sink is None else sink.data_ptr(), 0, _stream(t.device)),
          "halo_hbm_read_probe")


def walk_probe(t, nbytes, plane_bytes, planes, out, offset=0):
    check(lib().halo_hbm_walk_probe(t.data_ptr() + offset, int(nbytes), int(plane_bytes), int(planes), out.data_ptr(), _stream(t.device)),
          "halo_hbm_walk_probe")


def flat_read_gbps(t, reps=5):
    """{GB/s, avg_ms, bytes} of a flat non-temporal read of the contiguous tensor `t`, alone on the device (first pass = warm-up)."""
    assert t.is_contiguous()
    dev = t.device
    nb = t.numel() * t.element_size()
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 2)]
    torch.cuda.synchronize(dev)
    for i in range(reps + 2):
        read_probe(t, nb, sink)
        ev[i].record()
    torch.cuda.synchronize(dev)
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(1, reps + 1)]
    avg = sum(ms) / len(ms)
    return {"GB/s": round(nb / (avg * 1e-3) / 1e9, 1), "avg_ms": round(avg, 4), "bytes": nb}


# ---- diagnostics: where a streamed pool lives in HBM.  On MI355X the scorer's walk over C planes 16 MiB apart runs ~7 % slower
# over some 16-48 GiB stretches of a large allocation than over the rest (6.35 vs 6.83 TB/s per 16 GiB window), while a flat read
# is equally fast everywhere; that is the "plateau" a run lands on (NOTES.md, tools/probe_placement.py).  The functions below
# allocate a tensor in one physically contiguous range and time the walk per window.  They are measurement aids: no placement
# policy is built on them (the windows do not predict the full scoring call well enough), and bench.py takes the memory
# torch's allocator hands it.
_TYPESTR = {torch.float64: "<f8", torch.float32: "<f4", torch.int64: "<i8", torch.int32: "<i4", torch.uint8: "|u1"}


class _RawBlock(object):
    """One halo_pool_alloc allocation (physically contiguous when the driver can), exposed through __cuda_array_interface__;
    freed when the last tensor viewing it is gone."""

    def __init__(self, nbytes, shape, dtype, index):
        self._lib, self.nbytes, self.index = lib(), int(nbytes), int(index)
        self.ptr = self._lib.halo_pool_alloc(self.nbytes, self.index, None)
        if not self.ptr:
            raise torch.cuda.OutOfMemoryError("halo_pool_alloc: %d bytes on device %d" % (self.nbytes, self.index))
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": _TYPESTR[dtype], "data": (self.ptr, False), "version": 2,
                                         "strides": None}

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.halo_pool_free(self.ptr, self.nbytes, self.index, None)
            self.ptr = None


def alloc_contiguous(shape, dtype, device):
    """An uninitialised tensor in its own physically contiguous HBM range (hipDeviceMallocContiguous through halo_pool_alloc;
    plain hipMalloc, counted in contiguous_memory_stats(), when no such range is free).  Not from torch's caching allocator:
    the memory goes back to the driver when the tensor and its views are gone."""
    device = torch.device(device)
    shape = tuple(int(v) for v in shape)
    nbytes = max(16, math.prod(shape) * torch.empty((), dtype=dtype).element_size())
    torch.cuda.synchronize(device)
    with torch.cuda.device(device):
        torch.zeros(1, device=device)                                   # HIP context up before the raw allocation
        return torch.as_tensor(_RawBlock(nbytes, shape, dtype, device.index or 0), device=device)


def contiguous_memory_stats():
    """{contiguous_bytes, fallback_bytes (hipMalloc: no contiguous range of that size was free), live, failed} so far."""
    out = (C.c_uint64 * 4)()
    check(lib().halo_pool_alloc_stats(out), "halo_pool_alloc_stats")
    return {"contiguous_bytes": int(out[0]), "fallback_bytes": int(out[1]), "live": int(out[2]), "failed": int(out[3])}


def probe_streaming(t, planes, plane_bytes, window_bytes=16 << 30, reps=3):
    """Per window of the contiguous tensor `t`: GB/s of the scorer's plane walk (halo_hbm_walk_probe: groups of `planes` planes of
    `plane_bytes`) and of a flat read (halo_hbm_read_probe).  [(byte offset, bytes, walk GB/s, flat GB/s)]."""
    dev = t.device
    assert t.is_contiguous()
    group = int(planes) * int(plane_bytes)
    total = t.numel() * t.element_size()
    per = max(1, int(window_bytes) // group) * group
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty((per // int(planes),), dtype=torch.uint8, device=dev)
    out, off = [], 0
    while off + group <= total:
        nb = min(per, (total - off) // group * group)
        rates = []
        for fn in (lambda: walk_probe(t, nb, plane_bytes, planes, scratch, off), lambda: read_probe(t, nb, sink, off)):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            fn()
            ev[0].record()
            for _ in range(reps):
                fn()
            ev[1].record()
            torch.cuda.synchronize(dev)
            rates.append(nb * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9)
        out.append((off, nb, rates[0], rates[1]))
        off += nb
    return out
