"""ctypes binding of libhalo_host.so (halo_amd/csrc/halo_host.c): plain-C host helpers of the persistence step."""
import ctypes as C
import threading

from . import _build

ABI_VERSION = 1
_lock = threading.Lock()
_handle = None


def lib():
    global _handle
    if _handle is None:
        with _lock:
            if _handle is None:
                h = C.CDLL(_build.build_host())
                h.halo_host_version.restype = C.c_int
                if h.halo_host_version() != ABI_VERSION:
                    raise RuntimeError("libhalo_host.so has ABI %d, this package binds %d: python -m halo_amd._build --force"
                                       % (h.halo_host_version(), ABI_VERSION))
                h.halo_png_gray8_bound.restype = C.c_size_t
                h.halo_png_gray8_bound.argtypes = [C.c_int64, C.c_int64]
                h.halo_png_gray8_encode.restype = C.c_size_t
                h.halo_png_gray8_encode.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t]
                h.halo_png_gray8_write.restype = C.c_int
                h.halo_png_gray8_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
                _handle = h
    return _handle


def png_gray8_write(path, arr):
    """Write the (H, W) uint8 numpy array `arr` (rows contiguous) as an 8-bit greyscale PNG.  The call releases the GIL."""
    import os
    assert arr.ndim == 2 and arr.dtype.itemsize == 1 and arr.strides[1] == 1 and arr.strides[0] >= arr.shape[1] and arr.size
    rc = lib().halo_png_gray8_write(os.fsencode(path), arr.ctypes.data, arr.shape[0], arr.shape[1], arr.strides[0])
    if rc != 0:
        raise OSError("halo_png_gray8_write(%r) failed (%d)" % (path, rc))
