"""ctypes binding of libhalo_host.so (halo_amd/csrc/halo_host.c): plain-C host helpers of the persistence step."""
import ctypes as C
import threading

from . import _build

ABI_VERSION = 4
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
                h.halo_crc32.restype = C.c_uint32
                h.halo_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
                h.halo_compose_mask.restype = C.c_int
                h.halo_compose_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                                C.c_int64, C.c_int64]
                h.halo_write_indicator.restype = C.c_int
                h.halo_write_indicator.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                   C.c_size_t, C.c_void_p, C.c_void_p]
                h.halo_retire_image.restype = C.c_int
                h.halo_retire_image.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                                C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                                C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
                h.halo_compose_indicators.restype = C.c_int
                h.halo_compose_indicators.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                      C.c_int64, C.c_int64, C.c_int64]
                _handle = h
    return _handle


def png_gray8_write(path, arr):
    """Write the (H, W) uint8 numpy array `arr` (rows contiguous) as an 8-bit greyscale PNG.  The call releases the GIL."""
    import os
    assert arr.ndim == 2 and arr.dtype.itemsize == 1 and arr.strides[1] == 1 and arr.strides[0] >= arr.shape[1] and arr.size
    rc = lib().halo_png_gray8_write(os.fsencode(path), arr.ctypes.data, arr.shape[0], arr.shape[1], arr.strides[0])
    if rc != 0:
        raise OSError("halo_png_gray8_write(%r) failed (%d)" % (path, rc))


def compose_indicators(prior_active, prior_selected, picks, k, radius, mask_radius):
    """-> (active, selected) (H, W) bool arrays after the round: the maps the image entered it with plus the windows of the first k
    picks (halo_compose_indicators, build.py:56-59).  GIL-free."""
    import numpy as np
    H, W = prior_active.shape
    pa, ps = np.ascontiguousarray(prior_active).view(np.uint8), np.ascontiguousarray(prior_selected).view(np.uint8)
    pk = np.ascontiguousarray(picks, dtype=np.float64)
    a, s = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
    rc = lib().halo_compose_indicators(a.ctypes.data, s.ctypes.data, pa.ctypes.data, ps.ctypes.data, H, W, pk.ctypes.data, int(k),
                                       int(radius), int(mask_radius))
    if rc != 0:
        raise ValueError("halo_compose_indicators failed (%d)" % rc)
    return a.view(np.bool_), s.view(np.bool_)


def retire_image(path_png, path_indicator, origin_mask, origin_label, picks, k, radius, active, selected, template, compose_mask_radius=-1):
    """One image's two files in one GIL-free call (halo_retire_image): origin_mask / origin_label (H, W) contiguous integer
    numpy arrays, picks (>= k, 3) contiguous float64, active / selected (H, W) contiguous bool / uint8 arrays (pinned staging
    memory is fine), template: the shape's _IndicatorTemplate (raw bytes + field offsets) or None to skip the indicator.
    compose_mask_radius >= 0: active / selected are the maps the image ENTERED the round with; the round's windows (that mask
    radius; `radius` for selected) are added on the way."""
    import os
    H, W = origin_mask.shape
    for a in (origin_mask, origin_label, active, selected):
        assert a.shape == (H, W) and a.flags["C_CONTIGUOUS"]
    assert picks.dtype.itemsize == 8 and picks.flags["C_CONTIGUOUS"] and picks.shape[0] >= k and (k == 0 or picks.shape[1] == 3)
    if template is not None:
        tpl, tlen, off_a, off_s, fa, fs = template.raw.ctypes.data, template.raw.size, template.off["active"], template.off["selected"], \
            template.crc["active"].ctypes.data, template.crc["selected"].ctypes.data
        pind = os.fsencode(path_indicator)
    else:
        tpl, tlen, off_a, off_s, fa, fs, pind = None, 0, 0, 0, None, None, None
    rc = lib().halo_retire_image(os.fsencode(path_png), pind, origin_mask.ctypes.data, origin_mask.dtype.itemsize, origin_label.ctypes.data,
                                 origin_label.dtype.itemsize, H, W, picks.ctypes.data, int(k), int(radius), active.ctypes.data,
                                 selected.ctypes.data, int(compose_mask_radius), tpl, tlen, off_a, off_s, fa, fs)
    if rc != 0:
        raise OSError("halo_retire_image(%r, %r) failed (%d)" % (path_png, path_indicator, rc))
