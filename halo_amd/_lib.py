"""ctypes binding of libhalo_hip.so (C ABI: include/halo_hip.h).

This is the only place that touches the shared library.  Nothing here falls back to a CPU
implementation: a missing library, a missing symbol or a non-ROCm tensor raises.
"""
import ctypes as C
import os
import threading

from . import _build

F32, F64 = 0, 1
UNC = {"entropy": 0, "pixel_entropy": 1, "oracle_acc": 2}          # everything else -> zeros (3)
UNC_ZEROS = 3
PUR = {"ripu": 0, "oracle_ripu": 1, "hyper": 2, "none": 3, "radius": 4, "euc_norm": 5}
E_UNSUPPORTED = -2
SELECT = {"auto": 0, "serial": 1, "binned": 2}                      # HALO_SELECT_* of include/halo_hip.h
SWEEP_REASONS = ("done", "bad_values", "bin_overflow", "survivors", "exhausted", "not_run")      # HALO_SWEEP_*
PAD = {"zeros": 0, "reflect": 1, "replicate": 2, "circular": 3}     # HALO_PAD_*: nn.Conv2d's padding_mode values
FLAG_NORMALIZE = 1


def score_flags(normalize, padding_mode="zeros"):
    """the flags word of the halo_score_maps* calls: bit 0 normalise, bits 8-9 the padding mode"""
    return (FLAG_NORMALIZE if normalize else 0) | (PAD[padding_mode] << 8)

_i64, _dbl, _int, _vp, _sz = C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/halo_hip.h declares
SIGNATURES = {
    "halo_version": (_int, []),
    "halo_last_error": (C.c_char_p, []),
    "halo_expmap0_project": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _dbl, _vp]),
    "halo_logmap0_project": (_int, [_vp, _vp, _i64, _i64, _i64, _dbl, _vp]),
    "halo_dist0": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _dbl, _vp]),
    "halo_pdist": (_int, [_vp, _vp, _vp, _i64, _i64, _dbl, _vp]),
    "halo_hypermlr_workspace_bytes": (_sz, [_i64, _i64]),
    "halo_hypermlr_logits": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _dbl, _vp, _sz, _vp]),
    "halo_head_tail": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _dbl, _vp, _sz, _vp]),
    "halo_expmap0_project_bwd": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _i64, _dbl, _vp]),
    "halo_hypermlr_bwd_terms": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _sz, _vp]),
    "halo_hypermlr_backward_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "halo_hypermlr_backward": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "halo_bilinear_upsample": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _vp]),
    "halo_score_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "halo_score_maps": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int,
                               _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "halo_score_maps_timed": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int,
                                     _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "halo_score_maps_split": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int,
                                     _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "halo_score_lr_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "halo_score_maps_lr": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                  _int, _int, _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "halo_score_lr_gram_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _i64]),
    "halo_score_maps_lr_gram": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                       _int, _int, _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "halo_score_maps_lr_timed": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                        _int, _int, _int, _int, _int, _i64, _dbl, _vp, _vp, _vp, _vp, _sz, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "halo_region_uncertainty": (_int, [_vp, _i64, _int, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "halo_region_impurity": (_int, [_vp, _i64, _i64, _i64, _int, _i64, _vp, _vp, _int, _vp]),
    "halo_quantize_radius": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _vp, _vp, _sz, _vp]),
    "halo_loss_workspace_bytes": (_sz, [_i64]),
    "halo_negative_learning_fwd": (_int, [_vp, _i64, _dbl, _vp, _vp, _sz, _vp]),
    "halo_negative_learning_bwd": (_int, [_vp, _i64, _dbl, _vp, _vp, _vp, _vp]),
    "halo_local_consistent_fwd": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "halo_local_consistent_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "halo_event_create": (_vp, []),
    "halo_event_record": (_int, [_vp, _vp]),
    "halo_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "halo_event_destroy": (_int, [_vp]),
    "halo_pack_pick_tables": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "halo_reset_round_state": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "halo_undo_picks": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "halo_device_identity": (_int, [_int, C.c_char_p, _sz]),
    "halo_select_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "halo_score_range_bytes": (_sz, [_i64]),
    "halo_score_range": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp]),
    "halo_greedy_select_ranged": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _sz, _int, _vp, _vp]),
    "halo_greedy_select_ex": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _sz, _int, _vp, _vp, _vp]),
    "halo_greedy_select": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _sz, _int, _vp]),
}

# must equal HALO_ABI_VERSION of include/halo_hip.h; bumped whenever an exported signature changes, so a stale
# library with the same symbol names but older argument lists is refused instead of being called with shifted arguments
ABI_VERSION = 8

_lock = threading.Lock()
_handle = None


class HaloHipError(RuntimeError):
    pass


class HaloUnsupported(HaloHipError):
    """The kernel declined this configuration (HALO_E_UNSUPPORTED); the caller may take another HIP route."""


def library_path():
    return _build.SO


def _preload_torch_hip_runtime():
    """Device pointers and streams come from torch, so the kernels must be launched through the
    SAME HIP runtime instance torch uses.  PyTorch-ROCm wheels bundle their own libamdhip64.so;
    loading it first (RTLD_GLOBAL) makes libhalo_hip.so's NEEDED libamdhip64.so.N resolve to it
    instead of a second copy from /opt/rocm (two runtimes = "no ROCm-capable device")."""
    import torch
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("libamdhip64.so",):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def hip_runtime():
    """ctypes handle of the HIP runtime torch itself runs on (its bundled libamdhip64.so where the wheel has one): for the few
    plain runtime calls the host code makes directly (a stream outside torch's pool, core/active/build.py:_capture_stream)."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    return C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so", mode=C.RTLD_GLOBAL)


def lib():
    """Load (building first if the in-tree .so is missing or stale) and type the library."""
    global _handle
    if _handle is not None:
        return _handle
    with _lock:
        if _handle is not None:
            return _handle
        try:
            # HALO_LIB_PATH: load a specific build of the library (A/B timing of kernel variants)
            path = os.environ.get("HALO_LIB_PATH") or _build.build()
        except Exception as exc:  # no hipcc: a prebuilt in-tree .so is fine if it is current, never a stale one
            if not os.path.exists(_build.SO):
                raise HaloHipError("libhalo_hip.so is missing and could not be built: %s" % exc) from exc
            if _build.is_stale() and not os.environ.get("HALO_ALLOW_STALE_LIB"):
                raise HaloHipError("libhalo_hip.so is older than its sources and could not be rebuilt (%s); "
                                   "set HALO_ALLOW_STALE_LIB=1 to load it anyway" % exc) from exc
            path = _build.SO
        _preload_torch_hip_runtime()
        h = C.CDLL(path)
        ns = _Library()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name, None)
            if fn is None:
                raise HaloHipError("%s does not export %s" % (path, name))
            fn.restype = res
            fn.argtypes = args
            setattr(ns, name, _on_stream_device(fn) if res is _int and args and _vp in args else fn)
        got = h.halo_version()
        if got != ABI_VERSION:
            raise HaloHipError("%s has ABI version %d, this package binds version %d: rebuild it "
                               "(python -m halo_amd._build --force)" % (path, got, ABI_VERSION))
        ns._cdll = h
        _handle = ns
    return _handle


class _Library(object):
    """Typed entry points of libhalo_hip.so (attributes named like the C functions)."""


class StreamPtr(C.c_void_p):
    """hipStream_t as void* that remembers which device it belongs to."""
    device_index = None


def _on_stream_device(fn):
    """Launch on the device that owns the stream: kernels go to the CURRENT HIP device, so tensors on
    cuda:N with cuda:M current would otherwise be launched on M's null stream against N's pointers."""
    def call(*args):
        import torch
        for a in reversed(args):
            if isinstance(a, StreamPtr):
                idx = a.device_index
                if idx is not None and idx != torch.cuda.current_device():
                    with torch.cuda.device(idx):
                        return fn(*args)
                break
        return fn(*args)
    call.__name__ = getattr(fn, "__name__", "halo_call")
    return call


def check(rc, what=""):
    if rc == 0:
        return
    msg = lib().halo_last_error().decode("utf-8", "replace")
    if rc == E_UNSUPPORTED and "not implemented" in msg:
        raise NotImplementedError(msg)
    if rc == E_UNSUPPORTED:
        raise HaloUnsupported("%s: %s" % (what or "halo call", msg))
    raise HaloHipError("%s failed (%d): %s" % (what or "halo call", rc, msg))


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError("halo_amd: expected float32/float64 tensor, got %s" % t.dtype)


def require_device(*tensors):
    """All tensors must live on one ROCm device.  No CPU path exists."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HaloHipError(
                "halo_amd runs on ROCm devices only (got a %s tensor); there is no CPU fallback" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HaloHipError("halo_amd: tensors on different devices (%s vs %s)" % (dev, t.device))
    return dev


def stream_ptr(device=None):
    import torch
    st = torch.cuda.current_stream(device)
    p = StreamPtr(st.cuda_stream)
    p.device_index = st.device.index
    return p


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())
