"""TEST INFRASTRUCTURE -- numpy/ctypes front end of the CPU oracle (oracle/halo_oracle.c).

Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Mirrors the reference's call surface for the hot
path so the parity tests read like calls into the reference:

    expmap / logmap / dist0 / dist        core/utils/hyperbolic.py:28-83   (geoopt formulas)
    hypermlr                              core/utils/hyperbolic.py:120-184
    bilinear                              F.interpolate(align_corners=True), core/active/build.py:123-135
    floating_region_score                 core/active/floating_region.py:129-217
    select_pixels_to_label                core/active/build.py:27-64
    region_selection                      core/active/build.py:71-186
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhalo_oracle.so")

UNC = {"entropy": 0, "pixel_entropy": 1, "oracle_acc": 2}   # anything else: zeros (floating_region.py:84-87)
PUR = {"ripu": 0, "oracle_ripu": 1, "hyper": 2, "none": 3, "radius": 4, "euc_norm": 5}
F32, F64 = 0, 1
_i64, _dbl, _int, _vp = C.c_longlong, C.c_double, C.c_int, C.c_void_p


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("halo_oracle.c", "halo_oracle_math.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        import fcntl
        os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
        with open(os.path.join(_HERE, "_build", ".lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


_lib = None


def usable_cpus():
    """Cores this process may really use: min(visible, affinity, cgroup quota).  The GPU boxes show 256 CPUs under
    a 16-core quota; an OpenMP team as wide as the machine then spends most of its time throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.environ.get("HALO_ORACLE_LIB") or build())      # (the variable: a sanitizer build, tests/test_sanitizers.py)
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(usable_cpus())
        except OSError:
            pass
        _lib.halo_o_expf.restype = C.c_float
        _lib.halo_o_expf.argtypes = [C.c_float]
        _lib.halo_o_logf.restype = C.c_float
        _lib.halo_o_logf.argtypes = [C.c_float]
        _lib.halo_o_log.restype = _dbl
        _lib.halo_o_log.argtypes = [_dbl]
        _lib.halo_o_log_cr.restype = _dbl
        _lib.halo_o_log_cr.argtypes = [_dbl]
        _lib.halo_o_select.restype = _i64
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _dt(a):
    if a.dtype == np.float64:
        return F64
    if a.dtype == np.float32:
        return F32
    raise TypeError(a.dtype)


def _split(shape, dim):
    dim = dim % len(shape)
    outer = int(np.prod(shape[:dim], dtype=np.int64))
    inner = int(np.prod(shape[dim + 1:], dtype=np.int64))
    return outer, shape[dim], inner


def expmap(x, c=1.0, dim=-1):
    x = np.ascontiguousarray(x)
    y = np.empty(x.shape, np.float64)
    o, ch, i = _split(x.shape, dim)
    lib().halo_o_expmap0_project(_p(x), _int(_dt(x)), _p(y), _i64(o), _i64(ch), _i64(i), _dbl(c))
    return y


def logmap(x, c=1.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    o, ch, i = _split(x.shape, -1)
    lib().halo_o_logmap0_project(_p(x), _p(y), _i64(o), _i64(ch), _i64(i), _dbl(c))
    return y


def dist0(x, c=1.0, dim=-1):
    x = np.ascontiguousarray(x)
    o, ch, i = _split(x.shape, dim)
    d = dim % x.ndim
    out = np.empty(x.shape[:d] + x.shape[d + 1:], x.dtype)
    lib().halo_o_dist0(_p(x), _int(_dt(x)), _p(out), _i64(o), _i64(ch), _i64(i), _dbl(c))
    return out


def dist(x, y, c=1.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(x.shape[:-1], np.float64)
    lib().halo_o_dist(_p(x), _p(y), _p(out), _i64(out.size), _i64(x.shape[-1]), _dbl(c))
    return out


def hypermlr(x, P, A, c=1.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    B, Cc, h, w = x.shape
    O = P.shape[0]
    out = np.empty((B, O, h, w), np.float64)
    lib().halo_o_hypermlr(_p(x), _p(P), _p(A), _p(out), _i64(B), _i64(Cc), _i64(O), _i64(h * w), _dbl(c))
    return out


def bilinear(x, size):
    x = np.ascontiguousarray(x)
    H, W = int(size[0]), int(size[1])
    h, w = x.shape[-2:]
    out = np.empty(x.shape[:-2] + (H, W), x.dtype)
    planes = int(np.prod(x.shape[:-2], dtype=np.int64))
    fn = lib().halo_o_bilinear_f64 if x.dtype == np.float64 else lib().halo_o_bilinear_f32
    fn(_p(x), _p(out), _i64(planes), _i64(h), _i64(w), _i64(H), _i64(W))
    return out


def gram_radius(feat_lr, size, mode="radius", c=1.0):
    """Per-pixel radius ('radius') or norm ('euc_norm') of the bilinear upsampling of a float64 low-res embedding
    (C,h,w) to `size`, evaluated through the Gram form -- the CPU twin of the product's 'gram' low-res mode
    (halo_o_gram_radius)."""
    f = np.ascontiguousarray(feat_lr, dtype=np.float64)
    if f.ndim == 4:
        f = f[0]
    Cc, h, w = f.shape
    H, W = int(size[0]), int(size[1])
    out = np.empty((H, W), np.float64)
    lib().halo_o_gram_radius(_p(f), _i64(Cc), _i64(h), _i64(w), _i64(H), _i64(W), _int(0 if mode == "radius" else 1), _dbl(c), _p(out))
    return out


PAD = {"zeros": 0, "reflect": 1, "replicate": 2, "circular": 3}


def floating_region_score(logit, decoder_out=None, unc_type=None, pur_type=None, normalize=False,
                          ground_truth=None, size=3, purity_type=None, K=100, c=1.0, impurity_raw=None, padding_mode="zeros"):
    """FloatingRegionScore(in_channels=O, padding_mode=padding_mode, size=size, purity_type=purity_type, K=K)(logit, ...).
    impurity_raw: (H,W) float64 map of per-pixel radii / norms to use INSTEAD of reducing decoder_out (gram_radius)."""
    logit = np.ascontiguousarray(logit, dtype=np.float32)
    if logit.ndim == 4:
        logit = logit[0]
    O, H, W = logit.shape
    if pur_type not in PUR:
        raise NotImplementedError("Error: purity type '{}' not implemented".format(pur_type))
    feat = None
    Cc = 0
    fdt = F64
    if pur_type in ("hyper", "radius", "euc_norm") and impurity_raw is not None:
        feat = np.ascontiguousarray(impurity_raw, dtype=np.float64)
        assert feat.shape == (H, W)
        Cc, fdt = 0, F64
    elif pur_type in ("hyper", "radius", "euc_norm"):
        feat = np.ascontiguousarray(decoder_out)
        if feat.ndim == 4:
            feat = feat[0]
        Cc = feat.shape[0]
        fdt = _dt(feat)
    gt = None if ground_truth is None else np.ascontiguousarray(ground_truth, dtype=np.int64)
    f64out = pur_type in ("radius", "euc_norm") and fdt == F64
    odt = np.float64 if f64out else np.float32
    score = np.empty((H, W), odt)
    imp = np.empty((H, W), odt)
    unc = np.empty((H, W), np.float32)
    sdt = _int(0)
    pk = 3 if purity_type == "hyper" else size
    rc = lib().halo_o_floating_region_score(
        _p(logit), _p(feat), _int(fdt), _p(gt), _i64(O), _i64(Cc), _i64(H), _i64(W),
        _int(UNC.get(unc_type, 3)), _int(PUR[pur_type]), _int((1 if normalize else 0) | (PAD[padding_mode] << 8)),
        _int(size), _int(pk), _i64(K), _dbl(c), _p(score), _p(imp), _p(unc), C.byref(sdt))
    assert rc == 0
    return score, imp, unc


def _map_f32(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fn(_p(x), _p(y), _i64(x.size))
    return y


def expf(x):
    """The contract's float32 exp (Sleef expf_u10 = ATen's Vectorized<float>::exp) elementwise."""
    return _map_f32(lib().halo_o_expf_v, x)


def logf(x):
    """The contract's float32 log (correctly rounded) elementwise."""
    return _map_f32(lib().halo_o_logf_v, x)


def sum_dim0(t):
    """torch.sum(t, dim=0) of a float32 (n, ...) array in ATen's cascade order."""
    t = np.ascontiguousarray(t, dtype=np.float32)
    out = np.empty(t.shape[1:], np.float32)
    lib().halo_o_sum_dim0(_p(t), _i64(t.shape[0]), _i64(out.size), _p(out))
    return out


def box_sum(x, size=3, padding_mode="zeros"):
    """entropy_conv (floating_region.py:42-51): k x k all-ones box sum of an (H,W) float32 map."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().halo_o_box_sum(_p(x), _p(out), _i64(x.shape[0]), _i64(x.shape[1]), _int(size), _int(PAD[padding_mode]))
    return out


def softmax(logit):
    logit = np.ascontiguousarray(logit, dtype=np.float32)
    O, H, W = logit.shape
    p = np.empty_like(logit)
    lib().halo_o_softmax(_p(logit), _i64(O), _i64(H), _i64(W), _p(p))
    return p


def uncertainty_from_probs(p, unc_type, ground_truth=None, size=3, do_box=True, padding_mode="zeros"):
    """compute_region_uncertainty / compute_pixel_entropy (floating_region.py:70-92,123-127) -> (1,1,H,W)."""
    p = np.ascontiguousarray(p, dtype=np.float32)
    O, H, W = p.shape
    gt = None if ground_truth is None else np.ascontiguousarray(ground_truth, dtype=np.int64)
    out = np.empty((1, 1, H, W), np.float32)
    lib().halo_o_uncertainty_from_probs(_p(p), _p(gt), _i64(O), _i64(H), _i64(W), _int(UNC.get(unc_type, 3)),
                                        _int(size), _int((1 if do_box else 0) | (PAD[padding_mode] << 8)), _p(out))
    return out


def region_impurity(predict, K, size=3, padding_mode="zeros"):
    """compute_region_impurity (floating_region.py:112-121) -> (imp, count), each (1,1,H,W)."""
    pred = np.ascontiguousarray(predict, dtype=np.int64)
    H, W = pred.shape
    imp = np.empty((1, 1, H, W), np.float32)
    cnt = np.empty((1, 1, H, W), np.float32)
    lib().halo_o_region_impurity(_p(pred), _i64(K), _int(size), _i64(H), _i64(W), _p(imp), _p(cnt), _int(PAD[padding_mode]))
    return imp, cnt


def quantize_uncert_map(decoder_out, K, c=1.0):
    """quantize_uncert_map (floating_region.py:94-110) -> (H,W) int64."""
    feat = np.ascontiguousarray(decoder_out)
    if feat.ndim == 4:
        feat = feat[0]
    Cc, H, W = feat.shape
    pred = np.empty((H, W), np.int64)
    lib().halo_o_quantize(_p(feat), _int(_dt(feat)), _i64(Cc), _i64(H), _i64(W), _i64(K), _dbl(c), _p(pred))
    return pred


def select_pixels_to_label(score, active_regions, active_radius, mask_radius, active, selected,
                           active_mask, ground_truth, return_picks=False):
    """In-place on its numpy arguments like the reference (build.py:27-64)."""
    assert score.flags.c_contiguous and score.dtype in (np.float32, np.float64)
    H, W = score.shape
    a8 = active.view(np.uint8)
    s8 = selected.view(np.uint8)
    assert active_mask.dtype == np.int64 and ground_truth.dtype == np.int64
    picks = np.zeros((max(int(active_regions), 1), 3), np.float64)
    n = lib().halo_o_select(_p(score), _int(_dt(score)), _i64(H), _i64(W), _i64(int(active_regions)),
                            _i64(active_radius), _i64(mask_radius), _p(a8), _p(s8), _p(active_mask),
                            _p(np.ascontiguousarray(ground_truth)), _p(picks))
    if return_picks:
        return score, active, selected, active_mask, picks[:n]
    return score, active, selected, active_mask


def region_selection(cfg, images, c=None, lowres_mode="exact"):
    """RegionSelection's per-image body (build.py:113-166) on pre-computed low-res head
    outputs.  `images`: list of dicts with logit_lr (1,O,h,w) f32, embed_lr (1,C,h,w) f64,
    origin_mask, origin_label (H,W) i64, active, selected (H,W) bool.  Returns a list of
    (active_mask uint8, active, selected, picks).  lowres_mode 'gram': the radius / norm of a float64
    embedding through gram_radius (the product's 'gram' mode) instead of upsample-then-reduce."""
    per_region = (2 * cfg.ACTIVE.RADIUS_K + 1) ** 2
    budget = cfg.ACTIVE.BUDGET / len(cfg.ACTIVE.SELECT_ITER)
    unc, pur = cfg.ACTIVE.UNCERTAINTY, cfg.ACTIVE.PURITY
    c = cfg.MODEL.CURVATURE if c is None else c
    out = []
    for im in images:
        H, W = im["origin_label"].shape
        logit = bilinear(im["logit_lr"], (H, W))
        dec = im["embed_lr"]
        raw = None
        if lowres_mode == "gram" and pur in ("hyper", "radius", "euc_norm") and np.asarray(dec).dtype == np.float64:
            raw = gram_radius(dec, (H, W), "euc_norm" if pur == "euc_norm" else "radius", c)
        elif unc in ("certainty", "hyperbolic") or pur in ("hyper", "radius", "euc_norm") or \
                (unc == "none" and cfg.MODEL.HYPER):
            dec = bilinear(dec, (H, W))
        score, _, _ = floating_region_score(
            logit, decoder_out=dec, unc_type=unc, pur_type=pur, normalize=cfg.ACTIVE.NORMALIZE,
            ground_truth=im["origin_label"], size=2 * cfg.ACTIVE.RADIUS_K + 1, purity_type=pur,
            K=cfg.ACTIVE.K, c=c, impurity_raw=raw)
        active = im["active"].copy()
        selected = im["selected"].copy()
        amask = im["origin_mask"].copy()
        score[active] = -np.inf
        n = math.ceil(H * W * budget / per_region)
        _, _, _, _, picks = select_pixels_to_label(
            score, n, cfg.ACTIVE.RADIUS_K, cfg.ACTIVE.MASK_RADIUS_K, active, selected, amask,
            im["origin_label"], return_picks=True)
        out.append((amask.astype(np.uint8), active, selected, picks))
    return out
