"""FloatingRegionScore on HIP kernels -- host mirror of core/active/floating_region.py.

Same constructor, same `forward(logit, decoder_out, unc_type, pur_type, normalize, ground_truth)
-> (score, region_impurity, prediction_uncertainty)`, same dtypes (score float64 when the purity is
'radius'/'euc_norm' on a float64 embedding, float32 otherwise), same errors (AssertionError for an
even size, NotImplementedError for an unknown purity type; an unknown uncertainty type is a zero
map, floating_region.py:84-90).  All arithmetic runs in halo_amd/csrc/halo_score.hip; there is no
CPU path and no one-hot / conv tensors are materialised.

`score_maps` is the batched form (B images per launch) the acquisition driver and bench use.
"""
import os

import torch
import torch.nn as nn

from ... import _lib
from ..configs import cfg
from ..utils.hyperbolic import HyperMapper

_WS = {}


import threading as _threading

_ws_tls = _threading.local()


class private_workspaces(object):
    """While active (on this thread), every scratch request is served by a FRESH allocation that is appended to `.held` instead of
    coming from the per-stream cache: a launch group recorded into a HIP graph must own the scratch it points to -- the cache
    re-allocates a stream's buffer when a later, larger call needs more (and release_workspaces() drops it), which would leave the
    recording with a dangling pointer (round 5: a memory access fault after a batch-4 group had grown the stream's workspace)."""

    def __enter__(self):
        self.held = []
        self._prev = getattr(_ws_tls, "held", None)
        _ws_tls.held = self.held
        return self

    def __exit__(self, *exc):
        _ws_tls.held = self._prev
        return False


def _workspace(dev, nbytes, tag):
    """Stream-ordered scratch, cached per (device, stream, tag) and grown on demand."""
    held = getattr(_ws_tls, "held", None)
    if held is not None:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        held.append(buf)
        return buf
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def score_workspace(B, H, W, dev):
    """A scratch buffer of the scorer's size for a caller that keeps several score_maps calls in flight (one per call)."""
    return torch.empty(int(_lib.lib().halo_score_workspace_bytes(int(B), int(H), int(W))), dtype=torch.uint8, device=dev)


def score_dtype(pur_type, decoder_out):
    if pur_type in ("radius", "euc_norm") and decoder_out is not None and decoder_out.dtype == torch.float64:
        return torch.float64
    return torch.float32


def new_score_range(B, dev):
    """Storage for the value-range records of B score maps (opaque, 64 bytes each): filled by score_maps /
    score_maps_lowres (`score_range=`), consumed by greedy_select (`score_range=`), which then skips its own pass over the
    map to find the range."""
    n = _lib.lib().halo_score_range_bytes(1)
    return torch.empty((int(B), int(n)), dtype=torch.uint8, device=dev)


def score_maps(logit, decoder_out=None, unc_type=None, pur_type=None, normalize=False, ground_truth=None,
               size=3, purity_size=None, K=100, c=1.0, active=None, want_maps=True, out=None, events=None, score_range=None,
               tail_stream=None, workspace=None, maps=None, padding_mode="zeros"):
    """Batched FloatingRegionScore.forward.

    logit (B,O,H,W) float32; decoder_out (B,C,H,W) float64|float32; ground_truth (B,H,W) int64;
    active (B,H,W) bool, optional: fuses `score[active] = -inf` (core/active/build.py:146).
    out: optional pre-allocated (B,H,W) score tensor to write into (pipelined callers own their
    buffers); events: optional (start, stop) handles from halo_event_create, recorded around the
    feature-reduction kernel; score_range: optional new_score_range(B, dev) to receive the maps' value ranges.
    tail_stream (pipelined callers): a torch stream that receives everything behind the passes over the inputs
    (halo_score_maps_split; needs `events`, whose stop event is the fork, a `workspace` of its own per call in flight --
    score_workspace(B, H, W, dev) -- and preallocated `maps` = (impurity, uncertainty)); the results are complete on it.
    padding_mode: nn.Conv2d's padding_mode of the two box windows ('zeros' | 'reflect' | 'replicate' | 'circular').
    Returns (score, impurity, uncertainty), each (B,H,W); the last two are None if not want_maps.
    """
    if pur_type not in _lib.PUR:
        raise NotImplementedError("Error: purity type '{}' not implemented".format(pur_type))
    if padding_mode not in _lib.PAD:
        raise ValueError("padding_mode must be one of %s, got %r" % (sorted(_lib.PAD), padding_mode))
    dev = _lib.require_device(logit, decoder_out, ground_truth, active)
    assert logit.dim() == 4, "logit must be (B,O,H,W)"
    if logit.dtype != torch.float32:
        logit = logit.float()
    B, O, H, W = logit.shape
    if logit.stride()[1:] != (H * W, W, 1):
        logit = logit.contiguous()
    need_feat = pur_type in ("hyper", "radius", "euc_norm")
    feat = None
    Cc, fdt, fbs = 0, _lib.F64, 0
    if need_feat:
        if decoder_out is None:
            raise ValueError("decoder_out is required for purity type '%s'" % pur_type)
        feat = decoder_out
        if feat.dtype not in (torch.float32, torch.float64):
            feat = feat.float()
        Cc = feat.shape[1]
        assert feat.shape[0] == B and feat.shape[2:] == (H, W), "decoder_out shape mismatch"
        if feat.stride()[1:] != (H * W, W, 1):
            feat = feat.contiguous()
        fdt, fbs = _lib.dtype_code(feat), feat.stride(0)
    need_gt = unc_type == "oracle_acc" or pur_type == "oracle_ripu"
    gt = None
    if need_gt:
        if ground_truth is None:
            raise ValueError("ground_truth is required for '%s'/'%s'" % (unc_type, pur_type))
        gt = ground_truth.reshape(B, H, W).to(torch.int64).contiguous()
    act = None
    if active is not None:
        act = active.reshape(B, H, W).contiguous()
        act = act.view(torch.uint8) if act.dtype == torch.bool else act.to(torch.uint8)
    odt = score_dtype(pur_type, feat)
    if B == 0 or H == 0 or W == 0:                      # empty batch / empty image: nothing to launch
        e = torch.empty((B, H, W), dtype=odt, device=dev)
        return (e, e.clone(), torch.empty((B, H, W), dtype=torch.float32, device=dev)) if want_maps else (e, None, None)
    if out is not None:
        assert out.shape == (B, H, W) and out.dtype == odt and out.is_contiguous() and out.device == dev
        score = out
    else:
        score = torch.empty((B, H, W), dtype=odt, device=dev)
    if maps is not None:
        imp, unc = maps
        assert imp.shape == (B, H, W) and imp.dtype == odt and imp.is_contiguous() and unc.shape == (B, H, W) and unc.dtype == torch.float32
    else:
        imp = torch.empty((B, H, W), dtype=odt, device=dev) if want_maps else None
        unc = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_maps else None
    L = _lib.lib()
    nws = L.halo_score_workspace_bytes(B, H, W)
    if workspace is not None:
        assert workspace.dtype == torch.uint8 and workspace.numel() >= nws and workspace.device == dev
        ws = workspace
    else:
        ws = _workspace(dev, nws, "score")
    psize = size if purity_size is None else purity_size
    ev0, ev1 = events if events is not None else (None, None)
    if score_range is not None:
        assert score_range.is_contiguous() and score_range.device == dev and score_range.numel() >= L.halo_score_range_bytes(B)
    if tail_stream is not None:
        assert events is not None and workspace is not None and (maps is not None or not want_maps), \
            "tail_stream needs events, a workspace of the call's own and preallocated maps"
        rc = L.halo_score_maps_split(_lib.ptr(logit), logit.stride(0), _lib.ptr(feat), fdt, fbs, _lib.ptr(gt),
                                     _lib.ptr(act), B, O, Cc, H, W, _lib.UNC.get(unc_type, _lib.UNC_ZEROS),
                                     _lib.PUR[pur_type], _lib.score_flags(normalize, padding_mode), int(size), int(psize), int(K), float(c),
                                     _lib.ptr(score), _lib.ptr(imp), _lib.ptr(unc), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr(dev), _lib.C.c_void_p(tail_stream.cuda_stream), ev0, ev1, _lib.ptr(score_range))
        _lib.check(rc, "halo_score_maps_split")
        return score, imp, unc
    rc = L.halo_score_maps_timed(_lib.ptr(logit), logit.stride(0), _lib.ptr(feat), fdt, fbs, _lib.ptr(gt),
                                 _lib.ptr(act), B, O, Cc, H, W, _lib.UNC.get(unc_type, _lib.UNC_ZEROS),
                                 _lib.PUR[pur_type], _lib.score_flags(normalize, padding_mode), int(size), int(psize), int(K), float(c),
                                 _lib.ptr(score), _lib.ptr(imp), _lib.ptr(unc), _lib.ptr(ws), ws.numel(),
                                 _lib.stream_ptr(dev), ev0, ev1, _lib.ptr(score_range))
    _lib.check(rc, "halo_score_maps")
    return score, imp, unc


LOWRES_MODES = ("exact", "gram")


def lowres_mode(mode=None):
    """'exact' (default: the reference's evaluation order, bit-identical to upsample-then-score) or 'gram' (opt-in);
    None reads HALO_LOWRES from the environment"""
    mode = os.environ.get("HALO_LOWRES", "exact") if mode is None else mode
    if mode not in LOWRES_MODES:
        raise ValueError("low-res mode must be one of %s, got %r" % (LOWRES_MODES, mode))
    return mode


def score_maps_lowres(logit_lr, decoder_lr, size, unc_type=None, pur_type=None, normalize=False, ground_truth=None,
                      ksize=3, purity_size=None, K=100, c=1.0, active=None, want_maps=True, mode=None, events=None,
                      score_range=None, padding_mode="zeros"):
    """FloatingRegionScore.forward on the bilinear (align_corners=True) upsampling of LOW-RES sources to
    `size`, without materialising the upsampled tensors -- core/active/build.py:122-144 in one call.
    logit_lr (B,O,hl,wl) float32, decoder_lr (B,C,hf,wf) float64|float32.

    mode 'exact' (default): bit-identical to bilinear_align_corners(...) followed by score_maps(...) -- the reference's
    evaluation order (build.py:133-135 then floating_region.py:129-217).
    mode 'gram' (opt-in: mode="gram" or HALO_LOWRES=gram; float64 embeddings only, float32 ones take 'exact'): the embedding's
    radius / norm through the 10 inner products of each low-res cell's corner vectors -- bit-identical to the
    CPU oracle's statement of that form (oracle.halo_oracle.gram_radius), 2x faster at C = 256, but NOT the reference's
    order: squared norms agree with the exact order to 1.3e-10 relative (observed < 1e-12; pixels whose Gram terms cancel
    are evaluated in the exact order), which can move a radius-bin boundary of the 'hyper' purity or the order of two
    near-tied scores.  Everything else is the same in both modes.  `events`: four to six optional handles from halo_event_create recorded around the logit and embedding passes,
    between the two kernels of the gram route, and behind the tail."""
    if pur_type not in _lib.PUR:
        raise NotImplementedError("Error: purity type '{}' not implemented".format(pur_type))
    mode = lowres_mode(mode)
    dev = _lib.require_device(logit_lr, decoder_lr, ground_truth, active)
    H, W = int(size[0]), int(size[1])
    logit_lr = logit_lr.float().contiguous()
    B, O, hl, wl = logit_lr.shape
    need_feat = pur_type in ("hyper", "radius", "euc_norm")
    feat, Cc, fdt, fbs, hf, wf = None, 0, _lib.F64, 0, 0, 0
    if need_feat:
        if decoder_lr is None:
            raise ValueError("decoder_out is required for purity type '%s'" % pur_type)
        feat = decoder_lr if decoder_lr.dtype in (torch.float32, torch.float64) else decoder_lr.float()
        feat = feat.contiguous()
        assert feat.shape[0] == B
        Cc, hf, wf = feat.shape[1:]
        fdt, fbs = _lib.dtype_code(feat), feat.stride(0)
    need_gt = unc_type == "oracle_acc" or pur_type == "oracle_ripu"
    if need_gt and ground_truth is None:
        raise ValueError("ground_truth is required for '%s'/'%s'" % (unc_type, pur_type))
    gt = ground_truth.reshape(B, H, W).to(torch.int64).contiguous() if need_gt else None
    act = None
    if active is not None:
        act = active.reshape(B, H, W).contiguous()
        act = act.view(torch.uint8) if act.dtype == torch.bool else act.to(torch.uint8)
    odt = score_dtype(pur_type, feat)
    score = torch.empty((B, H, W), dtype=odt, device=dev)
    imp = torch.empty((B, H, W), dtype=odt, device=dev) if want_maps else None
    unc = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_maps else None
    L = _lib.lib()
    gram = mode == "gram" and need_feat and fdt == _lib.F64
    nws = L.halo_score_lr_gram_workspace_bytes(B, O, H, W, hf, wf) if gram else L.halo_score_lr_workspace_bytes(B, O, H, W)
    ws = _workspace(dev, nws, "score")
    psize = ksize if purity_size is None else purity_size
    fn, name = (L.halo_score_maps_lr_gram, "halo_score_maps_lr_gram") if gram else (L.halo_score_maps_lr, "halo_score_maps_lr")
    args = (_lib.ptr(logit_lr), logit_lr.stride(0), hl, wl, _lib.ptr(feat), fdt, fbs, hf, wf,
            _lib.ptr(gt), _lib.ptr(act), B, O, Cc, H, W, _lib.UNC.get(unc_type, _lib.UNC_ZEROS),
            _lib.PUR[pur_type], _lib.score_flags(normalize, padding_mode), int(ksize), int(psize), int(K), float(c),
            _lib.ptr(score), _lib.ptr(imp), _lib.ptr(unc), _lib.ptr(ws), ws.numel(),
            _lib.stream_ptr(dev))
    if events is not None or score_range is not None:
        # events: (logit start, logit stop, embedding start, embedding stop[, embedding mid (gram: between the Gram pass and the
        # radius pass), tail stop]) from halo_event_create, or None each
        if score_range is not None:
            assert score_range.is_contiguous() and score_range.device == dev and score_range.numel() >= L.halo_score_range_bytes(B)
        name = "halo_score_maps_lr_timed"
        ev = tuple(events or ()) + (None,) * 6
        rc = L.halo_score_maps_lr_timed(*(args + (1 if gram else 0,) + ev[:4] + (_lib.ptr(score_range),) + ev[4:6]))
    else:
        rc = fn(*args)
    _lib.check(rc, name)
    return score, imp, unc


class FloatingRegionScore(nn.Module):
    def __init__(self, in_channels=19, padding_mode="zeros", size=33, purity_type=None, K=100):
        """
        purity window: size*size (3*3 over K bins when purity_type == 'hyper')
        entropy window: size*size
        """
        super(FloatingRegionScore, self).__init__()
        self.in_channels = in_channels
        assert size % 2 == 1, "error size"
        if padding_mode not in _lib.PAD:
            # nn.Conv2d's own check (the reference forwards the argument to it, floating_region.py:49,63)
            raise ValueError("padding_mode must be one of ['zeros', 'reflect', 'replicate', 'circular'], but got padding_mode='{}'"
                             .format(padding_mode))
        self.padding_mode = padding_mode
        if purity_type is None:
            purity_type = cfg.ACTIVE.PURITY
        self.size = size
        self.purity_size = size
        self.purity_channels = in_channels
        if purity_type == "hyper":                      # floating_region.py:54-55
            self.K, self.purity_size, self.purity_channels = K, 3, K
        self.mapper = HyperMapper(c=cfg.MODEL.CURVATURE)

    def _check_purity_channels(self, pur_type):
        # the reference's depthwise purity_conv has a fixed channel count; a mismatching one-hot
        # makes F.conv2d raise -- keep that an error instead of inventing a result
        if pur_type in ("ripu", "oracle_ripu") and self.purity_channels != self.in_channels:
            raise RuntimeError("purity window was built for %d bins ('hyper'), got a %d-class prediction"
                               % (self.purity_channels, self.in_channels))
        if pur_type == "hyper" and not hasattr(self, "K"):
            raise AttributeError("'FloatingRegionScore' object has no attribute 'K'")
        if pur_type == "hyper" and self.purity_channels != self.K:
            raise RuntimeError("purity window channel mismatch")

    # ---- helper methods the reference exposes by convention (floating_region.py:70-127) ----
    def _uncertainty(self, x, is_prob, unc_type, ground_truth, do_box):
        dev = _lib.require_device(x, ground_truth)
        x = x.float().contiguous()
        O, H, W = x.shape
        out = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        gt = None if ground_truth is None else ground_truth.to(torch.int64).contiguous()
        ws = _workspace(dev, H * W * 4 + 256, "unc")
        rc = _lib.lib().halo_region_uncertainty(_lib.ptr(x), x.numel(), 1 if is_prob else 0, _lib.ptr(gt), 1, O, H, W,
                                                _lib.UNC.get(unc_type, _lib.UNC_ZEROS), int(self.size),
                                                (1 if do_box else 0) | (_lib.PAD[self.padding_mode] << 8), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr(dev))
        _lib.check(rc, "halo_region_uncertainty")
        return out

    def compute_region_uncertainty(self, unc_type, logit, p, ground_truth=None):
        """p (O,H,W) softmax probabilities -> (1,1,H,W); box-summed unless unc_type == 'none'
        (floating_region.py:70-92)."""
        return self._uncertainty(p, True, unc_type, ground_truth, do_box=(unc_type != "none"))

    def compute_pixel_entropy(self, p):
        """(floating_region.py:123-127)"""
        return self._uncertainty(p, True, "pixel_entropy", None, do_box=False)

    def quantize_uncert_map(self, decoder_out):
        """decoder_out (1,C,H,W) -> (H,W) int64 bins in [0, K-1] (floating_region.py:94-110)."""
        dev = _lib.require_device(decoder_out)
        feat = decoder_out if decoder_out.dtype in (torch.float32, torch.float64) else decoder_out.float()
        feat = feat.contiguous()
        B, Cc, H, W = feat.shape
        assert B == 1
        pred = torch.empty((H, W), dtype=torch.int64, device=dev)
        L = _lib.lib()
        nws = L.halo_score_workspace_bytes(1, H, W)
        ws = _workspace(dev, nws, "score")
        rc = L.halo_quantize_radius(_lib.ptr(feat), _lib.dtype_code(feat), feat.stride(0), 1, Cc, H, W, int(self.K),
                                    float(self.mapper.c), _lib.ptr(pred), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "halo_quantize_radius")
        return pred

    def compute_region_impurity(self, predict, K):
        """predict (H,W) int64 -> (region_impurity, count), each (1,1,H,W) float32 (floating_region.py:112-121)."""
        dev = _lib.require_device(predict)
        if K != self.purity_channels:
            raise RuntimeError("purity window was built for %d channels, got K=%d" % (self.purity_channels, K))
        pred = predict.to(torch.int64).contiguous()
        H, W = pred.shape
        imp = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        cnt = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
        rc = _lib.lib().halo_region_impurity(_lib.ptr(pred), 1, H, W, int(self.purity_size), int(K), _lib.ptr(imp),
                                             _lib.ptr(cnt), _lib.PAD[self.padding_mode], _lib.stream_ptr(dev))
        _lib.check(rc, "halo_region_impurity")
        return imp, cnt

    def forward(self, logit: torch.Tensor, decoder_out: torch.Tensor = None, unc_type: str = None,
                pur_type: str = None, normalize: bool = False, ground_truth=None):
        """
        Compute regions score, impurity and uncertainty.

        Args:
            logit: (1, O, H, W) float32 on a ROCm device
            decoder_out: (1, C, H, W) embedding (float64 from the hyperbolic heads)
            unc_type: entropy | pixel_entropy | oracle_acc | anything else -> zeros
            pur_type: ripu | oracle_ripu | hyper | none | radius | euc_norm
            normalize: min-max normalise the impurity and uncertainty maps

        Return:
            score, purity, entropy -- each (H, W)
        """
        if pur_type not in _lib.PUR:
            raise NotImplementedError("Error: purity type '{}' not implemented".format(pur_type))
        self._check_purity_channels(pur_type)
        if logit.dim() == 3:
            logit = logit.unsqueeze(0)
        if logit.shape[0] != 1:
            raise ValueError("FloatingRegionScore.forward scores one image (as the reference); "
                             "use score_maps() for a batch")
        if decoder_out is not None and decoder_out.dim() == 3:
            decoder_out = decoder_out.unsqueeze(0)
        gt = None if ground_truth is None else ground_truth.unsqueeze(0)
        score, imp, unc = score_maps(logit, decoder_out, unc_type, pur_type, normalize, gt, size=self.size,
                                     purity_size=self.purity_size, K=getattr(self, "K", 100), c=self.mapper.c,
                                     padding_mode=self.padding_mode)
        return score[0], imp[0], unc[0]
