"""RegionSelection / select_pixels_to_label on HIP kernels -- host mirror of core/active/build.py.

`select_pixels_to_label` keeps the reference's signature and in-place semantics (build.py:27-64)
but runs the whole greedy loop on the device (halo_amd/csrc/halo_select.hip): zero host syncs
instead of >= 3 `.item()` per region.  `RegionSelection` keeps the reference's signature and its
on-disk side effects (uint8 mode-L PNG mask + torch.save'd {'active','selected'} indicator,
build.py:162-166) and batches images through the fused score -> mask -> select pipeline.
"""
import math
import os
import struct
import zlib

import numpy as np
import torch
from PIL import Image

from ... import _lib
from ..configs import cfg
from ..utils.hyperbolic import HyperMapper, bilinear_align_corners
from .floating_region import FloatingRegionScore, new_score_range, score_maps, score_maps_lowres, _workspace


def greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth,
                  return_picks=True, method=None, out=None, score_range=None, handover=None):
    """Batched device-side selection.  score (B,H,W) f32|f64, active/selected (B,H,W) bool,
    active_mask/ground_truth (B,H,W) int64 -- all on one ROCm device, all mutated in place.
    Returns (picks (B,n,3) float64 rows (h, w, value), n_picked (B,) int32) or None.
    method: "auto" (default; environment HALO_SELECT overrides) = value-binned sweep with the serial kernel
    behind it, "serial" = the tile-table kernel only, "binned" = the sweep or HaloUnsupported.  Same results.
    out: optional (picks (B,n,3) float64, n_picked (B,) int32) contiguous device tensors to write the tables into
    (pipelined callers collect a whole round's tables in one buffer); rows past an image's count are left as they are.
    score_range: the records score_maps / score_maps_lowres filled for these maps (new_score_range): the sweep then skips its
    pass over the map for the value range (the records only have to bound the values: the picks do not depend on them).
    handover: optional (B,2) int32 device tensor that receives, per image, {reason, picks made by the sweep} -- reason 0
    (_lib.SWEEP_REASONS[0]) = the value-binned sweep finished the image, anything else = the serial kernel continued it from that
    pick on.  A cost counter only: results never depend on which kernel made a pick."""
    dev = _lib.require_device(score, active, selected, active_mask, ground_truth)
    B, H, W = score.shape
    for t in (score, active, selected, active_mask, ground_truth):
        assert t.shape == (B, H, W) and t.is_contiguous(), "greedy_select expects contiguous (B,H,W) tensors"
    assert active.dtype == torch.bool and selected.dtype == torch.bool
    assert active_mask.dtype == torch.int64 and ground_truth.dtype == torch.int64
    n = int(max(0, min(int(n_regions), H * W)))
    picks = n_picked = None
    if out is not None:
        picks, n_picked = out
        assert picks.shape == (B, max(n, 1), 3) and picks.dtype == torch.float64 and picks.is_contiguous() and picks.device == dev
        assert n_picked.shape == (B,) and n_picked.dtype == torch.int32 and n_picked.is_contiguous() and n_picked.device == dev
        return_picks = True
    elif return_picks:
        picks = torch.zeros((B, max(n, 1), 3), dtype=torch.float64, device=dev)
        n_picked = torch.zeros((B,), dtype=torch.int32, device=dev)
    if handover is not None:
        assert handover.shape == (B, 2) and handover.dtype == torch.int32 and handover.is_contiguous() and handover.device == dev
    if n == 0 or B == 0:
        if handover is not None:
            handover.zero_()
        return (picks[:, :0], n_picked) if return_picks else None
    L = _lib.lib()
    name = method or os.environ.get("HALO_SELECT", "auto")
    if name not in _lib.SELECT:
        raise ValueError("greedy_select: method must be one of %s, got %r" % (sorted(_lib.SELECT), name))
    method = _lib.SELECT[name]
    nws = L.halo_select_workspace_bytes(B, H, W, n, int(mask_radius)) if method != _lib.SELECT["serial"] else 256
    ws = _workspace(dev, nws, "select")
    if score_range is not None:
        assert score_range.is_contiguous() and score_range.device == dev and score_range.numel() >= L.halo_score_range_bytes(B)
    rc = L.halo_greedy_select_ex(_lib.ptr(score), _lib.dtype_code(score), B, H, W, n, int(active_radius),
                                 int(mask_radius), _lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask),
                                 _lib.ptr(ground_truth), _lib.ptr(picks), _lib.ptr(n_picked), _lib.ptr(ws), ws.numel(),
                                 method, _lib.ptr(score_range), _lib.ptr(handover), _lib.stream_ptr(dev))
    _lib.check(rc, "halo_greedy_select")
    return (picks, n_picked) if return_picks else None


def _stage(t, dev, dtype):
    """Device staging copy of `t` unless it already is a contiguous `dtype` tensor on `dev`."""
    if t.device == dev and t.dtype == dtype and t.is_contiguous():
        return t, False
    return t.to(device=dev, dtype=dtype).contiguous(), True


def select_pixels_to_label(score, active_regions, active_radius, mask_radius, active, selected, active_mask,
                           ground_truth):
    """Drop-in for build.py:27-64.  `score` must be on a ROCm device.  As in the reference's call
    site (build.py:115-120) `active`/`selected` may be CPU tensors while `score`/`active_mask`/
    `ground_truth` are on the device: they are staged, updated, and written back in place."""
    dev = _lib.require_device(score)
    assert score.dim() == 2, "score must be (H, W)"
    sc, sc_c = _stage(score, dev, score.dtype if score.dtype in (torch.float32, torch.float64) else torch.float32)
    ac, ac_c = _stage(active, dev, torch.bool)
    se, se_c = _stage(selected, dev, torch.bool)
    am, am_c = _stage(active_mask, dev, torch.int64)
    gt, _ = _stage(ground_truth, dev, torch.int64)
    greedy_select(sc[None], active_regions, active_radius, mask_radius, ac[None], se[None], am[None], gt[None],
                  return_picks=False)
    for dst, src, copied in ((score, sc, sc_c), (active, ac, ac_c), (selected, se, se_c), (active_mask, am, am_c)):
        if copied:
            dst.copy_(src)
    return score, active, selected, active_mask


def to_np_array(tensor):
    return np.array(tensor.cpu().numpy(), dtype=np.uint8)


def acquire_batch(logit, decoder_out, ground_truth, active, selected, active_mask, *, unc_type, pur_type, normalize,
                  n_regions, active_radius, mask_radius, size=None, purity_size=None, K=100, c=1.0):
    """score -> `score[active] = -inf` -> greedy selection for a batch of full-resolution images
    (build.py:137-160 for B images at once).  Mutates active/selected/active_mask; returns
    (picks (B,n,3), n_picked (B,))."""
    size = 2 * active_radius + 1 if size is None else size
    # normalised maps are bounded by [0, 1]: the scorer hands the selector their range for free
    rng = new_score_range(logit.shape[0], logit.device) if normalize and logit.shape[0] else None
    score, _, _ = score_maps(logit, decoder_out, unc_type, pur_type, normalize, ground_truth, size=size,
                             purity_size=purity_size, K=K, c=c, active=active, want_maps=False, score_range=rng)
    return greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth, score_range=rng)


def acquire_batch_lowres(logit_lr, decoder_lr, size, ground_truth, active, selected, active_mask, *, unc_type, pur_type,
                         normalize, n_regions, active_radius, mask_radius, ksize=None, purity_size=None, K=100, c=1.0,
                         lowres_mode=None):
    """build.py:122-160 for B images: resize of the head outputs fused into the scorer (the C x H x W
    embedding is never materialised), then mask + greedy selection.  Falls back to explicit HIP
    upsampling when the fused kernel declines the geometry (strong downsampling)."""
    ksize = 2 * active_radius + 1 if ksize is None else ksize
    rng = new_score_range(logit_lr.shape[0], logit_lr.device) if normalize and logit_lr.shape[0] else None
    try:
        score, _, _ = score_maps_lowres(logit_lr, decoder_lr, size, unc_type, pur_type, normalize, ground_truth,
                                        ksize=ksize, purity_size=purity_size, K=K, c=c, active=active, want_maps=False,
                                        mode=lowres_mode, score_range=rng)
    except _lib.HaloUnsupported:
        logit = bilinear_align_corners(logit_lr.float(), size)
        dec = bilinear_align_corners(decoder_lr, size) if pur_type in ("hyper", "radius", "euc_norm") else decoder_lr
        score, _, _ = score_maps(logit, dec, unc_type, pur_type, normalize, ground_truth, size=ksize,
                                 purity_size=purity_size, K=K, c=c, active=active, want_maps=False, score_range=rng)
    return greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth, score_range=rng)


class AcquisitionParams:
    """What RegionSelection reads from cfg (build.py:75-88), resolved once."""

    def __init__(self, cfg_):
        act = cfg_.ACTIVE
        self.radius = act.RADIUS_K
        self.mask_radius = act.MASK_RADIUS_K
        self.window = 2 * self.radius + 1
        self.round_budget = act.BUDGET / len(act.SELECT_ITER)          # per image, per round
        self.unc, self.pur, self.K = act.UNCERTAINTY, act.PURITY, act.K
        self.normalize = act.NORMALIZE
        self.scorer = FloatingRegionScore(in_channels=cfg_.MODEL.NUM_CLASSES, size=self.window,
                                          purity_type=self.pur, K=self.K)
        # the reference's scorer reads the curvature from the one global cfg (floating_region.py:68); here the
        # cfg that was PASSED IN decides, whether or not halo_amd.core.configs.use() was called
        self.scorer.mapper = HyperMapper(c=cfg_.MODEL.CURVATURE)
        if getattr(act, "VIZ_MASK", False):
            import warnings
            warnings.warn("halo_amd RegionSelection: cfg.ACTIVE.VIZ_MASK is set, but the visualisation plots of "
                          "core/active/build.py:168-183 are not produced (scoring and masking are fused; "
                          "plots are out of scope)", RuntimeWarning, stacklevel=3)
        if self.pur not in _lib.PUR:
            raise NotImplementedError("Error: purity type '{}' not implemented".format(self.pur))
        self.scorer._check_purity_channels(self.pur)

    def regions(self, n_pixels):
        """build.py:148-150"""
        return math.ceil(n_pixels * self.round_budget / self.window ** 2)


class _InFlight:
    """One loader batch (b images of one label size, b = 1 in the reference) whose staging, scoring, selection and device->host
    copies have been enqueued on a side stream."""
    __slots__ = ("done", "keep", "picks", "npk", "slot", "buf", "b", "left", "lock", "table", "radius", "compose", "mask_radius", "write",
                 "out_picks")


class _SlotBuffers:
    """Staging memory of one pipeline slot for b images of H x W labels, allocated once and reused by every batch of that shape
    that passes through the slot (pinning host pages costs milliseconds; per-image device allocations cost allocator traffic on
    the launching thread): device masks and pinned outputs."""

    def __init__(self, b, H, W, dev):
        shp = (b, H, W)
        self.shape, self.dev = shp, dev
        self.d_amask = torch.empty(shp, dtype=torch.int64, device=dev)
        self.d_gt = torch.empty(shp, dtype=torch.int64, device=dev)
        self.d_active = torch.empty(shp, dtype=torch.bool, device=dev)
        self.d_selected = torch.empty(shp, dtype=torch.bool, device=dev)
        self.d_mask8 = torch.empty(shp, dtype=torch.uint8, device=dev)
        self.out_mask = torch.empty(shp, dtype=torch.uint8, pin_memory=True)
        self.out_active = torch.empty(shp, dtype=torch.bool, pin_memory=True)
        self.out_selected = torch.empty(shp, dtype=torch.bool, pin_memory=True)
        self.out_npk = torch.empty((b,), dtype=torch.int32, pin_memory=True)
        self.out_picks_by_n = {}

    def picks_out(self, n):
        """pinned (b, n, 3) float64 for the round's pick tables (one buffer per table width, never re-allocated: a recorded launch
        group copies into it).  The batch in flight remembers WHICH one it wrote (`rec.out_picks`): one slot shape can serve
        several table widths (another ACTIVE.BUDGET or RADIUS_K in the same process), and a replayed recording fills the buffer
        of the width it was recorded with."""
        got = self.out_picks_by_n.get(int(n))
        if got is None:
            got = self.out_picks_by_n[int(n)] = torch.empty((self.shape[0], int(n), 3), dtype=torch.float64, pin_memory=True)
        return got


class _Slot:
    """One pipeline slot: a side stream position plus its staging buffers per batch shape (at most four shapes are kept: a pool
    of mixed label sizes re-allocates, a uniform one -- Cityscapes -- never does)."""

    def __init__(self, stream):
        self.stream = stream
        self.bufs = {}
        self.graphs = {}        # launch-group key -> _SlotGraph (or a use count before it is captured)

    def buffers(self, b, H, W, dev):
        key = (int(b), int(H), int(W))
        buf = self.bufs.pop(key, None)
        if buf is None:
            if len(self.bufs) >= 4:
                self.bufs.pop(next(iter(self.bufs)))
                self.graphs.clear()                              # recorded launch groups point into the slot's buffers
            buf = _SlotBuffers(key[0], key[1], key[2], dev)
        self.bufs[key] = buf                                     # most recently used last
        return buf


class _SlotGraph:
    """The launch group of one (slot, batch shape, acquisition parameters) captured ONCE as a HIP graph: fused resize + score ->
    `score[active] = -inf` -> selection -> device-to-host copies of the pick table, reading STATIC device copies of the head's
    outputs and writing the slot's pinned buffers.  A batch then costs the launching thread three copies into the static inputs, one
    replay and one event record instead of ~25 launches through Python and ctypes (0.24-0.35 ms per image)."""
    __slots__ = ("graph", "logits", "embed", "picks", "npk", "scratch", "out_picks")


_SIDE = {}
_SLOTS = {}
_CAPTURE = {}


def _capture_stream(dev):
    """The stream a slot's launch group is RECORDED on: one HIP stream per device, created outside torch's stream pool and
    used for nothing else.  A recording that fails (the runtime invalidates a capture for reasons outside this code: seen once
    in ~1700 captures under GPU_MAX_HW_QUEUES=8) leaves its stream in capture mode on ROCm 7.2 / torch 2.10 (observed: after torch's
    capture_end raised, the stream still refused copies and 'recorded' events into the dead capture) -- so it must not be the slot's own stream (every later copy, launch and event on
    it failed: the round died in a writer thread's event wait) nor one of torch's 32 pooled streams (handed out again later).
    The graph replays on the slot's stream whatever stream recorded it."""
    s = _CAPTURE.get(dev.index)
    if s is None:
        import ctypes
        hip = _lib.hip_runtime()                                  # the runtime torch runs on, not a second copy
        raw = ctypes.c_void_p()
        with torch.cuda.device(dev):
            err = hip.hipStreamCreateWithFlags(ctypes.byref(raw), ctypes.c_uint(1))     # hipStreamNonBlocking
        if err != 0 or not raw.value:
            raise RuntimeError("hipStreamCreateWithFlags failed (%d)" % err)
        s = _CAPTURE[dev.index] = torch.cuda.ExternalStream(raw.value, device=dev)
    return s


def _side_streams(dev, n):
    """The acquisition's side streams, created once per device and reused by every round: the scorer's and the selector's
    scratch buffers are cached per stream (floating_region._workspace), so fresh streams per call would strand them."""
    have = _SIDE.setdefault(dev.index, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(dev, priority=-1))
    return have[:n]


def _launch(prm, logits_lr, embed_lr, size, origin_mask, origin_label, active_in, selected_in, dev, slot, lowres_mode=None, stats=None,
            mask_staging="table", write_files=True):
    """Enqueue b images of one label size (build.py:113-166 for each) on the slot's stream: stage what the device needs, score ->
    mask -> select as ONE batch, copy the results back into the slot's pinned buffers.  Fully asynchronous: `rec.done` fires
    when the pinned buffers hold the results of all b images.

    What crosses PCIe.  The loader hands `origin_mask` / `origin_label` as int64, 16.8 MB each per 1024x2048
    image, and the reference copies both to the GPU only to write `active_mask[window] = ground_truth[window]` at the round's
    <= 2331 picks (build.py:58-62) before the mask comes back and is cast to uint8 (build.py:67-68,162).
      mask_staging="table" (default): neither map travels.  The device scores and selects; it needs `active` (2.1 MB in: the
        scorer masks those pixels) and returns ONLY the 56 KB pick table: the writer thread composes all three results on the host
        -- the mask = the low byte of `origin_mask` (what the uint8 cast keeps) with the labels of the picks' windows copied in
        (~21 000 pixels), `active` / `selected` = the loader's maps with the picks' windows set (round 4 copied both back, 4.2 MB).
        The label map still goes to the device when the scorer itself reads it (oracle_acc / oracle_ripu).  In this mode the
        whole launch group is replayed from a HIP graph per (slot, shape) from its third use on (_SlotGraph; HALO_RS_GRAPH=0: eager).
      mask_staging="device": the reference's data flow -- both maps are DMA'd from the loader's (pinned) tensors into the slot's
        device buffers (37.7 MB, 0.69 ms per image), the selection kernel writes the windows, the mask returns as uint8.
    Same files either way (tests/test_gpu_parity.py runs both against the reference's PNGs)."""
    import time
    rec = _InFlight()
    rec.out_picks = None
    rec.slot, rec.b = slot, int(origin_mask.shape[0])
    H, W = int(origin_mask.shape[-2]), int(origin_mask.shape[-1])
    buf = rec.buf = slot.buffers(rec.b, H, W, dev)
    stream = slot.stream
    t1 = time.perf_counter()
    scorer_reads_gt = prm.unc == "oracle_acc" or prm.pur == "oracle_ripu"
    # masks that already live on the device are composed there
    rec.table = mask_staging == "table" and not origin_mask.is_cuda and not origin_label.is_cuda
    # ... and so are the indicator maps, when the loader's copies are host tensors (they always are in the reference)
    rec.compose = rec.table and not active_in.is_cuda and not selected_in.is_cuda
    rec.write = bool(write_files)
    n_regions = prm.regions(size[0] * size[1])
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))             # the head outputs are complete from here on
    def body(lg, em):
        """the group's device work behind its inputs (eager, or recorded into the slot's graph)"""
        # table staging: the selection kernel still writes its windows (active_mask[window] = ground_truth[window]) -- into the
        # slot's scratch mask, from itself when no label map is resident: nobody reads that buffer
        gt_dev = buf.d_gt if (not rec.table or scorer_reads_gt) else buf.d_amask
        # the two F.interpolate(align_corners=True) calls of build.py:122-135 are fused into the scorer:
        # the C x H x W float64 embedding (4.3 GB at C=256) is never written or read
        picks, npk = acquire_batch_lowres(
                             lg, em, size, gt_dev, buf.d_active, buf.d_selected, buf.d_amask,
                             unc_type=prm.unc, pur_type=prm.pur, normalize=prm.normalize,
                             n_regions=n_regions, active_radius=prm.radius,
                             mask_radius=prm.mask_radius, ksize=prm.scorer.size, purity_size=prm.scorer.purity_size,
                             K=prm.K, c=prm.scorer.mapper.c, lowres_mode=lowres_mode)
        if rec.table:
            rec.out_picks = buf.picks_out(picks.shape[1])
            rec.out_picks.copy_(picks, non_blocking=True)
        else:
            # uint8 on the device first: 2 MB instead of 16 MB over PCIe per 1024x2048 mask (same values as the
            # reference's cast-after-copy, build.py:67-68,162)
            buf.d_mask8.copy_(buf.d_amask)
            buf.out_mask.copy_(buf.d_mask8, non_blocking=True)
        if not rec.compose:
            buf.out_active.copy_(buf.d_active, non_blocking=True)
            buf.out_selected.copy_(buf.d_selected, non_blocking=True)
        buf.out_npk.copy_(npk, non_blocking=True)           # (a .item() in the writer thread would queue behind the backbone's kernels)
        return picks, npk

    # ---- graph replay: the common case (host-composed results, nothing but `active` and the head outputs to stage)
    gkey = sg = None
    if rec.compose and not scorer_reads_gt and os.environ.get("HALO_RS_GRAPH", "1") != "0":
        gkey = (rec.b, H, W, tuple(logits_lr.shape), logits_lr.dtype, tuple(embed_lr.shape), embed_lr.dtype, tuple(size), prm.unc, prm.pur,
                bool(prm.normalize), n_regions, prm.radius, prm.mask_radius, prm.scorer.size, prm.scorer.purity_size, prm.K,
                float(prm.scorer.mapper.c), lowres_mode)
        sg = slot.graphs.get(gkey)
    with torch.cuda.stream(stream):
        stream.wait_event(ready)
        if isinstance(sg, _SlotGraph):
            sg.logits.copy_(logits_lr, non_blocking=True)
            sg.embed.copy_(embed_lr, non_blocking=True)
            buf.d_active.copy_(active_in, non_blocking=True)
            sg.graph.replay()
            rec.picks, rec.npk, rec.out_picks = sg.picks, sg.npk, sg.out_picks
        else:
            if not rec.table:
                buf.d_amask.copy_(origin_mask, non_blocking=True)
            if not rec.table or scorer_reads_gt:
                buf.d_gt.copy_(origin_label, non_blocking=True)
            buf.d_active.copy_(active_in, non_blocking=True)
            if not rec.compose:                                   # (composed on the host: the kernel's window writes land in scratch)
                buf.d_selected.copy_(selected_in, non_blocking=True)
            rec.picks, rec.npk = body(logits_lr, embed_lr)
            if gkey is not None:
                # the first group of a key runs eagerly (kernels load, workspaces and LDS limits settle); the second is ALSO recorded
                # -- behind its eager run, so this batch is unaffected -- and every later one replays the recording
                seen = slot.graphs.get(gkey, 0) + 1
                slot.graphs[gkey] = seen
                if seen == 2:
                    try:
                        if len(slot.graphs) > 3:                  # pools of many label sizes: keep the newest keys only
                            for k_ in list(slot.graphs)[:-3]:
                                slot.graphs.pop(k_)
                        g_ = _SlotGraph()
                        g_.logits, g_.embed = torch.empty_like(logits_lr), torch.empty_like(embed_lr)
                        g_.graph = torch.cuda.CUDAGraph()
                        from .floating_region import private_workspaces
                        with private_workspaces() as pw, torch.cuda.graph(g_.graph, stream=_capture_stream(dev), capture_error_mode="thread_local"):
                            g_.picks, g_.npk = body(g_.logits, g_.embed)
                        g_.out_picks = rec.out_picks              # the pinned table the recording copies into
                        g_.scratch = pw.held                      # the recording owns the scratch buffers it points to
                        slot.graphs[gkey] = g_
                    except Exception as exc:                      # capture refused (driver / torch build): stay eager, say so once
                        slot.graphs[gkey] = -(1 << 30)
                        _CAPTURE.pop(dev.index, None)             # that stream may be stuck in capture mode: never used again
                        torch.cuda.set_stream(stream)             # (torch's context manager raises before it restores the stream)
                        import warnings
                        warnings.warn("halo_amd RegionSelection: HIP graph capture of the launch group failed (%s); launching eagerly" % exc,
                                      RuntimeWarning)
        rec.done = torch.cuda.Event(blocking=True)           # the writer threads sleep on it instead of spinning
        rec.done.record(stream)
    # what the side stream and the writer threads still read: the head outputs and the loader's tensors, kept alive until the
    # batch is retired
    rec.keep = (logits_lr, embed_lr, origin_mask, origin_label, active_in, selected_in)
    rec.radius, rec.mask_radius = prm.radius, prm.mask_radius
    if stats is not None:
        stats["main_launch_s"] += time.perf_counter() - t1
    return rec


def compose_mask(origin_mask, origin_label, picks, active_radius):
    """The uint8 mask file of one image from host data and the pick table (mask_staging="table"): the low byte of every
    `origin_mask` value -- what the reference's uint8 cast keeps (to_np_array, build.py:67-68) -- with
    `ground_truth[h-r : h+r+1, w-r : w+r+1]` copied in around every pick (h, w) of the round, windows clipped at the image
    borders as the reference's slices are (build.py:52-62).  numpy throughout: the big copies release the GIL.
    origin_mask, origin_label: (H, W) integer arrays; picks: (k, >=2) rows (h, w, ...)."""
    H, W = origin_mask.shape
    mask = np.empty((H, W), dtype=np.uint8)
    np.copyto(mask, origin_mask, casting="unsafe")               # int64 -> uint8 wraps modulo 256, like numpy's astype
    k = picks.shape[0]
    if k:
        r = int(active_radius)
        d = np.arange(-r, r + 1, dtype=np.int64)
        rows = picks[:, 0].astype(np.int64)[:, None, None] + d[None, :, None]
        cols = picks[:, 1].astype(np.int64)[:, None, None] + d[None, None, :]
        rows, cols = np.broadcast_arrays(rows, cols)
        ok = (rows >= 0) & (rows < H) & (cols >= 0) & (cols < W)
        rr, cc = rows[ok], cols[ok]
        mask[rr, cc] = origin_label[rr, cc].astype(np.uint8)     # the same wrap for the labels
    return mask


def compose_indicators(prior_active, prior_selected, picks, active_radius, mask_radius):
    """`active` / `selected` (H, W) bool after a round, from the maps the image entered it with and the round's picks
    (build.py:56-59: active[h-R:h+R+1, w-R:w+R+1] = True with R = mask radius, selected[...] with R = radius; slices clipped at the
    borders like the reference's).  The native twin is halo_compose_indicators (libhalo_host.so); this is its numpy statement."""
    H, W = prior_active.shape
    act = np.array(prior_active, dtype=bool, copy=True)
    sel = np.array(prior_selected, dtype=bool, copy=True)
    for row in np.asarray(picks)[:, :2].astype(np.int64):
        h, w = int(row[0]), int(row[1])
        for dst, r in ((act, int(mask_radius)), (sel, int(active_radius))):
            dst[max(h - r, 0):h + r + 1, max(w - r, 0):w + r + 1] = True
    return act, sel


_PNG_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def _write_png_gray8_zlib(path, arr):
    """The pure-Python statement of write_png_gray8 (round 2-3): filter type 0 on every row and ONE zlib stream with the
    run-length strategy.  ~3-4.5 ms per 1024x2048 mask; kept as the A/B twin of the native encoder (HALO_PNG_ZLIB=1) and for
    hosts without a C compiler."""
    h, w = arr.shape
    rows = np.empty((h, w + 1), dtype=np.uint8)
    rows[:, 0] = 0                                                   # filter type "None" per scanline
    rows[:, 1:] = arr
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)              # bit depth 8, colour type 0 (greyscale), no interlace
    z = zlib.compressobj(1, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    idat = z.compress(rows) + z.flush()
    with open(path, "wb") as f:
        f.write(_PNG_SIGNATURE + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", idat) + _png_chunk(b"IEND", b""))


_NATIVE_PNG = None


def write_png_gray8(path, arr):
    """8-bit greyscale (PIL mode "L") PNG of a (H, W) uint8 array: the file `Image.fromarray(a).save(path)` of
    build.py:163-164 produces, pixel for pixel, written by the native run-length encoder of halo_amd/csrc/halo_host.c (one
    fixed-Huffman deflate block, runs found 8 bytes at a time: ~0.4 ms per 1024x2048 acquisition mask where PIL's encoder takes
    ~12 ms and zlib's run-length strategy 3-4.5 ms; the call releases the GIL, so writer threads run in parallel).  Decodes to
    the identical image."""
    global _NATIVE_PNG
    if _NATIVE_PNG is None:
        try:
            from ... import _hostlib
            _hostlib.lib()
            _NATIVE_PNG = _hostlib.png_gray8_write
        except Exception as exc:                                      # no C compiler and no prebuilt library: the zlib twin
            import warnings
            warnings.warn("halo_amd: libhalo_host.so unavailable (%s); PNG masks go through zlib" % exc, RuntimeWarning)
            _NATIVE_PNG = False
    if _NATIVE_PNG and not os.environ.get("HALO_PNG_ZLIB"):
        if arr.strides[1] != 1 or arr.strides[0] < arr.shape[1]:
            arr = np.ascontiguousarray(arr)
        _NATIVE_PNG(path, arr)
    else:
        _write_png_gray8_zlib(path, arr)


class _IndicatorTemplate:
    """The bytes torch.save writes for {'active': bool (H, W), 'selected': bool (H, W)} (build.py:165-166), produced ONCE per
    shape by torch.save itself: the zip archive stores the two tensor payloads uncompressed, so an image's file is the template
    with the two payloads and their CRC-32 fields (data descriptor or local header, and central directory) replaced -- which
    libhalo_host.so does without the interpreter lock (halo_write_indicator).  The format is whatever the installed torch
    writes; the parsed template is checked once by torch.load on a rendered random pair, and any surprise (compressed payloads,
    a layout this parser does not know) disables it: the writer then calls torch.save."""

    _cache = {}
    _lock = None

    @classmethod
    def get(cls, shape):
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
        key = tuple(int(v) for v in shape)
        tpl = cls._cache.get(key)
        if tpl is None:
            with cls._lock:
                tpl = cls._cache.get(key)
                if tpl is None:
                    if len(cls._cache) >= 8:
                        cls._cache.pop(next(iter(cls._cache)))
                    tpl = cls._cache[key] = cls(key)
        return tpl

    def __init__(self, shape):
        self.ok, self.shape = False, shape
        try:
            self._build(shape)
            self.ok = self._self_check()
        except Exception:
            self.ok = False

    def _build(self, shape):
        import io
        import zipfile
        n = int(np.prod(shape))
        a = torch.zeros(shape, dtype=torch.bool)
        b = torch.zeros(shape, dtype=torch.bool)
        a.view(-1)[0] = True                                     # two different payloads: two different CRCs to tell apart
        buf = io.BytesIO()
        torch.save({"active": a, "selected": b}, buf)
        raw = buf.getvalue()
        self.raw = np.frombuffer(raw, dtype=np.uint8).copy()
        zf = zipfile.ZipFile(io.BytesIO(raw))
        want = {a.numpy().tobytes(): "active", b.numpy().tobytes(): "selected"}
        self.off, self.crc = {}, {}                               # name -> payload offset, the two CRC field offsets
        central, pos = {}, zf.start_dir
        for _ in zf.infolist():                                   # central directory: signature, ..., crc at +16, lengths at +28..
            if raw[pos:pos + 4] != b"PK\x01\x02":
                raise ValueError("central directory entry expected")
            fl, el, cl = struct.unpack_from("<HHH", raw, pos + 28)
            central[raw[pos + 46:pos + 46 + fl].decode("utf-8")] = pos + 16
            pos += 46 + fl + el + cl
        for info in zf.infolist():
            if info.file_size != n or "/data/" not in "/" + info.filename:
                continue
            name = want.get(zf.read(info.filename))
            if name is None:
                continue
            if info.compress_type != zipfile.ZIP_STORED or info.file_size >= 0xffffffff:
                raise ValueError("stored 32-bit zip entries expected")
            sig, _, flag, _, _, _, _, _, _, fl, el = struct.unpack_from("<IHHHHHIIIHH", raw, info.header_offset)
            if sig != 0x04034b50:
                raise ValueError("local header expected")
            off = info.header_offset + 30 + fl + el
            if flag & 8:                                          # data descriptor behind the payload: [signature] crc sizes
                d = off + n
                second = d + 4 if raw[d:d + 4] == b"PK\x07\x08" else d
            else:
                second = info.header_offset + 14
            fields = [central[info.filename], second]
            for c in fields:
                if struct.unpack_from("<I", raw, c)[0] != info.CRC:
                    raise ValueError("CRC field not where expected")
            self.off[name], self.crc[name] = off, np.array(fields, dtype=np.uint64)
        if set(self.off) != {"active", "selected"}:
            raise ValueError("payload records not found")

    def _self_check(self):
        import io
        import tempfile
        from ... import _hostlib
        rng = np.random.default_rng(7)
        a, b = np.ascontiguousarray(rng.random(self.shape) < 0.3), np.ascontiguousarray(rng.random(self.shape) < 0.6)
        with tempfile.TemporaryDirectory(prefix="halo_tpl_") as tmp:
            p = os.path.join(tmp, "i.pth")
            rc = _hostlib.lib().halo_write_indicator(os.fsencode(p), self.raw.ctypes.data, self.raw.size, a.ctypes.data, b.ctypes.data, a.size,
                                                     self.off["active"], self.off["selected"], self.crc["active"].ctypes.data,
                                                     self.crc["selected"].ctypes.data)
            if rc != 0:
                return False
            got = torch.load(p)
        return (set(got) == {"active", "selected"} and got["active"].dtype == torch.bool and got["selected"].dtype == torch.bool
                and tuple(got["active"].shape) == tuple(self.shape) and np.array_equal(got["active"].numpy(), a)
                and np.array_equal(got["selected"].numpy(), b))


def _persist(mask_np, active, selected, path_mask, path_indicator, stats=None):
    """build.py:162-166: uint8 mode-L PNG + torch.save'd indicator dict (what cityscapes.py:234-251 reads back)."""
    import time
    t0 = time.perf_counter()
    if mask_np.ndim == 2 and mask_np.dtype == np.uint8 and mask_np.size and str(path_mask).lower().endswith(".png"):
        write_png_gray8(path_mask, mask_np)
    else:
        Image.fromarray(mask_np).save(path_mask)
    t1 = time.perf_counter()
    torch.save({"active": active, "selected": selected}, path_indicator)
    if stats is not None:
        with stats["lock"]:
            stats["writer_png_s"] += t1 - t0
            stats["writer_save_s"] += time.perf_counter() - t1


def _native_retire():
    """the one-call native writer (libhalo_host.so), or None: HALO_RETIRE_PYTHON=1 (A/B switch) or no host library"""
    if os.environ.get("HALO_RETIRE_PYTHON"):
        return None
    try:
        from ... import _hostlib
        _hostlib.lib()
        return _hostlib.retire_image
    except Exception:
        return None


def _finish(rec, i, paths, slots, stats=None, backlog=None):
    try:
        return _finish_image(rec, i, paths, slots, stats)
    finally:
        if backlog is not None:
            backlog.release()


def _finish_image(rec, i, paths, slots, stats=None):
    """Writer-thread half of image i of a batch: wait for the batch's copies, turn the image's results in the slot's pinned
    buffers into its two files (the slot goes back when the last image of the batch is done with the buffers)."""
    import time
    t0 = time.perf_counter()
    native = False
    t_png = t_copy = t_save = 0.0
    released = False

    def release():
        nonlocal released
        if released:
            return
        released = True
        with rec.lock:
            rec.left -= 1
            last = rec.left == 0
        if last:
            rec.keep = rec.buf = None                         # (references taken before keep the loader's tensors alive)
            slots.put(rec.slot)
    try:
        rec.done.synchronize()
        t1 = time.perf_counter()
        buf = rec.buf
        k = int(buf.out_npk[i])
        if not rec.write:                                        # global-budget rounds: the files follow once the pool's keep-mask is known
            return (torch.from_numpy(rec.out_picks[i].numpy().copy()) if rec.table else rec.picks[i], k)
        if rec.compose:
            # everything the files need is host data: the image's 56 KB pick table leaves the slot's pinned buffer FIRST and the slot goes
            # back to the launching thread; the 2-3 ms of composing, encoding and writing then hold no pipeline resource (round 5: with
            # the slot held until the files were written, 8 slots / 2.7 ms bounded the round at 0.34 ms per image whatever else improved)
            table = rec.out_picks[i].numpy().copy()
            keep = rec.keep
            release()
            persist_image(paths[0], paths[1], keep[2][i], keep[3][i], keep[4][i], keep[5][i], table, k, rec.radius, rec.mask_radius)
            if stats is not None:
                with stats["lock"]:
                    stats["writer_event_wait_s"] += t1 - t0
                    stats["writer_png_s"] += time.perf_counter() - t1
            return (torch.from_numpy(table), k)
        is_png = bool(buf.out_mask[i].numel()) and str(paths[0]).lower().endswith(".png")
        mask = active = selected = None
        retire = _native_retire() if (rec.table and is_png) else None
        # the indicator maps: the device's results in the pinned buffers (host-composed batches returned above)
        ind_a, ind_s = buf.out_active[i].numpy(), buf.out_selected[i].numpy()
        if retire is not None:
            # mask_staging="table": ONE call without the interpreter lock composes the mask from the loader's maps and the pick
            # table, encodes it, and writes the indicator from the pinned maps through the shape's template
            om, gt = rec.keep[2][i].numpy(), rec.keep[3][i].numpy()
            if om.flags["C_CONTIGUOUS"] and gt.flags["C_CONTIGUOUS"] and om.dtype.kind in "iub" and gt.dtype.kind in "iub" \
                    and om.ctypes.data % om.dtype.itemsize == 0 and gt.ctypes.data % gt.dtype.itemsize == 0:      # (naturally aligned elements)
                tpl = _IndicatorTemplate.get(om.shape)
                retire(paths[0], paths[1], om, gt, rec.out_picks[i].numpy(), k, rec.radius, ind_a, ind_s, tpl if tpl.ok else None,
                       compose_mask_radius=-1)
                native = True
                if not tpl.ok:
                    active, selected = torch.from_numpy(ind_a.copy()), torch.from_numpy(ind_s.copy())
                t_png = time.perf_counter() - t1
        if not native:
            if rec.table:
                table = rec.out_picks[i, :k].numpy().copy()
                origin_mask, origin_label = rec.keep[2][i].numpy(), rec.keep[3][i].numpy()
            elif is_png:
                # the PNG encoder makes ONE pass over the mask: it reads the pinned buffer directly
                write_png_gray8(paths[0], buf.out_mask[i].numpy())
            else:
                mask = buf.out_mask[i].numpy().copy()
            t2 = time.perf_counter()
            t_png = t2 - t1
            # the indicator must hold plain tensors as from `.cpu()`: one streaming copy each (numpy on purpose: a torch CPU op
            # here would wake an intra-op thread pool as wide as the host)
            active, selected = torch.from_numpy(ind_a.copy()), torch.from_numpy(ind_s.copy())
            t_copy = time.perf_counter() - t2
        # the table handed back (return_tables): a copy of the pinned rows where the device tensor belongs to a replayed graph
        out = (torch.from_numpy(rec.out_picks[i].numpy().copy()) if rec.table else rec.picks[i], k)
    finally:
        release()
    t3 = time.perf_counter()
    if not native:
        if rec.table:
            mask = compose_mask(origin_mask, origin_label, table, rec.radius)
            if is_png:
                write_png_gray8(paths[0], mask)
        if mask is not None and not is_png:
            Image.fromarray(mask).save(paths[0])
        t_png += time.perf_counter() - t3
    t4 = time.perf_counter()
    if active is not None:
        torch.save({"active": active, "selected": selected}, paths[1])
        t_save = time.perf_counter() - t4
    if stats is not None:
        with stats["lock"]:
            stats["writer_event_wait_s"] += t1 - t0
            stats["writer_png_s"] += t_png
            stats["writer_copy_s"] += t_copy
            stats["writer_save_s"] += t_save
    return out


def persist_image(path_mask, path_indicator, origin_mask, origin_label, prior_active, prior_selected, picks, k, active_radius, mask_radius):
    """One image's two files (build.py:162-166) from HOST data and the first `k` rows of its pick table: the mask = origin_mask with
    the labels of the picks' windows, the indicators = the maps the image entered the round with plus the picks' windows.  No GPU
    involved: what RegionSelection's writer threads do in table staging, and what a global-budget round does once the pool-wide
    keep-mask says how many of an image's picks were kept.  origin_* / prior_*: (H, W) torch tensors or numpy arrays."""
    def np_(x):
        return x.numpy() if torch.is_tensor(x) else np.asarray(x)
    om, gt, pa, ps = np_(origin_mask), np_(origin_label), np_(prior_active), np_(prior_selected)
    pk = np.ascontiguousarray(np_(picks)[:int(k)], dtype=np.float64).reshape(-1, 3)
    retire = _native_retire() if str(path_mask).lower().endswith(".png") else None
    ok = retire is not None and all(a.flags["C_CONTIGUOUS"] for a in (om, gt, pa, ps)) and om.dtype.kind in "iub" and gt.dtype.kind in "iub" \
        and pa.itemsize == 1 and ps.itemsize == 1
    if ok:
        tpl = _IndicatorTemplate.get(om.shape)
        if tpl.ok:
            retire(path_mask, path_indicator, om, gt, pk if len(pk) else np.zeros((1, 3)), len(pk), active_radius, pa, ps, tpl,
                   compose_mask_radius=mask_radius)
            return
    act, sel = compose_indicators(pa, ps, pk, active_radius, mask_radius)
    _persist(compose_mask(om, gt, pk, active_radius), torch.from_numpy(act), torch.from_numpy(sel), path_mask, path_indicator)


def persist_from_tables(cfg, loader, tables, counts, writer_threads=None, expect_paths=None):
    """Write the round's files for every image of `loader` (in loader order) from pick tables: image j gets the windows of the first
    counts[j] rows of tables[j] (rows (h, w, score), as RegionSelection(return_tables=True) / the all-gather deliver them).  Used by
    the global-budget mode of halo_amd.pool.region_selection_sharded, where counts = the KEPT picks per image.  Host only.
    `expect_paths`: the `path_to_mask` sequence of the pass that computed the tables; a loader that yields another order (a shuffling
    sampler, a dataset that changed in between) is refused before the first mismatched file is written."""
    from concurrent.futures import ThreadPoolExecutor
    if writer_threads is None:
        from ..._host import host_threads_per_rank
        writer_threads = host_threads_per_rank(cap=8)
    radius, mask_radius = cfg.ACTIVE.RADIUS_K, cfg.ACTIVE.MASK_RADIUS_K
    tables = tables.cpu().numpy() if torch.is_tensor(tables) else np.asarray(tables)
    counts = [int(c) for c in (counts.cpu().tolist() if torch.is_tensor(counts) else counts)]
    j, pending = 0, []
    with ThreadPoolExecutor(max_workers=max(1, writer_threads)) as writers:
        for batch in loader:
            for i in range(len(batch["origin_mask"])):
                if expect_paths is not None and (j >= len(expect_paths) or str(batch["path_to_mask"][i]) != expect_paths[j]):
                    raise RuntimeError("persist_from_tables: image %d of the loader is %r, the tables were computed for %r -- the loader "
                                       "must yield the same order in both passes" % (j, batch["path_to_mask"][i],
                                                                                      expect_paths[j] if j < len(expect_paths) else None))
                pending.append(writers.submit(persist_image, batch["path_to_mask"][i], batch["path_to_indicator"][i], batch["origin_mask"][i],
                                              batch["origin_label"][i], batch["active"][i], batch["selected"][i], tables[j], counts[j],
                                              radius, mask_radius))
                j += 1
            while len(pending) > 4 * max(1, writer_threads):      # bound the loader items kept alive by queued work
                pending.pop(0).result()
        for f in pending:
            f.result()
    assert j == len(counts), "loader yielded %d images, %d tables given" % (j, len(counts))


def RegionSelection(cfg, feature_extractor, classifier, tgt_epoch_loader, round_number, *, in_flight=8, writer_threads=None,
                    streams=4, return_tables=False, lowres_mode=None, stats=None, mask_staging=None, write_files=True):
    """Drop-in for build.py:71-186: same positional arguments, same files written (uint8 mode-L PNG mask
    at path_to_mask, torch.save({'active','selected'}) at path_to_indicator), models left in train mode,
    every file on disk when the call returns.  Returns None like the reference, or -- keyword-only
    `return_tables=True`, used by halo_amd.pool.region_selection_sharded -- the per-image pick tables
    [(picks (n,3) float64 rows (h, w, score), count)] in loader order.

    Inside the images of the pool are independent, so a loader batch's host->device staging,
    score + selection (ONE launch group for the whole batch when its images share a label size; the reference's loader has
    batch size 1) and device->host copies are enqueued on one of `streams` side streams (round
    robin) while the backbone processes the next batch on the caller's stream; the host thread never waits
    for the GPU: a pool of writer threads waits for each batch's event, encodes the PNGs and writes the
    indicators.  At most `in_flight` batches are between "launched" and "copied back to the host" (bounds device
    and pinned memory; each slot owns its staging buffers and reuses them from round to round); `in_flight=0` runs strictly
    one batch at a time like the reference.  `writer_threads` defaults to min(8, usable host cores / LOCAL_WORLD_SIZE).
    `lowres_mode`: 'exact' (default; environment HALO_LOWRES) interpolates every channel and is
    bit-identical to upsample-then-score, the reference's order; 'gram' (opt-in, float64 embeddings) evaluates the radius through
    per-cell Gram terms (floating_region.score_maps_lowres: bit-identical to its oracle twin, squared norms within 1.3e-10 of
    the exact order, the reference's files on every test vector -- but not the reference's evaluation order).
    `stats`: an optional dict that receives where the host time went (seconds per phase, main thread and writers);
    `mask_staging`: "table" (default; environment HALO_MASK_STAGING) or "device" -- whether the int64 mask / label maps travel to
    the GPU at all (see _launch): same files either way.
    `write_files=False` (with return_tables): score and select only -- halo_amd.pool.region_selection_sharded's global-budget mode
    writes the files afterwards from the KEPT picks (persist_from_tables)."""
    import queue
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    prm = AcquisitionParams(cfg)
    dev = torch.device("cuda", torch.cuda.current_device())
    mask_staging = os.environ.get("HALO_MASK_STAGING", "table") if mask_staging is None else mask_staging
    if mask_staging not in ("table", "device"):
        raise ValueError("mask_staging must be 'table' or 'device', got %r" % (mask_staging,))
    if writer_threads is None:
        # this rank's share of the usable host cores (8 ranks on a 16-core quota: 2 writers each, not 8 x 8 threads)
        from ..._host import host_threads_per_rank
        writer_threads = host_threads_per_rank(cap=8)
    depth = max(1, in_flight)
    side = _side_streams(dev, max(1, min(streams, depth)))
    # the slots (and their pinned / device staging buffers) live as long as the side streams: the next round reuses them
    have = _SLOTS.setdefault(dev.index, [])
    while len(have) < depth:
        have.append(_Slot(side[len(have) % len(side)]))
    slots = queue.Queue()
    for k in range(depth):
        slots.put(have[k])
    if stats is not None:
        for key in ("main_forward_s", "main_stage_s", "main_launch_s", "main_wait_slot_s", "main_loader_s",
                    "writer_event_wait_s", "writer_copy_s", "writer_png_s", "writer_save_s"):
            stats[key] = 0.0
        stats.update(images=0, batches=0, writer_threads=int(writer_threads), in_flight=depth, streams=len(side), lock=threading.Lock())
    feature_extractor.eval()
    classifier.eval()
    moved = False
    pending = []
    # a slot returns to the launcher as soon as a writer holds the image's pick table, so the slots no longer bound how far the
    # launches run ahead of the files: this does
    backlog = threading.BoundedSemaphore(depth + 3 * max(1, writer_threads))
    with ThreadPoolExecutor(max_workers=max(1, writer_threads)) as writers, torch.no_grad():
        try:
            t_prev = time.perf_counter()
            for batch in tgt_epoch_loader:
                t_a = time.perf_counter()
                images = batch["img"].to(dev, non_blocking=True)
                if not moved:
                    feature_extractor.to(dev)
                    classifier.to(dev)
                    moved = True
                logits_lr, embed_lr = classifier(feature_extractor(images), size=images.shape[-2:])
                t_b = time.perf_counter()
                nb = len(batch["origin_mask"])                      # loader batch size: 1 in the reference
                sizes = [(int(batch["size"][i][0]), int(batch["size"][i][1])) for i in range(nb)]
                # images of one label size go through as ONE launch group; otherwise one image at a time
                groups = [(0, nb)] if all(sz == sizes[0] for sz in sizes) else [(i, i + 1) for i in range(nb)]
                for lo, hi in groups:
                    t_d = time.perf_counter()
                    # blocks only while `in_flight` batches hold every slot (a slot returns when a writer thread has taken its
                    # batch out of the pinned buffers, so slow writers hold the launches back too: nothing piles up)
                    slot = slots.get()
                    t_e = time.perf_counter()
                    try:
                        rec = _launch(prm, logits_lr[lo:hi], embed_lr[lo:hi], sizes[lo], batch["origin_mask"][lo:hi],
                                      batch["origin_label"][lo:hi], batch["active"][lo:hi], batch["selected"][lo:hi], dev, slot,
                                      lowres_mode, stats, mask_staging, write_files)
                    except BaseException:
                        slots.put(slot)
                        raise
                    rec.left, rec.lock = hi - lo, threading.Lock()
                    for i in range(lo, hi):
                        backlog.acquire()                           # bounds the images (38 MB of loader tensors each) awaiting their files
                        pending.append(writers.submit(_finish, rec, i - lo, (batch["path_to_mask"][i], batch["path_to_indicator"][i]),
                                                      slots, stats, backlog))
                    if in_flight <= 0:
                        for f in pending[-(hi - lo):]:
                            f.result()
                    if stats is not None:
                        stats["main_wait_slot_s"] += t_e - t_d
                        stats["images"] += hi - lo
                        stats["batches"] += 1
                if stats is not None:
                    stats["main_loader_s"] += t_a - t_prev
                    stats["main_forward_s"] += t_b - t_a
                t_prev = time.perf_counter()
        finally:
            results = [f.result() for f in pending]                  # surface I/O errors; all files are on disk
    if stats is not None:
        stats.pop("lock", None)
    feature_extractor.train()
    classifier.train()
    return results if return_tables else None
