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
                  return_picks=True, method=None, out=None, score_range=None):
    """Batched device-side selection.  score (B,H,W) f32|f64, active/selected (B,H,W) bool,
    active_mask/ground_truth (B,H,W) int64 -- all on one ROCm device, all mutated in place.
    Returns (picks (B,n,3) float64 rows (h, w, value), n_picked (B,) int32) or None.
    method: "auto" (default; environment HALO_SELECT overrides) = value-binned sweep with the serial kernel
    behind it, "serial" = the tile-table kernel only, "binned" = the sweep or HaloUnsupported.  Same results.
    out: optional (picks (B,n,3) float64, n_picked (B,) int32) contiguous device tensors to write the tables into
    (pipelined callers collect a whole round's tables in one buffer); rows past an image's count are left as they are.
    score_range: the records score_maps / score_maps_lowres filled for these maps (new_score_range): the sweep then skips its
    pass over the map for the value range (the records only have to bound the values: the picks do not depend on them)."""
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
    if n == 0 or B == 0:
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
    rc = L.halo_greedy_select_ranged(_lib.ptr(score), _lib.dtype_code(score), B, H, W, n, int(active_radius),
                                     int(mask_radius), _lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask),
                                     _lib.ptr(ground_truth), _lib.ptr(picks), _lib.ptr(n_picked), _lib.ptr(ws), ws.numel(),
                                     method, _lib.ptr(score_range), _lib.stream_ptr(dev))
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
    """One image whose staging, scoring, selection and device->host copies have been enqueued on a side stream."""
    __slots__ = ("done", "path_mask", "path_indicator", "keep", "picks", "npk", "h_mask", "h_active", "h_selected", "slot")


class _Slot:
    """One pipeline slot: a side stream position plus its own pinned device->host staging buffers, allocated once
    per image shape and reused for every image that passes through the slot (pinning host pages costs
    milliseconds; the caching host allocator gives no guarantee to hand the same block back in time)."""

    def __init__(self, stream):
        self.stream = stream
        self.shape = None
        self.mask = self.active = self.selected = None

    def buffers(self, shape):
        if self.shape != tuple(shape):
            self.mask = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
            self.active = torch.empty(shape, dtype=torch.bool, pin_memory=True)
            self.selected = torch.empty(shape, dtype=torch.bool, pin_memory=True)
            self.shape = tuple(shape)
        return self.mask, self.active, self.selected


_SIDE = {}
_QUEUE_WARNED = False


def _check_hw_queues(n_streams):
    """Tell the user once that the acquisition alone measured faster on two hardware queues than on ROCm's default of four, when
    nobody chose a value.  Only a hint: the setting is process-wide (it also governs the training iterations' streams), so the
    package never sets it by itself -- halo_amd.configure(hw_queues=2), INTEGRATION.md section 3."""
    global _QUEUE_WARNED
    if _QUEUE_WARNED or "GPU_MAX_HW_QUEUES" in os.environ:
        return
    _QUEUE_WARNED = True
    import warnings
    warnings.warn("halo_amd RegionSelection drives %d side streams beside the caller's and GPU_MAX_HW_QUEUES is unset (ROCm's "
                  "default: 4 hardware queues).  The acquisition alone measured 2-12 %% faster on 2; if that suits the training "
                  "process too, call halo_amd.configure(hw_queues=2) (or export the variable) before the first HIP call "
                  "(INTEGRATION.md section 3)." % n_streams, RuntimeWarning, stacklevel=3)


def _side_streams(dev, n):
    """The acquisition's side streams, created once per device and reused by every round: the scorer's and the selector's
    scratch buffers are cached per stream (floating_region._workspace), so fresh streams per call would strand them."""
    have = _SIDE.setdefault(dev.index, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(dev, priority=-1))
    return have[:n]


def _launch_one(prm, logit_lr, embed_lr, size, origin_mask, origin_label, active_cpu, selected_cpu, dev, slot, lowres_mode=None):
    """Enqueue one image of the pool (build.py:113-166) on `stream`: stage its masks, score -> mask -> select,
    copy the results back into pinned host buffers.  Fully asynchronous: `rec.done` fires when the host
    buffers hold the image's final mask / indicator maps."""
    rec = _InFlight()
    rec.slot = slot
    stream = slot.stream
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))             # the head outputs are complete from here on
    with torch.cuda.stream(stream):
        stream.wait_event(ready)
        amask = origin_mask.to(dev, non_blocking=True).long().contiguous()
        gt = origin_label.to(dev, non_blocking=True).long().contiguous()
        active = active_cpu.to(dev, non_blocking=True).bool().contiguous()
        selected = selected_cpu.to(dev, non_blocking=True).bool().contiguous()
        # the two F.interpolate(align_corners=True) calls of build.py:122-135 are fused into the scorer:
        # the C x H x W float64 embedding (4.3 GB at C=256) is never written or read
        rec.picks, rec.npk = acquire_batch_lowres(
                             logit_lr, embed_lr, size, gt[None], active[None], selected[None], amask[None],
                             unc_type=prm.unc, pur_type=prm.pur, normalize=prm.normalize,
                             n_regions=prm.regions(size[0] * size[1]), active_radius=prm.radius,
                             mask_radius=prm.mask_radius, ksize=prm.scorer.size, purity_size=prm.scorer.purity_size,
                             K=prm.K, c=prm.scorer.mapper.c, lowres_mode=lowres_mode)
        # uint8 on the device first: 2 MB instead of 16 MB over PCIe per 1024x2048 mask (same values as the
        # reference's cast-after-copy, build.py:67-68,162)
        rec.h_mask, rec.h_active, rec.h_selected = slot.buffers(amask.shape)
        rec.h_mask.copy_(amask.to(torch.uint8), non_blocking=True)
        rec.h_active.copy_(active, non_blocking=True)
        rec.h_selected.copy_(selected, non_blocking=True)
        rec.done = torch.cuda.Event(blocking=True)           # the writer thread sleeps on it instead of spinning
        rec.done.record(stream)
    # device tensors the side stream still reads: kept alive until the image is retired
    rec.keep = (logit_lr, embed_lr, gt, amask, active, selected)
    return rec


_PNG_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def write_png_gray8(path, arr):
    """8-bit greyscale (PIL mode "L") PNG of a (H, W) uint8 array: the file `Image.fromarray(a).save(path)` of
    build.py:163-164 produces, pixel for pixel, written directly -- filter type 0 on every row and ONE zlib stream
    with the run-length strategy (masks are long runs of 255 with small labelled windows; zlib releases the GIL, so
    writer threads run in parallel).  PIL spends ~12 ms per 1024x2048 mask in its row-by-row encoder; this takes
    about half and decodes to the identical image."""
    h, w = arr.shape
    rows = np.empty((h, w + 1), dtype=np.uint8)
    rows[:, 0] = 0                                                   # filter type "None" per scanline
    rows[:, 1:] = arr
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)              # bit depth 8, colour type 0 (greyscale), no interlace
    z = zlib.compressobj(1, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    idat = z.compress(rows.tobytes()) + z.flush()
    with open(path, "wb") as f:
        f.write(_PNG_SIGNATURE + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", idat) + _png_chunk(b"IEND", b""))


def _persist(mask_np, active, selected, path_mask, path_indicator):
    """build.py:162-166: uint8 mode-L PNG + torch.save'd indicator dict (what cityscapes.py:234-251 reads back)."""
    if mask_np.ndim == 2 and mask_np.dtype == np.uint8 and mask_np.size and str(path_mask).lower().endswith(".png"):
        write_png_gray8(path_mask, mask_np)
    else:
        Image.fromarray(mask_np).save(path_mask)
    torch.save({"active": active, "selected": selected}, path_indicator)


def _finish(rec, slots, backlog):
    """Writer-thread half of one image: wait for its copies, hand its pipeline slot back, write the two files."""
    try:
        return _finish_inner(rec, slots)
    finally:
        backlog.release()


def _finish_inner(rec, slots):
    try:
        rec.done.synchronize()
        # ONE streaming copy out of the pinned staging buffers first (the slot is then free for the next image; the
        # encoders make several passes over their input, and the indicator must hold plain tensors as from `.cpu()`).
        # numpy copies on purpose: a torch CPU op here would wake an intra-op thread pool as wide as the host
        mask = rec.h_mask.numpy().copy()
        active, selected = torch.from_numpy(rec.h_active.numpy().copy()), torch.from_numpy(rec.h_selected.numpy().copy())
        out = (rec.picks[0], int(rec.npk[0]))
        rec.keep = rec.h_mask = rec.h_active = rec.h_selected = None
    finally:
        slots.put(rec.slot)
    _persist(mask, active, selected, rec.path_mask, rec.path_indicator)
    return out


def RegionSelection(cfg, feature_extractor, classifier, tgt_epoch_loader, round_number, *, in_flight=8, writer_threads=None,
                    streams=4, return_tables=False, lowres_mode=None):
    """Drop-in for build.py:71-186: same positional arguments, same files written (uint8 mode-L PNG mask
    at path_to_mask, torch.save({'active','selected'}) at path_to_indicator), models left in train mode,
    every file on disk when the call returns.  Returns None like the reference, or -- keyword-only
    `return_tables=True`, used by halo_amd.pool.region_selection_sharded -- the per-image pick tables
    [(picks (n,3) float64 rows (h, w, score), count)] in loader order.

    Inside the images of the pool are independent, so image i's host->device staging,
    score + selection and device->host copies are enqueued on one of `streams` side streams (round
    robin) while the backbone processes image i+1 on the caller's stream; the host thread never waits
    for the GPU: a pool of writer threads waits for each image's event, encodes the PNG and writes the
    indicator.  At most `in_flight` images are between "launched" and "copied back to the host" (bounds device
    and pinned memory; each slot owns its pinned staging buffers); `in_flight=0` runs strictly one image at a
    time like the reference.  `writer_threads` defaults to min(8, usable host cores / LOCAL_WORLD_SIZE).  `lowres_mode`: 'exact' (default; environment HALO_LOWRES) interpolates every channel and is
    bit-identical to upsample-then-score, the reference's order; 'gram' (opt-in, float64 embeddings) evaluates the radius through
    per-cell Gram terms (floating_region.score_maps_lowres: bit-identical to its oracle twin, squared norms within 1.3e-10 of
    the exact order, the reference's files on every test vector -- but not the reference's evaluation order)."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    prm = AcquisitionParams(cfg)
    dev = torch.device("cuda", torch.cuda.current_device())
    if writer_threads is None:
        # this rank's share of the usable host cores (8 ranks on a 16-core quota: 2 writers each, not 8 x 8 threads)
        from ..._host import host_threads_per_rank
        writer_threads = host_threads_per_rank(cap=8)
    depth = max(1, in_flight)
    side = _side_streams(dev, max(1, min(streams, depth)))
    _check_hw_queues(len(side))
    backlog = threading.Semaphore(depth + 4 * max(1, writer_threads))   # images whose files are not on disk yet (host copies)
    slots = queue.Queue()
    for k in range(depth):
        slots.put(_Slot(side[k % len(side)]))
    feature_extractor.eval()
    classifier.eval()
    moved = False
    pending = []
    with ThreadPoolExecutor(max_workers=max(1, writer_threads)) as writers, torch.no_grad():
        try:
            for batch in tgt_epoch_loader:
                images = batch["img"].to(dev, non_blocking=True)
                if not moved:
                    feature_extractor.to(dev)
                    classifier.to(dev)
                    moved = True
                logits_lr, embed_lr = classifier(feature_extractor(images), size=images.shape[-2:])
                for i in range(len(batch["origin_mask"])):          # loader batch size is 1 in the reference
                    size = (int(batch["size"][i][0]), int(batch["size"][i][1]))
                    backlog.acquire()
                    slot = slots.get()                               # blocks only while `in_flight` images hold every slot
                    try:
                        rec = _launch_one(prm, logits_lr[i:i + 1], embed_lr[i:i + 1], size, batch["origin_mask"][i],
                                          batch["origin_label"][i], batch["active"][i], batch["selected"][i], dev, slot,
                                          lowres_mode)
                    except BaseException:
                        slots.put(slot)
                        backlog.release()
                        raise
                    rec.path_mask, rec.path_indicator = batch["path_to_mask"][i], batch["path_to_indicator"][i]
                    pending.append(writers.submit(_finish, rec, slots, backlog))
                    if in_flight <= 0:
                        pending[-1].result()
        finally:
            results = [f.result() for f in pending]                  # surface I/O errors; all files are on disk
    feature_extractor.train()
    classifier.train()
    return results if return_tables else None
