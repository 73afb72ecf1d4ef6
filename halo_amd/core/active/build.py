"""RegionSelection / select_pixels_to_label on HIP kernels -- host mirror of core/active/build.py.

`select_pixels_to_label` keeps the reference's signature and in-place semantics (build.py:27-64)
but runs the whole greedy loop on the device (halo_amd/csrc/halo_select.hip): zero host syncs
instead of >= 3 `.item()` per region.  `RegionSelection` keeps the reference's signature and its
on-disk side effects (uint8 mode-L PNG mask + torch.save'd {'active','selected'} indicator,
build.py:162-166) and batches images through the fused score -> mask -> select pipeline.
"""
import math
import os

import numpy as np
import torch
from PIL import Image

from ... import _lib
from ..configs import cfg
from ..utils.hyperbolic import HyperMapper, bilinear_align_corners
from .floating_region import FloatingRegionScore, score_maps, score_maps_lowres, _workspace


def greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth,
                  return_picks=True, method=None):
    """Batched device-side selection.  score (B,H,W) f32|f64, active/selected (B,H,W) bool,
    active_mask/ground_truth (B,H,W) int64 -- all on one ROCm device, all mutated in place.
    Returns (picks (B,n,3) float64 rows (h, w, value), n_picked (B,) int32) or None.
    method: "auto" (default; environment HALO_SELECT overrides) = value-binned sweep with the serial kernel
    behind it, "serial" = the tile-table kernel only, "binned" = the sweep or HaloUnsupported.  Same results."""
    dev = _lib.require_device(score, active, selected, active_mask, ground_truth)
    B, H, W = score.shape
    for t in (score, active, selected, active_mask, ground_truth):
        assert t.shape == (B, H, W) and t.is_contiguous(), "greedy_select expects contiguous (B,H,W) tensors"
    assert active.dtype == torch.bool and selected.dtype == torch.bool
    assert active_mask.dtype == torch.int64 and ground_truth.dtype == torch.int64
    n = int(max(0, min(int(n_regions), H * W)))
    picks = n_picked = None
    if return_picks:
        picks = torch.zeros((B, max(n, 1), 3), dtype=torch.float64, device=dev)
        n_picked = torch.zeros((B,), dtype=torch.int32, device=dev)
    if n == 0 or B == 0:
        return (picks[:, :0], n_picked) if return_picks else None
    L = _lib.lib()
    method = _lib.SELECT[method or os.environ.get("HALO_SELECT", "auto")]
    nws = L.halo_select_workspace_bytes(B, H, W, n, int(mask_radius)) if method != _lib.SELECT["serial"] else 256
    ws = _workspace(dev, nws, "select")
    rc = L.halo_greedy_select(_lib.ptr(score), _lib.dtype_code(score), B, H, W, n, int(active_radius),
                              int(mask_radius), _lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask),
                              _lib.ptr(ground_truth), _lib.ptr(picks), _lib.ptr(n_picked), _lib.ptr(ws), ws.numel(),
                              method, _lib.stream_ptr(dev))
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
    score, _, _ = score_maps(logit, decoder_out, unc_type, pur_type, normalize, ground_truth, size=size,
                             purity_size=purity_size, K=K, c=c, active=active, want_maps=False)
    return greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth)


def acquire_batch_lowres(logit_lr, decoder_lr, size, ground_truth, active, selected, active_mask, *, unc_type, pur_type,
                         normalize, n_regions, active_radius, mask_radius, ksize=None, purity_size=None, K=100, c=1.0):
    """build.py:122-160 for B images: resize of the head outputs fused into the scorer (the C x H x W
    embedding is never materialised), then mask + greedy selection.  Falls back to explicit HIP
    upsampling when the fused kernel declines the geometry (strong downsampling)."""
    ksize = 2 * active_radius + 1 if ksize is None else ksize
    try:
        score, _, _ = score_maps_lowres(logit_lr, decoder_lr, size, unc_type, pur_type, normalize, ground_truth,
                                        ksize=ksize, purity_size=purity_size, K=K, c=c, active=active, want_maps=False)
    except _lib.HaloUnsupported:
        logit = bilinear_align_corners(logit_lr.float(), size)
        dec = bilinear_align_corners(decoder_lr, size) if pur_type in ("hyper", "radius", "euc_norm") else decoder_lr
        score, _, _ = score_maps(logit, dec, unc_type, pur_type, normalize, ground_truth, size=ksize,
                                 purity_size=purity_size, K=K, c=c, active=active, want_maps=False)
    return greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth)


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
    """One image whose scoring + selection has been enqueued on the acquisition stream."""
    __slots__ = ("amask", "active", "selected", "done", "path_mask", "path_indicator", "keep", "picks", "npk")


def _launch_one(prm, logit_lr, embed_lr, size, origin_mask, origin_label, active_cpu, selected_cpu, dev, stream):
    """Enqueue one image of the pool (build.py:113-160) on `stream`: stage its masks, then
    score -> mask -> select.  Asynchronous; the caller retires it with _retire()."""
    rec = _InFlight()
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))             # the head outputs are complete from here on
    with torch.cuda.stream(stream):
        stream.wait_event(ready)
        rec.amask = origin_mask.to(dev, non_blocking=True).long().contiguous()
        gt = origin_label.to(dev, non_blocking=True).long().contiguous()
        rec.active = active_cpu.to(dev, non_blocking=True).bool().contiguous()
        rec.selected = selected_cpu.to(dev, non_blocking=True).bool().contiguous()
        # the two F.interpolate(align_corners=True) calls of build.py:122-135 are fused into the scorer:
        # the C x H x W float64 embedding (4.3 GB at C=256) is never written or read
        rec.picks, rec.npk = acquire_batch_lowres(
                             logit_lr, embed_lr, size, gt[None], rec.active[None], rec.selected[None], rec.amask[None],
                             unc_type=prm.unc, pur_type=prm.pur, normalize=prm.normalize,
                             n_regions=prm.regions(size[0] * size[1]), active_radius=prm.radius,
                             mask_radius=prm.mask_radius, ksize=prm.scorer.size, purity_size=prm.scorer.purity_size,
                             K=prm.K, c=prm.scorer.mapper.c)
        rec.done = torch.cuda.Event()
        rec.done.record(stream)
    rec.keep = (logit_lr, embed_lr, gt)                      # alive until the side stream is done with them
    return rec


def _persist(mask_np, active, selected, path_mask, path_indicator):
    """build.py:162-166: uint8 mode-L PNG + torch.save'd indicator dict (what cityscapes.py:234-251 reads back)."""
    Image.fromarray(mask_np).save(path_mask)
    torch.save({"active": active, "selected": selected}, path_indicator)


def _retire(rec, writers, pending, tables=None):
    rec.done.synchronize()
    if tables is not None:
        tables.append((rec.picks[0], int(rec.npk[0])))
    # uint8 on the device first: 2 MB instead of 16 MB over PCIe per 1024x2048 mask (same values as the
    # reference's cast-after-copy, build.py:67-68,162)
    job = (rec.amask.to(torch.uint8).cpu().numpy(), rec.active.cpu(), rec.selected.cpu(), rec.path_mask, rec.path_indicator)
    rec.keep = None
    pending.append(writers.submit(_persist, *job))


def RegionSelection(cfg, feature_extractor, classifier, tgt_epoch_loader, round_number, *, in_flight=3, writer_threads=4,
                    return_tables=False):
    """Drop-in for build.py:71-186: same positional arguments, same files written (uint8 mode-L PNG mask
    at path_to_mask, torch.save({'active','selected'}) at path_to_indicator), models left in train mode,
    every file on disk when the call returns.  Returns None like the reference, or -- keyword-only
    `return_tables=True`, used by halo_amd.pool.region_selection_sharded -- the per-image pick tables
    [(picks (n,3) float64 rows (h, w, score), count)] in loader order.

    Inside, image i's score + greedy selection runs on a side stream while the backbone processes
    image i+1, and PNG encoding / torch.save run on a small thread pool: in the reference
    both serialise with the 2331-step selection loop of every image."""
    from concurrent.futures import ThreadPoolExecutor
    prm = AcquisitionParams(cfg)
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(dev, priority=-1)
    feature_extractor.eval()
    classifier.eval()
    moved = False
    queue, pending = [], []
    tables = [] if return_tables else None
    with ThreadPoolExecutor(max_workers=max(1, writer_threads)) as writers, torch.no_grad():
        for batch in tgt_epoch_loader:
            images = batch["img"].to(dev, non_blocking=True)
            if not moved:
                feature_extractor.to(dev)
                classifier.to(dev)
                moved = True
            logits_lr, embed_lr = classifier(feature_extractor(images), size=images.shape[-2:])
            for i in range(len(batch["origin_mask"])):          # loader batch size is 1 in the reference
                size = (int(batch["size"][i][0]), int(batch["size"][i][1]))
                rec = _launch_one(prm, logits_lr[i:i + 1], embed_lr[i:i + 1], size, batch["origin_mask"][i],
                                  batch["origin_label"][i], batch["active"][i], batch["selected"][i], dev, side)
                rec.path_mask, rec.path_indicator = batch["path_to_mask"][i], batch["path_to_indicator"][i]
                queue.append(rec)
                while len(queue) > max(0, in_flight):      # in_flight=0: fully serial, like the reference
                    _retire(queue.pop(0), writers, pending, tables)
        while queue:
            _retire(queue.pop(0), writers, pending, tables)
        for f in pending:
            f.result()                                           # surface I/O errors; all files are on disk
    feature_extractor.train()
    classifier.train()
    return tables
