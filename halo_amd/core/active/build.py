"""RegionSelection / select_pixels_to_label on HIP kernels -- host mirror of core/active/build.py.

`select_pixels_to_label` keeps the reference's signature and in-place semantics (build.py:27-64)
but runs the whole greedy loop on the device (halo_amd/csrc/halo_select.hip): zero host syncs
instead of >= 3 `.item()` per region.  `RegionSelection` keeps the reference's signature and its
on-disk side effects (uint8 mode-L PNG mask + torch.save'd {'active','selected'} indicator,
build.py:162-166) and batches images through the fused score -> mask -> select pipeline.
"""
import math

import numpy as np
import torch
from PIL import Image

from ... import _lib
from ..configs import cfg
from ..utils.hyperbolic import bilinear_align_corners
from .floating_region import FloatingRegionScore, score_maps, score_maps_lowres, _workspace


def greedy_select(score, n_regions, active_radius, mask_radius, active, selected, active_mask, ground_truth,
                  return_picks=True):
    """Batched device-side selection.  score (B,H,W) f32|f64, active/selected (B,H,W) bool,
    active_mask/ground_truth (B,H,W) int64 -- all on one ROCm device, all mutated in place.
    Returns (picks (B,n,3) float64 rows (h, w, value), n_picked (B,) int32) or None."""
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
    if n == 0:
        return (picks[:, :0], n_picked) if return_picks else None
    L = _lib.lib()
    nws = L.halo_select_workspace_bytes(B, H, W)
    ws = _workspace(dev, nws, "select")
    rc = L.halo_greedy_select(_lib.ptr(score), _lib.dtype_code(score), B, H, W, n, int(active_radius),
                              int(mask_radius), _lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask),
                              _lib.ptr(ground_truth), _lib.ptr(picks), _lib.ptr(n_picked), _lib.ptr(ws), ws.numel(),
                              _lib.stream_ptr(dev))
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


def needs_decoder_out(cfg_, uncertainty_type, purity_type):
    """build.py:127-131."""
    return (uncertainty_type in ["certainty", "hyperbolic"]
            or (purity_type in ["hyper", "radius", "euc_norm"])
            or (uncertainty_type == "none" and cfg_.MODEL.HYPER))


def RegionSelection(cfg, feature_extractor, classifier, tgt_epoch_loader, round_number):
    feature_extractor.eval()
    classifier.eval()

    per_region_pixels = (2 * cfg.ACTIVE.RADIUS_K + 1) ** 2
    active_radius = cfg.ACTIVE.RADIUS_K
    mask_radius = cfg.ACTIVE.MASK_RADIUS_K
    active_budget = cfg.ACTIVE.BUDGET / len(cfg.ACTIVE.SELECT_ITER)
    uncertainty_type = cfg.ACTIVE.UNCERTAINTY
    purity_type = cfg.ACTIVE.PURITY
    K = cfg.ACTIVE.K

    floating_region_score = FloatingRegionScore(
        in_channels=cfg.MODEL.NUM_CLASSES, size=2 * active_radius + 1, purity_type=purity_type, K=K)
    if purity_type not in _lib.PUR:
        raise NotImplementedError("Error: purity type '{}' not implemented".format(purity_type))
    floating_region_score._check_purity_channels(purity_type)
    dev = torch.device("cuda", torch.cuda.current_device())

    with torch.no_grad():
        idx = 0
        for tgt_data in tgt_epoch_loader:
            tgt_input, path2mask = tgt_data["img"], tgt_data["path_to_mask"]
            origin_mask, origin_label = tgt_data["origin_mask"], tgt_data["origin_label"]
            origin_size = tgt_data["size"]
            active_indicator = tgt_data["active"]
            selected_indicator = tgt_data["selected"]
            path2indicator = tgt_data["path_to_indicator"]

            tgt_input = tgt_input.to(dev, non_blocking=True)
            if idx == 0:
                feature_extractor.to(tgt_input.device)
                classifier.to(tgt_input.device)

            tgt_size = tgt_input.shape[-2:]
            tgt_feat = feature_extractor(tgt_input)
            tgt_out, decoder_out = classifier(tgt_feat, size=tgt_size)

            for i in range(len(origin_mask)):
                active_mask = origin_mask[i].to(dev, non_blocking=True).long().contiguous()
                ground_truth = origin_label[i].to(dev, non_blocking=True).long().contiguous()
                size = (int(origin_size[i][0]), int(origin_size[i][1]))
                num_pixel_cur = size[0] * size[1]
                active = active_indicator[i].to(dev).bool().contiguous()
                selected = selected_indicator[i].to(dev).bool().contiguous()

                active_regions = math.ceil(num_pixel_cur * active_budget / per_region_pixels)  # build.py:148-150
                # build.py:122-144: the two F.interpolate(align_corners=True) calls are fused into the
                # scorer, so the C x H x W float64 embedding (4.3 GB at C=256) is never written or read
                acquire_batch_lowres(tgt_out[i:i + 1], decoder_out[i:i + 1], size, ground_truth[None], active[None],
                                     selected[None], active_mask[None], unc_type=uncertainty_type,
                                     pur_type=purity_type, normalize=cfg.ACTIVE.NORMALIZE, n_regions=active_regions,
                                     active_radius=active_radius, mask_radius=mask_radius,
                                     ksize=floating_region_score.size, purity_size=floating_region_score.purity_size,
                                     K=K, c=floating_region_score.mapper.c)

                active_mask_np = to_np_array(active_mask)                                  # build.py:162-166
                Image.fromarray(active_mask_np).save(path2mask[i])
                indicator = {"active": active.cpu(), "selected": selected.cpu()}
                torch.save(indicator, path2indicator[i])
            idx += 1

    feature_extractor.train()
    classifier.train()
