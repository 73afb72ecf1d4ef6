"""Hyperbolic heads -- the part of core/models/classifier.py that is on the hot path.

ASPP_Classifier_V2_Hyper.forward (classifier.py:364-379) and DepthwiseSeparableASPP_Hyper.forward
(classifier.py:552-558) both end with

    embed = mapper.expmap(feat, dim=1)                 # float64
    out   = conv_seg(embed.double()).float()           # HyperMLR
    [bilinear(align_corners=True) of out (v3+) or of out AND embed (v2)]
    return out, embed

`hyper_head_tail` is that tail on HIP kernels.  The convolutional bodies in front of it stay on
PyTorch-ROCm/MIOpen (out of scope).  Two ways to reach the tail from the reference's
`forward(x: dict{'out','low'}, size=None) -> (out, embed)` interface:

  * `ASPP_Classifier_V2_Hyper` below is a complete drop-in class (its body is a sum of dilated convs);
  * `v2_hyper_forward` / `v3plus_hyper_forward` are `forward` replacements for the reference's own head
    classes -- they run the instance's own conv modules, then the HIP tail.  `halo_amd.install()`
    binds them onto core.models.classifier's classes, so checkpoints, constructors and the
    `build_classifier` factory (core/models/build.py) stay the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..configs import cfg
from ..utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners, head_tail_fused


def hyper_head_tail(feat, mapper: HyperMapper, conv_seg: HyperMLR, size=None, resize_embed=False):
    """feat (B,C,h,w) float32 from conv_reduce / the ASPP sum -> (out float32, embed float64).

    resize_embed=False: DeepLab-v3+ tail (classifier.py:552-558); True: DeepLab-v2 tail, which also
    resizes the embedding (classifier.py:375-377)."""
    training = torch.is_grad_enabled() and (feat.requires_grad or conv_seg.P_MLR.requires_grad)
    if training:
        # expmap and HyperMLR carry HIP backward kernels; the resize stays on F.interpolate, which autograd
        # already differentiates (it is outside the kernels' scope under training)
        embed = mapper.expmap(feat, dim=1)
        out = conv_seg._hyper_logits(embed, out_dtype=torch.float32)      # = conv_seg(embed).float(), the cast fused into the kernel's store
        if size is not None:
            out = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
            if resize_embed:
                embed = F.interpolate(embed, size=size, mode="bilinear", align_corners=True)
        return out, embed
    with torch.no_grad():
        # the heads' own shape (64 channels): expmap -> HyperMLR -> .float() in ONE kernel, the embedding never re-read; any other
        # shape: the two calls (same bits either way)
        fused = head_tail_fused(feat, conv_seg.P_MLR, conv_seg.A_MLR, conv_seg.c) if (feat.is_cuda and feat.dim() == 4 and float(mapper.c) == float(conv_seg.c)) else None
        if fused is not None:
            out, embed = fused
        else:
            embed = mapper.expmap(feat, dim=1)
            out = conv_seg._hyper_logits(embed, out_dtype=torch.float32)
        if size is not None:
            out = bilinear_align_corners(out, size)
            if resize_embed:
                embed = bilinear_align_corners(embed, size)
    return out, embed


def _tail_modules(head):
    """(mapper, conv_seg) of a head instance as HIP-backed objects.  A reference-built head holds the
    reference's HyperMapper / HyperMLR (geoopt-backed): same curvature and the SAME parameter tensors are
    re-used, so optimiser state and checkpoints are unaffected."""
    tail = head.__dict__.get("_halo_tail")
    if tail is None or tail[1].P_MLR is not head.conv_seg.P_MLR or tail[1].A_MLR is not head.conv_seg.A_MLR:
        mapper = head.mapper if isinstance(head.mapper, HyperMapper) else HyperMapper(c=head.mapper.c)
        seg = head.conv_seg
        if not isinstance(seg, HyperMLR):
            mlr = HyperMLR.__new__(HyperMLR)
            nn.Module.__init__(mlr)
            mlr.c, mlr.K, mlr.num_classes = seg.c, seg.K, seg.num_classes
            mlr.P_MLR, mlr.A_MLR = seg.P_MLR, seg.A_MLR          # shared Parameters, not copies
            seg = mlr
        tail = (mapper, seg)
        head.__dict__["_halo_tail"] = tail                        # not a registered submodule: state_dict unchanged
    return tail


def v2_hyper_forward(self, x, size=None):
    """forward of ASPP_Classifier_V2_Hyper (classifier.py:364-379): sum of the dilated 3x3 branches, HIP tail;
    DeepLab-v2 resizes the logits AND the embedding."""
    feat = x["out"]
    branches = iter(self.conv2d_list)
    embed = next(branches)(feat)
    for conv in branches:
        embed = embed + conv(feat)
    mapper, seg = _tail_modules(self)
    return hyper_head_tail(embed, mapper, seg, size=size, resize_embed=True)


def v3plus_hyper_forward(self, x, size=None):
    """forward of DepthwiseSeparableASPP_Hyper (classifier.py:486-558): the instance's own ASPP / decoder
    modules (PyTorch), optional weighted normalisation (`wn_mlp`, HFR), then the HIP tail."""
    low, top = x["low"], x["out"]
    pyramid = [branch(top) for branch in self.parallel_branches]
    pooled = self.global_branch(top)
    pyramid.append(F.interpolate(pooled, size=top.shape[2:], mode="bilinear", align_corners=True))
    fused = self.bottleneck(torch.cat(pyramid, dim=1))
    fused = F.interpolate(fused, size=low.shape[2:], mode="bilinear", align_corners=True)
    dec = self.decoder(torch.cat([fused, self.shortcut(low)], dim=1))
    dec = self.conv_reduce(dec)
    if getattr(self, "wn_mlp", None) is not None:                      # classifier.py:531-550
        b, ch, h, w = dec.shape
        weights = self.wn_mlp(dec.permute(0, 2, 3, 1).reshape(-1, ch)).view(b, h * w, ch).mean(dim=1)
        weights = weights.clamp(min=1e-5).view(b, ch, 1, 1)
        dec = F.normalize(dec.reshape(b, ch, h * w), dim=-1).reshape(b, ch, h, w) * weights
    mapper, seg = _tail_modules(self)
    return hyper_head_tail(dec, mapper, seg, size=size, resize_embed=False)


class ASPP_Classifier_V2_Hyper(nn.Module):
    """Drop-in for core/models/classifier.py:335-379 (DeepLab-v2 hyperbolic head): same constructor, same
    parameter names (`conv2d_list.N.weight/bias`, `conv_seg.P_MLR/A_MLR`), same forward interface."""

    def __init__(self, in_channels, dilation_series, padding_series, num_classes, reduced_channels):
        super().__init__()
        self.conv2d_list = nn.ModuleList(
            nn.Conv2d(in_channels, reduced_channels, kernel_size=3, stride=1, padding=p, dilation=d, bias=True)
            for d, p in zip(dilation_series, padding_series))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)
        self.mapper = HyperMapper(c=cfg.MODEL.CURVATURE)
        self.conv_seg = HyperMLR(reduced_channels, num_classes, c=cfg.MODEL.CURVATURE)

    forward = v2_hyper_forward


def patch_reference_heads(module):
    """Bind the HIP-tail forwards onto the reference's head classes found in `module`
    (core.models.classifier).  Returns the names patched."""
    done = []
    for name, fwd in (("ASPP_Classifier_V2_Hyper", v2_hyper_forward), ("DepthwiseSeparableASPP_Hyper", v3plus_hyper_forward)):
        cls = getattr(module, name, None)
        if isinstance(cls, type) and cls.__dict__.get("forward") is not fwd:
            cls._reference_forward = cls.__dict__.get("forward")
            cls.forward = fwd
            done.append(name)
    return done
