"""Hyperbolic head tails -- the part of core/models/classifier.py that is on the hot path.

ASPP_Classifier_V2_Hyper.forward (classifier.py:364-379) and DepthwiseSeparableASPP_Hyper.forward
(classifier.py:552-558) both end with

    embed = mapper.expmap(feat, dim=1)                 # float64
    out   = conv_seg(embed.double()).float()           # HyperMLR
    [bilinear(align_corners=True) of out (v3+) or of out AND embed (v2)]
    return out, embed

The convolutional bodies in front of it stay on PyTorch-ROCm/MIOpen (out of scope);
`hyper_head_tail` is the inference-time replacement for those last lines.
"""
import torch

from ..utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners


def hyper_head_tail(feat, mapper: HyperMapper, conv_seg: HyperMLR, size=None, resize_embed=False):
    """feat (B,C,h,w) float32 from conv_reduce / the ASPP sum -> (out float32, embed float64).

    resize_embed=False: DeepLab-v3+ tail (classifier.py:552-558); True: DeepLab-v2 tail, which also
    resizes the embedding (classifier.py:375-377)."""
    training = torch.is_grad_enabled() and (feat.requires_grad or conv_seg.P_MLR.requires_grad)
    if training:
        # expmap and HyperMLR carry HIP backward kernels; the resize stays on F.interpolate, which autograd
        # already differentiates (it is outside the kernels' scope under training)
        embed = mapper.expmap(feat, dim=1)
        out = conv_seg(embed).float()
        if size is not None:
            out = torch.nn.functional.interpolate(out, size=size, mode="bilinear", align_corners=True)
            if resize_embed:
                embed = torch.nn.functional.interpolate(embed, size=size, mode="bilinear", align_corners=True)
        return out, embed
    with torch.no_grad():
        embed = mapper.expmap(feat, dim=1)
        out = conv_seg._hyper_logits(embed, out_dtype=torch.float32)
        if size is not None:
            out = bilinear_align_corners(out, size)
            if resize_embed:
                embed = bilinear_align_corners(embed, size)
    return out, embed
