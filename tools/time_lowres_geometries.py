"""Timing aid: the fused low-res scorer at the real head geometries -- DeepLab-v3+ (logits 640x1280, embedding 160x320),
DeepLab-v2 (both 640x1280, classifier.py:375-377) -- and the x4 synthetic one, against explicit upsample + score."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
from halo_amd.core.utils.hyperbolic import HyperMapper, bilinear_align_corners
dev = torch.device("cuda:0")
H, W, O = 1024, 2048, 19
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for tag, C, (hf, wf), (hl, wl) in (("v3+ real", 64, (160, 320), (640, 1280)), ("v2 real", 64, (640, 1280), (640, 1280)), ("x4 C=256", 256, (256, 512), (256, 512))):
    emb = HyperMapper(1.0).expmap(torch.randn((1, C, hf, wf), device=dev) * 0.1, dim=1)
    lg = torch.randn((1, O, hl, wl), device=dev)
    a = t(lambda: score_maps_lowres(lg, emb, (H, W), "entropy", "radius", True, None, ksize=3, want_maps=False))
    b = t(lambda: score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(emb, (H, W)), "entropy", "radius", True, None, size=3, want_maps=False))
    print(f"{tag}: fused low-res {a:.3f} ms/image   explicit upsample + score {b:.3f} ms/image")
