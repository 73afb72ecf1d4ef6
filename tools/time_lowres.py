"""Timing aid: fused low-res scoring vs explicit upsample + full-res scoring (one MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
from halo_amd.core.utils.hyperbolic import HyperMapper, bilinear_align_corners

dev = torch.device("cuda:0")
B, C, O, H, W = 4, 256, 19, 1024, 2048
for (hl, wl, hf, wf, tag) in ((256, 512, 256, 512, "x4 synthetic"), (640, 1280, 160, 320, "real pipeline ratios")):
    g = torch.Generator(device=dev).manual_seed(1)
    lg = torch.randn((B, O, hl, wl), generator=g, device=dev)
    em = HyperMapper(1.0).expmap(torch.randn((B, C, hf, wf), generator=g, device=dev) * 0.1, dim=1)

    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    fused = t(lambda: score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, want_maps=False, mode="exact"))
    gram = t(lambda: score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, want_maps=False, mode="gram"))
    def explicit():
        a = bilinear_align_corners(lg, (H, W)); b = bilinear_align_corners(em, (H, W))
        return score_maps(a, b, "entropy", "radius", True, None, want_maps=False)
    expl = t(explicit)
    print(f"{tag}: fused {fused / B:.3f} ms/image (Gram mode {gram / B:.3f}), upsample+score {expl / B:.3f} ms/image, "
          f"ratio {expl / fused:.1f}x ({expl / gram:.1f}x)")
