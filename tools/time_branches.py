"""Timing aid: every (uncertainty, purity) branch of the scorer at 1024x2048, B=8 (one MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 8, 64, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
feat = (torch.randn((B, C, H, W), generator=g, device=dev, dtype=torch.float64) * 0.05)
gt = torch.randint(0, O, (B, H, W), generator=g, device=dev)
for unc, pur, norm in (("entropy", "radius", True), ("entropy", "ripu", False), ("entropy", "hyper", True),
                       ("oracle_acc", "oracle_ripu", False), ("pixel_entropy", "euc_norm", True), ("entropy", "none", False)):
    def run():
        score_maps(logit, feat, unc, pur, norm, gt, size=3, K=100, want_maps=False)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{unc:14s} {pur:12s} {ms / B:7.3f} ms/image")
