"""Probe: does the feature-stream bandwidth depend on WHICH allocation holds the pool (one MI355X)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
logit = torch.randn((B, O, H, W), device=dev)
keep = []
for trial in range(8):
    feat = torch.empty((B, C, H, W), dtype=torch.float64, device=dev)
    for b in range(B):
        feat[b].normal_(0, 0.01)
    def run():
        score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"allocation {trial}: ptr {feat.data_ptr():#x}  score_maps {ms:.3f} ms  ({B * H * W * (C * 8) / ms / 1e6:.0f} GB/s of features)")
    if trial % 2 == 0:
        keep.append(feat)          # hold some allocations so the next one lands elsewhere
    else:
        del feat
        keep.clear()
        torch.cuda.empty_cache()
