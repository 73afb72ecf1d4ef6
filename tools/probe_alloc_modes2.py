"""Probe: one allocation, many timed launches -- is the plateau a property of the allocation or of time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo_amd.core.active.floating_region import score_maps
dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
logit = torch.randn((B, O, H, W), device=dev)
for alloc in range(3):
    feat = torch.empty((B, C, H, W), dtype=torch.float64, device=dev)
    for b in range(B):
        feat[b].normal_(0, 0.01)
    ts = []
    for it in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if it == 11:
            time.sleep(2.0)
    print(f"alloc {alloc} ptr {feat.data_ptr():#x}: " + " ".join(f"{t:.1f}" for t in ts))
    del feat; torch.cuda.empty_cache()
