"""Probe: does re-allocating a batch tensor inside one process change the scoring call's speed on it?  Six 68.7 GB candidates, one
after the other; the previous candidate is still held while the next is allocated (so the next one gets other memory), then freed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd  # noqa: F401
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
src = (torch.randn((C, H, W), generator=g, device=dev, dtype=torch.float32) * 0.05).double()


def timeit(f):
    score_maps(logit, f, "entropy", "radius", True, None, size=3)
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        score_maps(logit, f, "entropy", "radius", True, None, size=3)
    b_.record(); torch.cuda.synchronize()
    return a.elapsed_time(b_) / 4


prev = None
for k in range(6):
    t = torch.empty((B, C, H, W), dtype=torch.float64, device=dev)
    for b in range(B):
        t[b].copy_(src)
    ms = timeit(t)
    print(f"candidate {k} at {t.data_ptr():#x} (previous one {'held' if prev is not None else 'none'}): scoring call {ms:.3f} ms", flush=True)
    del prev
    torch.cuda.empty_cache()
    prev = t
