"""Diagnostic (run under rocprofv3 --pmc ... --kernel-trace): the scoring call on each of three resident 16-image batches in turn,
eight rounds, so that the counters of k_feat_reduce launches on a fast and on a slow batch can be compared inside one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
nb = int(os.environ.get("BATCHES", "3"))
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
feat = torch.empty((nb * B, C, H, W), device=dev, dtype=torch.float64)
src = (torch.randn((C, H, W), generator=g, device=dev, dtype=torch.float32) * 0.05).double()
for b in range(nb * B):
    feat[b].copy_(src)
del src
torch.cuda.synchronize()
for rnd in range(8):
    for i in range(nb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        score_maps(logit, feat[i * B:(i + 1) * B], "entropy", "radius", True, None, size=3)
        e1.record(); torch.cuda.synchronize()
        if rnd >= 6:
            print(f"round {rnd} batch {i}: scoring call {e0.elapsed_time(e1):.3f} ms", flush=True)
