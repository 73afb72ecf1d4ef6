"""Probe: does giving a large allocation back to the driver slow down kernels that run in the seconds after it?  Two 68.7 GB
tensors; the scoring call is timed on the first one before and for ~8 s after the second one is freed (torch.cuda.empty_cache)."""
import os, sys, time
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
a = torch.empty((B, C, H, W), dtype=torch.float64, device=dev)
b = torch.empty((B, C, H, W), dtype=torch.float64, device=dev)
for i in range(B):
    a[i].copy_(src); b[i].copy_(src)


def call():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    score_maps(logit, a, "entropy", "radius", True, None, size=3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    call()
time.sleep(6.0)                                  # let whatever the set-up left behind settle
print("before the free:", " ".join(f"{call():.2f}" for _ in range(8)))
del b
torch.cuda.synchronize(); torch.cuda.empty_cache()
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < 9.0:
    ms = call()
    out.append((time.perf_counter() - t0, ms))
    time.sleep(0.15)
print("after freeing 68.7 GB (seconds since the free : ms):")
print("  " + "  ".join(f"{t:.1f}:{ms:.2f}" for t, ms in out))
