"""Probe: do the low-res logit pass (float32 VALU) and the exact embedding pass (float64 VALU + LDS) overlap when they run on two
streams?  Sequential on one stream vs concurrent on two, 16 images, x4 geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import halo_amd; halo_amd.configure(hw_queues=int(os.environ.get("Q", "2")))
from halo_amd.core.active.floating_region import score_maps_lowres
from halo_amd.core.utils.hyperbolic import HyperMapper

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
lg = torch.randn((B, O, 256, 512), generator=g, device=dev)
em = HyperMapper(1.0).expmap(torch.randn((B, C, 256, 512), generator=g, device=dev) * 0.1, dim=1)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def logit_only():
    return score_maps_lowres(lg, None, (H, W), "entropy", "none", True, None, want_maps=False, mode="exact")
def feat_only():
    return score_maps_lowres(lg, em, (H, W), "none", "radius", True, None, want_maps=False, mode="exact")
def both():
    return score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, want_maps=False, mode="exact")

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def concurrent():
    with torch.cuda.stream(s1):
        a = logit_only()
    with torch.cuda.stream(s2):
        b = feat_only()
    return a, b

for rep in range(2):
    print("logit pass + tail alone      : %.3f ms" % timeit(logit_only))
    print("embedding pass + tail alone  : %.3f ms" % timeit(feat_only))
    print("one call (sequential passes) : %.3f ms" % timeit(both))
    print("two streams, concurrent      : %.3f ms" % timeit(concurrent))
