"""A/B aid: k_feat_reduce's workgroup -> chunk map (HALO_FEAT_XCD_GRANULE, read per call; -1 = plain), INTERLEAVED on one
allocation (the plateau a run lands on is a property of the allocation, NOTES.md), with nothing else on the GPU.
    python tools/ab_feat_map.py [granule ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from tools import halo_probe
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
configs = [-1, 256, 32, 1024]
if len(sys.argv) > 1:
    configs = [int(a) for a in sys.argv[1:]]
for dt in (torch.float64, torch.float32):
    feat = torch.empty((B, C, H, W), device=dev, dtype=dt)
    for b in range(B):
        feat[b] = torch.randn((C, H, W), generator=g, device=dev, dtype=torch.float32) * 0.05
    nb = feat.numel() * feat.element_size()
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    for i in range(5):
        halo_probe.read_probe(feat, nb, sink); ev[i].record()
    torch.cuda.synchronize()
    flat = np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(1, 4)])
    print(f"{dt}: flat read of this allocation {flat:.3f} ms = {nb / flat / 1e6:.0f} GB/s")
    ref = None
    res = {c: [] for c in configs}
    for rep in range(5):
        for cfg in configs:
            os.environ["HALO_FEAT_XCD_GRANULE"] = str(cfg)
            out = score_maps(logit, feat, "entropy", "radius", True, None, size=3, want_maps=True)
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                score_maps(logit, feat, "entropy", "radius", True, None, size=3, want_maps=True)
            b_.record(); torch.cuda.synchronize()
            res[cfg].append(a.elapsed_time(b_) / 3)
            if ref is None:
                ref = [o.clone() for o in out[:3]]
            else:
                assert all(torch.equal(x, y) for x, y in zip(ref, out[:3])), cfg
    es = feat.element_size()
    by = B * H * W * (C * es + O * 4 + es + 4)
    for cfg in configs:
        ms = np.array(res[cfg])
        print(f"  granule {cfg:5d}: {ms.mean():7.3f} ms (min {ms.min():.3f} max {ms.max():.3f})  {by / ms.mean() / 1e6:6.0f} GB/s  frac {by / ms.mean() / 8e9:.4f}   (whole scoring call)")
    del feat
