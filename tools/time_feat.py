"""Timing aid: the scoring pass alone (no selection beside it) at the headline shape -- 16 x (256, 1024, 2048) embeddings
in float64 and float32 plus 19-class logits -- so that k_feat_reduce variants can be compared without the pipeline's
other streams.  Prints ms per 16-image call and the fraction of the 8 TB/s HBM spec from the algorithmic bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, int(os.environ.get("C", 256)), 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
del low
for dt in (torch.float64, torch.float32):
    feat = torch.empty((B, C, H, W), device=dev, dtype=dt)
    for b in range(B):
        feat[b] = torch.randn((C, H, W), generator=g, device=dev, dtype=torch.float32) * 0.05
    for unc, pur in (("entropy", "radius"), ("zeros", "radius")):
        def run():
            score_maps(logit, feat, unc, pur, True, None, size=3, want_maps=True)
        try:
            run()
        except Exception as e:          # 'zeros' may not be a public uncertainty name everywhere
            print(dt, unc, pur, "skipped:", str(e)[:80]); continue
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        n = 6
        for _ in range(n):
            run()
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        es = 8 if dt == torch.float64 else 4
        by = B * H * W * (C * es + (O * 4 if unc == "entropy" else 0) + es + 4)
        print(f"{str(dt):14s} {unc:8s} {pur:7s} {ms:7.3f} ms / {B} images   {by / ms / 1e6:6.0f} GB/s  frac {by / ms / 1e6 / 8000:.3f}", flush=True)
    del feat
