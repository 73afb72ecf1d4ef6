"""Timing aid: greedy selection alone vs beside the streaming feature kernel (one MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo_amd.core.active.build import greedy_select
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
H, W, n = 1024, 2048, 2331
for B in (1, 4, 16, 32):
    g = torch.Generator(device=dev).manual_seed(3)
    base = torch.randn((B, H // 4, W // 4), generator=g, device=dev, dtype=torch.float64)
    score0 = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=True)[0].contiguous()
    gt = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
    def run():
        sc = score0.clone()
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        greedy_select(sc, n, 1, 5, act, sel, am, gt, return_picks=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run()
    alone = min(run() for _ in range(3))
    # beside a streaming kernel
    feat = torch.randn((4, 256, H, W), device=dev, dtype=torch.float64) * 0.01
    logit = torch.randn((4, 19, H, W), device=dev)
    s2 = torch.cuda.Stream(dev)
    def run_loaded():
        sc = score0.clone()
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            for _ in range(12):
                score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
        t0 = time.perf_counter()
        greedy_select(sc, n, 1, 5, act, sel, am, gt, return_picks=False)
        torch.cuda.current_stream().synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        return dt
    run_loaded()
    loaded = min(run_loaded() for _ in range(3))
    print(f"B={B:2d}: alone {alone:7.2f} ms ({alone / n * 1e3:5.2f} us/step)   beside streaming {loaded:7.2f} ms ({loaded / n * 1e3:5.2f} us/step)")
    del feat, logit
