"""Timing aid: greedy selection (value-binned sweep vs the serial tile-table kernel) alone and beside the
streaming feature kernel (one MI355X).  Prints per-call wall time, per-pick time and, with --kernels,
HIP-event times of the sweep's individual launches (they are issued back to back on one stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.build import greedy_select
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
H, W, n = 1024, 2048, 2331
mrad = int(os.environ.get("MRAD", "5"))
methods = os.environ.get("METHODS", "auto,serial").split(",")
RANGED = os.environ.get("RANGED", "0") == "1"      # the pipeline's case: the scorer supplied the maps' value range (prepared untimed here)
from halo_amd import _lib
from halo_amd.core.active.floating_region import new_score_range
for B in (1, 4, 16, 32):
    g = torch.Generator(device=dev).manual_seed(3)
    base = torch.randn((B, H // 4, W // 4), generator=g, device=dev, dtype=torch.float64)
    score0 = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=True)[0].contiguous()
    gt = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
    feat = torch.randn((4, 256, H, W), device=dev, dtype=torch.float64) * 0.01
    logit = torch.randn((4, 19, H, W), device=dev)
    s2 = torch.cuda.Stream(dev)
    # the selection runs where the pipelines run it (bench.py, RegionSelection): on a HIGH-priority stream.  On the default
    # stream it can share a hardware queue with the streaming stream (GPU_MAX_HW_QUEUES=2, halo_amd/__init__.py) and then waits
    # for every 131 072-workgroup feature launch ahead of it in that queue to be dispatched (34 ms for B = 16 in one run).
    hi = torch.cuda.Stream(dev, priority=-1)
    ref = None
    for method in methods:
        def run(loaded):
            sc = score0.clone()
            act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
            am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
            rec = None
            if RANGED and method != "serial":
                rec = new_score_range(B, dev)
                _lib.check(_lib.lib().halo_score_range(_lib.ptr(sc), _lib.dtype_code(sc), B, H, W, _lib.ptr(rec), _lib.stream_ptr(dev)), "halo_score_range")
            torch.cuda.synchronize()
            if loaded:
                with torch.cuda.stream(s2):
                    for _ in range(12):
                        score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
            t0 = time.perf_counter()
            with torch.cuda.stream(hi):
                picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, gt, method=method, score_range=rec)
            hi.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            return dt, picks, npk
        run(False)
        alone = min(run(False)[0] for _ in range(3))
        run(True)
        res = [run(True) for _ in range(3)]
        loaded = min(r[0] for r in res)
        picks, npk = res[-1][1], res[-1][2]
        if ref is None:
            ref = picks
        same = bool(torch.equal(ref, picks))
        print(f"B={B:2d} {method:7s}: alone {alone:7.2f} ms ({alone / n * 1e3:5.2f} us/pick)   beside streaming {loaded:7.2f} ms "
              f"({loaded / n * 1e3:5.2f} us/pick)   picks {int(npk.min())}..{int(npk.max())}  same as first method: {same}", flush=True)
    del feat, logit
