"""Probe: where in HBM does the scoring pass run fast?  For placeholders of 0 / 32 / 64 / 96 GiB allocated first, a 128 GiB
float64 pool is allocated (contiguous), filled, and timed per 16 GiB window with halo_hbm_walk_probe and per 64 GiB batch with the
real scoring call; everything is freed before the next placement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from tools.halo_probe import alloc_contiguous, probe_streaming, contiguous_memory_stats
from halo_amd.core.active.floating_region import score_maps

dev = torch.device("cuda:0")
B, C, O, H, W = 16, 256, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(1)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
del low
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


shifts = [int(v) for v in sys.argv[1:]] or [0, 32, 64, 96]
for shift in shifts:
    hold = alloc_contiguous((shift << 30,), torch.uint8, dev) if shift else None
    try:
        t = alloc_contiguous((2 * B, C, H, W), torch.float64, dev)
    except torch.cuda.OutOfMemoryError:
        print(f"placeholder {shift} GiB: pool does not fit"); del hold; continue
    rows0 = probe_streaming(t, C, H * W * 8)
    for b in range(2 * B):
        t[b].copy_(src)
    rows = probe_streaming(t, C, H * W * 8)
    print(f"placeholder {shift:3d} GiB, pool at {t.data_ptr():#x}: walk per 16 GiB window, uninitialised {[round(r[2]) for r in rows0]}")
    print(f"                                                   filled        {[round(r[2]) for r in rows]}   flat {[round(r[3]) for r in rows]}")
    print(f"                                                   scoring call alone: batch 0 {timeit(t[:B]):.3f} ms, batch 1 {timeit(t[B:]):.3f} ms", flush=True)
    del t, hold, rows, rows0
print(contiguous_memory_stats())
