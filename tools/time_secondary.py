"""Timing aid: the secondary kernels of the head tail at the shapes round 1 profiled (one MI355X):
expmap0+project (float32 planes -> float64), bilinear align_corners resize, HyperMLR.  Prints ms and the
achieved fraction of the 8 TB/s HBM spec from algorithmic bytes; set HALO_EXPMAP_PLANES=1 / HALO_BILINEAR_FLAT=1
for the previous kernels (A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners

dev = torch.device("cuda:0")


def t(fn, n=10):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m = HyperMapper(1.0)
for (C, h, w) in ((256, 256, 512), (64, 160, 320), (64, 640, 1280), (512, 256, 512)):
    z = torch.randn((1, C, h, w), device=dev) * 0.1
    for env in ({}, {"HALO_EXPMAP_PLANES": "1"}):
        os.environ.update(env)
        ms = t(lambda: m.expmap(z, dim=1))
        for k in env:
            os.environ.pop(k)
        by = z.numel() * (4 + 8)
        print(f"expmap f32->f64 C={C} {h}x{w} {'two-pass' if env else 'LDS tile'}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s  frac {by / ms / 1e6 / 8000:.2f}", flush=True)
    dst = torch.empty((1, C, h, w), device=dev, dtype=torch.float64)
    ms_copy = t(lambda: dst.copy_(z))
    print(f"   torch copy_ f32 -> f64 of the same tensor (read 4 B, write 8 B per element, no arithmetic): {ms_copy:.3f} ms  {z.numel() * 12 / ms_copy / 1e6:.0f} GB/s", flush=True)
    del dst
for (dt, planes, hw_in, hw_out) in ((torch.float64, 256, (256, 512), (1024, 2048)), (torch.float32, 19, (256, 512), (1024, 2048)),
                                    (torch.float32, 19, (640, 1280), (1024, 2048)), (torch.float64, 64, (160, 320), (1024, 2048))):
    src = torch.randn((1, planes) + hw_in, device=dev, dtype=dt)
    for env in ({}, {"HALO_BILINEAR_FLAT": "1"}):
        os.environ.update(env)
        ms = t(lambda: bilinear_align_corners(src, hw_out))
        for k in env:
            os.environ.pop(k)
        by = (src.numel() + planes * hw_out[0] * hw_out[1]) * src.element_size()
        print(f"bilinear {str(dt)[6:]} {planes}x{hw_in}->{hw_out} {'flat' if env else 'rows'}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s  frac {by / ms / 1e6 / 8000:.2f}", flush=True)
    # what a plain store stream of the same output gets on this box (the resize writes 16-40x what it reads)
    dst = torch.empty((1, planes) + hw_out, device=dev, dtype=dt)
    ms_fill = t(lambda: dst.fill_(1.0))
    print(f"   flat fill of the same {dst.numel() * dst.element_size() / 1e6:.0f} MB output: {ms_fill:.3f} ms  {dst.numel() * dst.element_size() / ms_fill / 1e6:.0f} GB/s", flush=True)
    del dst
for (C, h, w) in ((256, 1024, 2048), (64, 160, 320)):
    x = m.expmap(torch.randn((1, C, h, w), device=dev) * 0.1, dim=1)
    mlr = HyperMLR(C, 19).to(dev)
    ms = t(lambda: mlr._hyper_logits(x, out_dtype=torch.float32), 5)
    flop = 2.0 * 2 * 19 * C * h * w
    print(f"hypermlr C={C} {h}x{w}: {ms:.3f} ms  {flop / ms / 1e9:.1f} useful TFLOP/s  {x.numel() * 8 / ms / 1e6:.0f} GB/s", flush=True)
