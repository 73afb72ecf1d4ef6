import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.utils.hyperbolic import HyperMapper
dev = torch.device("cuda:0"); m = HyperMapper(1.0)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (C, h, w) in ((256, 256, 512), (64, 160, 320), (64, 640, 1280), (512, 256, 512)):
    z = torch.randn((1, C, h, w), device=dev) * 0.1
    ref = None
    for kb in ("8", "16", "32", "48", "64", "80"):
        os.environ["HALO_EXPMAP_TILE_KB"] = kb
        out = m.expmap(z, dim=1); torch.cuda.synchronize()
        if ref is None: ref = out
        ms = t(lambda: m.expmap(z, dim=1))
        print(f"C={C} {h}x{w} tile {kb:>2} KiB: {ms:.3f} ms  {z.numel() * 12 / ms / 1e6:.0f} GB/s  same bits {torch.equal(out, ref)}", flush=True)
