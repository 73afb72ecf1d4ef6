"""Timing aid: HyperMLR on the f64 matrix cores vs the VALU kernel (one MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR

dev = torch.device("cuda:0")
for (C, h, w, tag) in ((64, 160, 320, "real head, C=64 160x320"), (256, 256, 512, "bench ring, C=256 256x512"), (256, 1024, 2048, "full res, C=256 1024x2048")):
    x = HyperMapper(1.0).expmap(torch.randn((1, C, h, w), device=dev) * 0.1, dim=1)
    mlr = HyperMLR(C, 19).to(dev)
    def t(n=5):
        with torch.no_grad():
            mlr._hyper_logits(x, out_dtype=torch.float32); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                mlr._hyper_logits(x, out_dtype=torch.float32)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    os.environ.pop("HALO_MLR_VALU", None)
    m = t()
    os.environ["HALO_MLR_VALU"] = "1"
    v = t()
    os.environ.pop("HALO_MLR_VALU", None)
    flop = 2.0 * 2 * 19 * C * h * w
    print(f"{tag}: mfma {m:.3f} ms ({flop / m / 1e9:.1f} useful TFLOP/s, {x.numel() * 8 / m / 1e6:.0f} GB/s)   valu {v:.3f} ms ({flop / v / 1e9:.1f} TFLOP/s)")
