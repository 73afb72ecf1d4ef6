"""A/B aid: the matrix-core HyperMLR with its one-quotient epilogue (default) against the reference-order epilogue
(HALO_MLR_EPI_REF=1, the round-4 statement) -- same launches, graph-replay timing, and the largest difference of the
logits between the two and against the oracle.  ASSERTS the agreement (a disagreement fails the collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
from tools.time_head import graph_ms          # noqa: E402  (prints nothing on import: guarded below)

dev = torch.device("cuda:0")
m = HyperMapper(1.0)


def both(fn):
    os.environ.pop("HALO_MLR_EPI_REF", None)
    new = fn()
    os.environ["HALO_MLR_EPI_REF"] = "1"
    try:
        ref = fn()
    finally:
        os.environ.pop("HALO_MLR_EPI_REF", None)
    return new, ref


for (tag, C, h, w, scale) in (("v3+ head", 64, 160, 320, 0.1), ("v2 head", 64, 640, 1280, 0.1), ("bench pool", 256, 256, 512, 0.1),
                              ("full size", 256, 1024, 2048, 0.1), ("v2 head, x at the ball's boundary", 64, 640, 1280, 0.6)):
    g = torch.Generator(device=dev).manual_seed(3)
    x = m.expmap(torch.randn((1, C, h, w), device=dev, generator=g) * scale, dim=1)
    mlr = HyperMLR(C, 19).to(dev)
    with torch.no_grad():
        a, b = both(lambda: mlr._hyper_logits(x, out_dtype=torch.float64))
        ta, tb = both(lambda: graph_ms(lambda: mlr._hyper_logits(x, out_dtype=torch.float32)))
    d = float((a - b).abs().max())
    line = f"{tag:36s} C={C:3d} {h}x{w}: one-quotient {ta * 1e3:8.1f} us   reference-order {tb * 1e3:8.1f} us   max |diff| {d:.2e}"
    if h * w <= 160 * 320:
        from oracle import halo_oracle as ho
        want = ho.hypermlr(x.cpu().numpy(), mlr.P_MLR.detach().cpu().numpy(), mlr.A_MLR.detach().cpu().numpy(), 1.0)
        eo = float(np.abs(a.cpu().numpy() - want).max())
        line += f"   vs oracle {eo:.2e}"
        assert eo < 1e-11, line
    print(line, flush=True)
    assert d < 1e-11 and bool(torch.isfinite(a).all()), line
