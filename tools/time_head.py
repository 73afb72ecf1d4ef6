"""Timing aid: the head-tail kernels at the REAL head shapes, free of host launch overhead.

tools/time_secondary.py times `n` eager calls between two host synchronisations; for kernels of 10-50 us that measures the Python /
ctypes / hipLaunch rate of the calling thread as much as the GPU.  Here every call sequence is captured ONCE into a HIP graph
(20 repetitions), the graph is replayed, and HIP events around the replay give the GPU's own time per call; the eager number is
printed beside it.  Shapes: the v3+ head (feat 64x160x320 -> logits 19x160x320 -> 640x1280), the v2 head (feat 64x640x1280 ->
1024x2048) and bench.py's synthetic pool (256x256x512)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners

dev = torch.device("cuda:0")
REP = 20


def eager_ms(fn, n=50):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graph_ms(fn):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(REP):
                    fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / REP)
    return best


def line(tag, nbytes, fn):
    e, g = eager_ms(fn), graph_ms(fn)
    print(f"{tag:78s} eager {e * 1e3:7.1f} us   graph {g * 1e3:7.1f} us   {nbytes / g / 1e6:6.0f} GB/s   frac {nbytes / g / 1e6 / 8000:.2f}", flush=True)


def main():
    m = HyperMapper(1.0)
    O = 19
    for (tag, C, h, w, up) in (("v3+ head", 64, 160, 320, (640, 1280)), ("v2 head", 64, 640, 1280, (1024, 2048)), ("bench pool", 256, 256, 512, (1024, 2048))):
        z = torch.randn((1, C, h, w), device=dev) * 0.1
        mlr = HyperMLR(C, O).to(dev)
        x = m.expmap(z, dim=1)
        lg = mlr._hyper_logits(x, out_dtype=torch.float32)
        n = C * h * w
        line(f"{tag}: expmap f32->f64 C={C} {h}x{w}", n * 12, lambda: m.expmap(z, dim=1))
        line(f"{tag}: hypermlr (prep + contraction + epilogue) -> f32 logits", n * 8 + O * h * w * 4, lambda: mlr._hyper_logits(x, out_dtype=torch.float32))
        line(f"{tag}: bilinear f32 {O}x{h}x{w} -> {up}", (O * h * w + O * up[0] * up[1]) * 4, lambda: bilinear_align_corners(lg, up))
        if tag == "v2 head":
            line(f"{tag}: bilinear f64 {C}x{h}x{w} -> {up} (the embedding, classifier.py:375-377)", (n + C * up[0] * up[1]) * 8, lambda: bilinear_align_corners(x, up))

        def tail():
            e = m.expmap(z, dim=1)
            o = mlr._hyper_logits(e, out_dtype=torch.float32)
            return bilinear_align_corners(o, up), e
        line(f"{tag}: whole tail, two kernels (expmap -> HyperMLR -> .float() -> resize of the logits)", n * 12 + n * 8 + O * h * w * 8 + O * up[0] * up[1] * 4, tail)
        if C == 64:          # the heads' own channel count: halo_head_tail, one kernel for expmap -> project -> HyperMLR -> .float()
            from halo_amd.core.utils.hyperbolic import head_tail_fused
            line(f"{tag}: FUSED expmap + HyperMLR (halo_head_tail: embedding written, never re-read)", n * 12 + O * h * w * 4,
                 lambda: head_tail_fused(z, mlr.P_MLR, mlr.A_MLR, 1.0))

            def tail_fused():
                o, e = head_tail_fused(z, mlr.P_MLR, mlr.A_MLR, 1.0)
                return bilinear_align_corners(o, up), e
            line(f"{tag}: whole tail, FUSED (what the head's forward runs under no_grad)", n * 12 + O * h * w * 8 + O * up[0] * up[1] * 4, tail_fused)


if __name__ == "__main__":
    main()
