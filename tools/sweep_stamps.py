"""Diagnostic: phase cycle sums of the value-binned sweep (wave 0) from a -DHALO_SWEEP_STAMPS build of the library.

    python tools/sweep_stamps.py --build      # build container: compiles tools/_stamps/libhalo_hip_stamps.so
    python tools/sweep_stamps.py              # GPU box: runs one selection per case and prints the sums

The stamped build writes into the selector header's padding words only (no output value depends on them)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_stamps", "libhalo_hip_stamps.so")
if "--build" in sys.argv:
    from halo_amd import _build
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = []
    for src in _build.SOURCES:
        obj = os.path.join(os.path.dirname(SO), src.replace(".hip", ".o"))
        subprocess.run([_build._hipcc()] + _build.FLAGS + ["-DHALO_SWEEP_STAMPS", "-c", os.path.join(_build.CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs, check=True)
    print("built", SO)
    sys.exit(0)
os.environ["HALO_LIB_PATH"] = SO
import numpy as np, torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active import build as B
from halo_amd.core.active.floating_region import _workspace, score_maps

dev = torch.device("cuda:0")
H, W, n = 1024, 2048, 2331
g = torch.Generator(device=dev).manual_seed(3)
base = torch.randn((1, H // 4, W // 4), generator=g, device=dev, dtype=torch.float64)
score0 = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=True)[0].contiguous()
gt = torch.zeros((1, H, W), dtype=torch.int64, device=dev)
feat = torch.randn((4, 256, H, W), device=dev, dtype=torch.float64) * 0.01
logit = torch.randn((4, 19, H, W), device=dev)
s2 = torch.cuda.Stream(dev)
for loaded in (False, True):
    for rep in range(2):
        sc = score0.clone()
        act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if loaded:
            with torch.cuda.stream(s2):
                for _ in range(12):
                    score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
        B.greedy_select(sc, n, 1, 5, act, sel, am, gt)
        torch.cuda.current_stream().synchronize()
        key = [k for k in B._workspace.__globals__["_WS"] if k[2] == "select"][0]
        ws = B._workspace.__globals__["_WS"][key]
        base_off = (-ws.data_ptr()) % 256
        hdr = ws[base_off:base_off + 64].cpu().numpy().view(np.uint32)
        f, a, r, b = (int(v) * 16 for v in hdr[12:16])
        nb = int(hdr[4])
        tot = f + a + r + b
        torch.cuda.synchronize()
        print(f"survivors total {int(hdr[6])} (avg {int(hdr[6]) / max(nb, 1):.1f} per bin, max {int(hdr[5])})")
        print(f"{'beside streaming' if loaded else 'alone':17s} bins {nb:5d}  cycles/bin: filter {f / nb:7.0f}  barrier A {a / nb:6.0f}  resolve {r / nb:7.0f}  "
              f"barrier B {b / nb:6.0f}   total {tot / 1e6:.2f} M shader cycles (s_memtime)  np {int(hdr[11])}")
