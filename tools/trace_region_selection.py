"""Timing aid: where one image's time goes in RegionSelection's host code (launch, event wait, staging copies, files)."""
import os, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd      # hardware queues: ROCm's default (tools/time_region_selection.py says why)
from halo_amd.core.active import build as B
from halo_amd.core.utils.hyperbolic import HyperMapper

dev = torch.device("cuda:0")
H, W, C, O = 1024, 2048, 64, 19
cfg = types.SimpleNamespace(
    MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
    ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                 BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
g = torch.Generator(device=dev).manual_seed(0)
emb = HyperMapper(1.0).expmap(torch.randn((1, C, 160, 320), generator=g, device=dev) * 0.1, dim=1)
logit = torch.nn.functional.interpolate(torch.randn((1, O, 160, 320), generator=g, device=dev), size=(640, 1280), mode="bilinear", align_corners=True)
prm = B.AcquisitionParams(cfg)
tmp = tempfile.mkdtemp()
gt = torch.randint(0, O, (1, H, W)).pin_memory()
om = torch.full((1, H, W), 255, dtype=torch.int64).pin_memory()
ac = torch.zeros(1, H, W, dtype=torch.bool).pin_memory()
se = torch.zeros(1, H, W, dtype=torch.bool).pin_memory()
slot = B._Slot(torch.cuda.Stream(dev, priority=-1))
for it in range(30):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    rec = B._launch(prm, logit, emb, (H, W), om, gt, ac, se, dev, slot); t.append(time.perf_counter())
    rec.done.synchronize(); t.append(time.perf_counter())
    m = rec.buf.out_mask[0].numpy().copy()
    a = torch.from_numpy(rec.buf.out_active[0].numpy().copy()); s = torch.from_numpy(rec.buf.out_selected[0].numpy().copy()); t.append(time.perf_counter())
    B.write_png_gray8(os.path.join(tmp, "m.png"), m); t.append(time.perf_counter())
    torch.save({"active": a, "selected": s}, os.path.join(tmp, "i.pth")); t.append(time.perf_counter())
    n = int(rec.npk[0]); t.append(time.perf_counter())
    names = ["launch (host)", "event wait", "copy out of pinned", "png", "torch.save", "npk.item"]
    print("  ".join("%s %.2f" % (nm, (t[i + 1] - t[i]) * 1e3) for i, nm in enumerate(names)), " total %.2f ms" % ((t[-1] - t[0]) * 1e3), flush=True)
