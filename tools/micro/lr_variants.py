import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
from halo_amd.core.utils.hyperbolic import HyperMapper, bilinear_align_corners
dev = torch.device("cuda:0")
O = 19
for (B, C, hf, wf, H, W) in ((2, 40, 160, 320, 1024, 2048), (2, 64, 160, 320, 1024, 2048), (2, 16, 160, 320, 1024, 2048), (2, 256, 256, 512, 1024, 2048), (2, 9, 256, 512, 1024, 2048), (2, 18, 256, 512, 1024, 2048), (2, 13, 256, 512, 1024, 2048)):
    g = torch.Generator(device=dev).manual_seed(1)
    lg = torch.randn((B, O, hf, wf), generator=g, device=dev)
    em = HyperMapper(1.0).expmap(torch.randn((B, C, hf, wf), generator=g, device=dev) * 0.1, dim=1)
    want = score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(em, (H, W)), "entropy", "radius", True, None, size=3)
    for env in ({}, {"HALO_LR_PPT4": "1"}, {"HALO_LR_NOFIXED": "1"}, {"HALO_LR_NODMA": "1"}):
        for k in ("HALO_LR_PPT4", "HALO_LR_NOFIXED", "HALO_LR_NODMA"): os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(2):
            got = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="exact")
            torch.cuda.synchronize()
            a, b = got[1].cpu().numpy(), want[1].cpu().numpy()
            bad = np.argwhere(a.view(np.int64) != b.view(np.int64))
            msg = "ok" if len(bad) == 0 else "MISMATCH %d px, first %s, rows %s..%s cols %s..%s, max rel %.3g" % (len(bad), bad[0], bad[:, 1].min(), bad[:, 1].max(), bad[:, 2].min(), bad[:, 2].max(), np.nanmax(np.abs(a - b) / np.abs(b)))
            print((B, C, hf, wf), env, rep, msg, flush=True)
