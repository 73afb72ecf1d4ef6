"""Evidence aid: the low-res scorer's exact and Gram modes on full-size synthetic images of the bench's distribution
(x4 head outputs, C = 256, 1024x2048, 2331 regions): largest difference of the score maps and the number of images whose
selection (pick list and masks) differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.build import acquire_batch_lowres
from halo_amd.core.active.floating_region import score_maps_lowres
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR

dev = torch.device("cuda:0")
N, B, C, O, h, w, H, W = int(os.environ.get("N", 64)), 8, 256, 19, 256, 512, 1024, 2048
n_regions = 2331
mlr = HyperMLR(C, O).to(dev)
diff_images, max_abs, max_rel = 0, 0.0, 0.0
for i0 in range(0, N, B):
    g = torch.Generator(device=dev).manual_seed(1234 + i0)
    z = torch.randn((B, C, h, w), generator=g, device=dev) * 0.1
    emb = HyperMapper(1.0).expmap(z, dim=1)
    with torch.no_grad():
        logit = mlr._hyper_logits(emb, out_dtype=torch.float32)
    gt = torch.randint(0, O, (B, H, W), generator=g, device=dev)
    res = []
    for mode in ("exact", "gram"):
        sc, _, _ = score_maps_lowres(logit, emb, (H, W), "entropy", "radius", True, None, ksize=3, mode=mode)
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        picks, npk = acquire_batch_lowres(logit, emb, (H, W), gt, act, sel, am, unc_type="entropy", pur_type="radius", normalize=True,
                                          n_regions=n_regions, active_radius=1, mask_radius=5, lowres_mode=mode)
        res.append((sc, picks[:, :, :2].clone(), npk.clone(), act, am))
    d = (res[0][0] - res[1][0]).abs()
    max_abs = max(max_abs, float(d.max()))
    max_rel = max(max_rel, float((d / res[0][0].abs().clamp_min(1e-300)).max()))
    for b in range(B):
        same = torch.equal(res[0][1][b], res[1][1][b]) and int(res[0][2][b]) == int(res[1][2][b]) and \
               torch.equal(res[0][3][b], res[1][3][b]) and torch.equal(res[0][4][b], res[1][4][b])
        diff_images += 0 if same else 1
print(f"{N} images, {n_regions} regions each: score maps differ by at most {max_abs:.3e} absolute ({max_rel:.3e} relative); "
      f"images whose selection differs between the modes: {diff_images}")
