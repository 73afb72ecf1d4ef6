"""A/B aid for the Gram low-res route under rocprofv3 (tools/kstats.sh): k_gram_lr2<UCH, NT> variants (HALO_GRAM_UCH / HALO_GRAM_NT, read per
call) at the bench shape (x4 head outputs -> 1024x2048, C=256, 16 images)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.active.floating_region import score_maps_lowres
dev = torch.device("cuda:0")
B, C, O, h, w, H, W = 16, 256, 19, 256, 512, 1024, 2048
g = torch.Generator(device=dev).manual_seed(3)
logit = torch.randn((B, O, h, w), generator=g, device=dev)
feat = torch.randn((B, C, h, w), generator=g, device=dev, dtype=torch.float64) * 0.05
ref = None
for rep in range(3):
    for uch, nt, rows in ((2, 0, 2), (4, 0, 2), (2, 1, 2), (1, 0, 4), (2, 0, 4), (3, 0, 4), (1, 0, 8)):
        if True:
            os.environ["HALO_GRAM_UCH"], os.environ["HALO_GRAM_NT"], os.environ["HALO_GRAM_ROWS"] = str(uch), str(nt), str(rows)
            out = score_maps_lowres(logit, feat, (H, W), "entropy", "radius", True, None, ksize=3, mode="gram")
            if ref is None:
                ref = [o.clone() for o in out]
            else:
                assert all(torch.equal(a, b) for a, b in zip(ref, out)), (uch, nt, rows)
torch.cuda.synchronize()
print("all variants bit-identical")
