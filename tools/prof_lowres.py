"""Profiling aid: a few fused low-resolution scoring passes (x4 head outputs -> 1024x2048, C=256, 16 images) to run
under rocprofv3 --pmc ... --kernel-trace: both low-res modes (k_feat_reduce_lr_dmaf; k_gram_lr + k_radius_gram; k_logit_maps_lr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps_lowres
dev = torch.device("cuda:0")
B, C, O, h, w, H, W = 16, 256, 19, 256, 512, 1024, 2048
g = torch.Generator(device=dev).manual_seed(3)
logit = torch.randn((B, O, h, w), generator=g, device=dev)
feat = torch.randn((B, C, h, w), generator=g, device=dev, dtype=torch.float64) * 0.05
for mode in ("exact", "gram"):
    for _ in range(4):
        score_maps_lowres(logit, feat, (H, W), "entropy", "radius", True, None, ksize=3, mode=mode)
torch.cuda.synchronize()
