"""Profiling aid: the stand-alone logit kernel (entropy + arg-max of 19-class logits, 16 x 1024x2048) a few times, to run
under rocprofv3 --kernel-trace / --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.floating_region import score_maps
dev = torch.device("cuda:0")
B, O, H, W = 16, 19, 1024, 2048
g = torch.Generator(device=dev).manual_seed(5)
low = torch.randn((B, O, H // 4, W // 4), generator=g, device=dev)
logit = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True).contiguous()
for _ in range(5):
    score_maps(logit, None, "entropy", "ripu", False, None, size=3, want_maps=False)
torch.cuda.synchronize()
