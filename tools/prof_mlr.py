"""Profiling aid: a few HyperMLR launches at 1024x2048, C=256 (run under rocprofv3 --pmc ... --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
dev = torch.device("cuda:0")
x = HyperMapper(1.0).expmap(torch.randn((1, 256, 1024, 2048), device=dev) * 0.1, dim=1)
mlr = HyperMLR(256, 19).to(dev)
with torch.no_grad():
    for _ in range(4):
        mlr._hyper_logits(x, out_dtype=torch.float32)
torch.cuda.synchronize()
