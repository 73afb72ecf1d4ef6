"""Profiling aid: forward + backward of the head tail at the training shape (2 x 64 x 160 x 320) -- run under
rocprofv3 --kernel-trace --stats to see which kernels the 0.8 ms are."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR

dev = torch.device("cuda:0")
feat = (torch.randn((2, 64, 160, 320), device=dev) * 0.1).requires_grad_(True)
mapper, mlr = HyperMapper(1.0), HyperMLR(64, 19).to(dev)


def head():
    emb = mapper.expmap(feat, dim=1)
    out = mlr._hyper_logits(emb, out_dtype=torch.float32)      # = mlr(emb).float() as halo_amd/core/models/classifier.py's training tail calls it
    out.sum().backward()


for _ in range(3):
    head()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    head()
torch.cuda.synchronize()
print("head tail fwd+bwd: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
