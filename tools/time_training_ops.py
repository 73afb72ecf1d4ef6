"""Timing aid: forward + backward of the training-side ops at the reference's training shapes (one MI355X):
head tail (expmap -> HyperMLR, batch 2, C=64, 160x320) and the two window losses (2 x 19 x 640 x 1280)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.loss import LocalConsistentLoss, NegativeLearningLoss
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR

dev = torch.device("cuda:0")


def timeit(fn, n=10, repeats=5):
    """best of `repeats` timings of n calls (after five warm-up calls: the caching allocator settles -- a hipMalloc inside the timed
    loop costs milliseconds -- and a busy host thread no longer shows up as 0.2 ms on a 0.3 ms number)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


feat = (torch.randn((2, 64, 160, 320), device=dev) * 0.1).requires_grad_(True)
mapper, mlr = HyperMapper(1.0), HyperMLR(64, 19).to(dev)


def head():
    emb = mapper.expmap(feat, dim=1)
    out = mlr._hyper_logits(emb, out_dtype=torch.float32)      # = mlr(emb).float() as halo_amd/core/models/classifier.py's training tail calls it
    out.sum().backward()


print(f"head tail fwd+bwd (2x64x160x320): {timeit(head):.3f} ms")
with torch.no_grad():
    print(f"head tail fwd only             : {timeit(lambda: mlr._hyper_logits(mapper.expmap(feat, dim=1), out_dtype=torch.float32)):.3f} ms")
x = torch.randn((2, 19, 640, 1280), device=dev, requires_grad=True)
label = torch.randint(0, 19, (2, 160, 320), device=dev).repeat_interleave(4, 1).repeat_interleave(4, 2)
for lt in ("l1", "kl"):
    crit = LocalConsistentLoss(19, lt)
    print(f"LocalConsistentLoss {lt} fwd+bwd (2x19x640x1280): {timeit(lambda: crit(x, label).backward()):.3f} ms")
p = torch.softmax(x.detach(), dim=1).requires_grad_(True)
nl = NegativeLearningLoss()
print(f"NegativeLearningLoss fwd+bwd                     : {timeit(lambda: nl(p).backward()):.3f} ms")
