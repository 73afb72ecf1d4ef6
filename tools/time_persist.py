"""Timing aid: the host-side pieces of retiring one 1024x2048 image in RegionSelection (pinned staging buffers,
device->host copies, PNG encode at several zlib levels, torch.save of the indicator)."""
import io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

dev = torch.device("cuda:0")
H, W = 1024, 2048
rng = np.random.default_rng(0)
mask = np.full((H, W), 255, np.uint8)
for _ in range(2331):
    y, x = rng.integers(1, H - 1), rng.integers(1, W - 1)
    mask[y - 1:y + 2, x - 1:x + 2] = rng.integers(0, 19, (3, 3))
act = torch.from_numpy(mask != 255)
tmp = tempfile.mkdtemp()


def t(fn, n=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


d_mask = torch.from_numpy(mask).to(dev)
print("pinned alloc 2MB (cached): %.3f ms" % t(lambda: torch.empty((H, W), dtype=torch.uint8, pin_memory=True)))
hb = torch.empty((H, W), dtype=torch.uint8, pin_memory=True)
print("D2H 2MB into pinned + sync: %.3f ms" % t(lambda: (hb.copy_(d_mask, non_blocking=True), torch.cuda.synchronize())))
print(".cpu() 2MB: %.3f ms" % t(lambda: d_mask.cpu()))
print("clone 2MB bool: %.3f ms" % t(lambda: act.clone()))
for lvl in (None, 6, 3, 1, 0):
    kw = {} if lvl is None else {"compress_level": lvl}
    ms = t(lambda: Image.fromarray(mask).save(os.path.join(tmp, "m.png"), **kw))
    print("PNG save level %s: %.2f ms, %d bytes" % (lvl, ms, os.path.getsize(os.path.join(tmp, "m.png"))))
print("torch.save indicator: %.2f ms" % t(lambda: torch.save({"active": act, "selected": act}, os.path.join(tmp, "i.pth"))))
pa = torch.empty((H, W), dtype=torch.bool, pin_memory=True); pa.copy_(act)
print("torch.save indicator (pinned tensors): %.2f ms" % t(lambda: torch.save({"active": pa, "selected": pa}, os.path.join(tmp, "i.pth"))))
big = torch.full((1, H, W), 255, dtype=torch.int64).pin_memory()
print("H2D 16.8MB pinned + sync: %.3f ms" % t(lambda: (big.to(dev, non_blocking=True), torch.cuda.synchronize())))
big2 = torch.full((1, H, W), 255, dtype=torch.int64)
print("H2D 16.8MB pageable: %.3f ms" % t(lambda: (big2.to(dev, non_blocking=True), torch.cuda.synchronize())))
# host-side narrowing of a loader mask (int64 -> its low byte, what the uint8 PNG keeps anyway) into a pinned buffer
src = torch.randint(0, 256, (H, W), dtype=torch.int64).pin_memory()
dst = torch.empty((H, W), dtype=torch.uint8, pin_memory=True)
sn, dn = src.numpy(), dst.numpy()
print("narrow int64 -> uint8, numpy astype into pinned: %.2f ms" % t(lambda: np.copyto(dn, sn, casting="unsafe")))
lowbyte = sn.view(np.uint8).reshape(H, W, 8)[:, :, 0]
print("narrow int64 -> uint8, low-byte strided view copy: %.2f ms" % t(lambda: np.copyto(dn, lowbyte)))
print("narrow int64 -> uint8, torch .to(uint8) copy_: %.2f ms" % t(lambda: dst.copy_(src)))
import threading
for nt in (2, 4, 8):
    srcs = [src.clone().pin_memory() for _ in range(nt)]; dsts = [torch.empty((H, W), dtype=torch.uint8, pin_memory=True) for _ in range(nt)]
    def work(k):
        for _ in range(10):
            np.copyto(dsts[k].numpy(), srcs[k].numpy(), casting="unsafe")
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    [x.start() for x in th]; [x.join() for x in th]
    print("  %d threads narrowing concurrently: %.2f ms per mask per thread" % (nt, (time.perf_counter() - t0) / 10 * 1e3))
d64 = torch.empty((H, W), dtype=torch.int64, device=dev)
print("H2D 16.8MB pinned into an EXISTING device tensor + sync: %.3f ms" % t(lambda: (d64.copy_(src, non_blocking=True), torch.cuda.synchronize())))
d8 = torch.empty((H, W), dtype=torch.uint8, device=dev)
print("H2D 2MB pinned uint8 + widen on device + sync: %.3f ms" % t(lambda: (d8.copy_(dst, non_blocking=True), d64.copy_(d8), torch.cuda.synchronize())))
