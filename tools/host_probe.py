"""Diagnostic: what the GPU box's host gives this process (CPU count / cgroup quota) and how the writer-side work of
RegionSelection scales over threads (pure CPU: direct PNG writer + torch.save)."""
import os, sys, tempfile, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo_amd.core.active.build import _persist
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try:
        print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e:
        print(f, "-", type(e).__name__)
H, W = 1024, 2048
rng = np.random.default_rng(0)
mask = np.full((H, W), 255, np.uint8)
for _ in range(2331):
    y, x = rng.integers(1, H - 1), rng.integers(1, W - 1)
    mask[y - 1:y + 2, x - 1:x + 2] = rng.integers(0, 19, (3, 3))
act = torch.from_numpy(mask != 255)
tmp = tempfile.mkdtemp()
for nt in (1, 2, 4, 8, 16):
    N = 64
    def work(k):
        for i in range(k, N, nt):
            _persist(mask.copy(), act.clone(), act.clone(), os.path.join(tmp, f"m{i}.png"), os.path.join(tmp, f"i{i}.pth"))
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{nt:2d} writer threads: {dt / N * 1e3:6.2f} ms/image")
try:
    print("/sys/fs/cgroup/cpu.stat", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | "))
except Exception:
    pass
