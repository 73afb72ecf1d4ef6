"""Timing aid: the RegionSelection driver end to end (staging, fused resize+score, greedy selection, PNG +
indicator files) on full-size synthetic pool images, serial vs pipelined (one MI355X).  The backbone is a
stand-in that only sleeps on the GPU for a configurable time, so the number isolates the acquisition side."""
import os, shutil, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd
# hardware queues: ROCm's default (4) unless HALO_RS_HW_QUEUES says otherwise -- what a training process gets without asking, and
# what this driver measures fastest on since round 6 (two queues: two single-workgroup sweeps in flight, 0.58 ms/image; four or
# eight: the writers' floor, 0.47-0.51: profiles/r06_hw_queues.txt; bench.py's 16-image steps still prefer two)
if os.environ.get("HALO_RS_HW_QUEUES"):
    halo_amd.configure(hw_queues=int(os.environ["HALO_RS_HW_QUEUES"]))
from halo_amd.core.active.build import RegionSelection
from halo_amd.core.utils.hyperbolic import HyperMapper

dev = torch.device("cuda:0")
N, H, W, C, O = int(os.environ.get('HALO_RS_IMAGES', '96')), 1024, 2048, 64, 19
cfg = types.SimpleNamespace(
    MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
    ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                 BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
g = torch.Generator(device=dev).manual_seed(0)
emb = HyperMapper(1.0).expmap(torch.randn((1, C, 160, 320), generator=g, device=dev) * 0.1, dim=1)
logit = torch.nn.functional.interpolate(torch.randn((1, O, 160, 320), generator=g, device=dev), size=(640, 1280), mode="bilinear", align_corners=True)


class Ident(torch.nn.Module):
    def forward(self, x):
        return x


class Head(torch.nn.Module):
    def __init__(self, busy_ms):
        super().__init__()
        self.busy_ms = busy_ms
        self.a = torch.randn((4096, 4096), device=dev)

    def forward(self, x, size=None):
        nb = x.shape[0]
        if self.busy_ms > 0:                                          # stand-in for the backbone forward:
            if MODE == "sleep":                                       #   a spin kernel on one CU (GPU mostly idle)
                torch.cuda._sleep(int(self.busy_ms * 1e-3 * 2.1e9))
            else:                                                     #   or back-to-back full-GPU GEMMs
                for _ in range(int(self.busy_ms / GEMM_MS)):
                    self.a @ self.a
            torch.cuda.current_stream().synchronize()
        return (logit, emb) if nb == 1 else (logit.expand(nb, -1, -1, -1).contiguous(), emb.expand(nb, -1, -1, -1).contiguous())


def pool(tmp):
    # pinned like the reference's DataLoader(pin_memory=True) batches (train_learners.py:283-289)
    gt = torch.randint(0, O, (1, H, W)).pin_memory()
    return [{"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
             "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64).pin_memory(), "origin_label": gt, "size": torch.tensor([[H, W]]),
             "active": torch.zeros(1, H, W, dtype=torch.bool).pin_memory(), "selected": torch.zeros(1, H, W, dtype=torch.bool).pin_memory(),
             "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"img{i}"]} for i in range(N)]


GEMM_MS = 1.0
a_ = torch.randn((4096, 4096), device=dev); a_ @ a_; torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    a_ @ a_
torch.cuda.synchronize(); GEMM_MS = (time.perf_counter() - t0) / 20 * 1e3
print(f"4096^3 f32 GEMM: {GEMM_MS:.2f} ms")
def fmt(st):
    n = max(1, st["images"])
    main = "  ".join("%s %.2f" % (k[5:-2], st[k] / n * 1e3) for k in ("main_loader_s", "main_forward_s", "main_stage_s", "main_launch_s", "main_wait_slot_s"))
    wr = "  ".join("%s %.2f" % (k[7:-2], st[k] / n * 1e3) for k in ("writer_event_wait_s", "writer_copy_s", "writer_png_s", "writer_save_s"))
    return "      per image, main thread [ms]: %s | writer threads (%d) [ms, summed over threads]: %s" % (main, st["writer_threads"], wr)


def batched(items, nb):
    """the same pool through a loader of batch size nb (default collate: tensors concatenated, lists extended)"""
    out = []
    for k in range(0, len(items), nb):
        grp = items[k:k + nb]
        out.append({key: (torch.cat([g_[key] for g_ in grp]).pin_memory() if torch.is_tensor(grp[0][key]) else sum((g_[key] for g_ in grp), []))
                    for key in grp[0]})
    return out


STAGING = os.environ.get("HALO_RS_STAGING", "table")
REPEATS = int(os.environ.get("HALO_RS_REPEATS", "5"))
for MODE, busy in (() if os.environ.get("HALO_RS_FLOOR_ONLY") else (("none", 0.0), ("sleep", 30.0), ("gemm", 30.0))):
    CONFIGS = ((0, 1, 1, "serial (in_flight=0, 1 writer)"), (8, 2, 1, "pipelined (in_flight=8, 4 streams, 2 writers: a rank's share of 16 cores under 8 ranks)"),
               (8, 8, 1, "pipelined (in_flight=8, 4 streams, 8 writers)"), (8, 16, 1, "pipelined (in_flight=8, 4 streams, 16 writers)"),
               (8, 12, 1, "pipelined (in_flight=8, 4 streams, 12 writers)"), (None, None, 1, "pipelined (defaults)"),
               (None, None, 2, "pipelined (defaults), loader batch 2"), (None, None, 4, "pipelined (defaults), loader batch 4"))
    if os.environ.get("HALO_RS_SWEEP"):            # depth x writers sweep (no backbone only)
        if MODE != "none":
            continue
        CONFIGS = tuple((d, w, 1, "in_flight=%d, %d writers" % (d, w)) for d in (8, 12, 16, 24) for w in (8, 12, 16))
    for (infl, wr, nb, tag) in CONFIGS:
        if MODE != "none" and (wr in (2, 12, 16) or nb > 1):
            continue
        tmp = tempfile.mkdtemp(prefix="halo_rs_t_")
        items = batched(pool(tmp), nb) if nb > 1 else pool(tmp)
        kw = {} if infl is None else {"in_flight": infl}
        if os.environ.get("HALO_RS_STREAMS"):
            kw["streams"] = int(os.environ["HALO_RS_STREAMS"])
        RegionSelection(cfg, Ident(), Head(busy), items[:8], 1, writer_threads=wr, mask_staging=STAGING, **kw)
        torch.cuda.synchronize()
        # without a backbone a round over the pool takes ~0.1 s and its 0.4 GB of files land in the page cache: the number is
        # the MEDIAN of HALO_RS_REPEATS rounds (default 5; the files are overwritten), the range beside it
        runs = []
        for _ in range(REPEATS if MODE == "none" else 1):
            st = {}
            t0 = time.perf_counter()
            RegionSelection(cfg, Ident(), Head(busy), items, 1, writer_threads=wr, stats=st, mask_staging=STAGING, **kw)
            runs.append((time.perf_counter() - t0, st))
        runs.sort(key=lambda r: r[0])
        dt, st = runs[len(runs) // 2]
        rng = "" if len(runs) == 1 else "  [%d rounds: %.2f .. %.2f]" % (len(runs), runs[0][0] / N * 1e3, runs[-1][0] / N * 1e3)
        print(f"backbone stand-in {MODE:5s} {busy:4.0f} ms: {tag:48s} {dt / N * 1e3:7.2f} ms/image  ({N / dt:6.1f} images/s){rng}")
        print(fmt(st), flush=True)
        shutil.rmtree(tmp, ignore_errors=True)


def writers_alone(n_threads):
    """The host side's own floor: halo_retire_image (mask composed from the pick table, PNG, indicator maps composed, CRC-32,
    4.3 MB written) for the same N images from n_threads threads, with NO GPU work, launching thread or slot logic around it."""
    import threading
    import numpy as np
    from halo_amd import _hostlib
    from halo_amd.core.active.build import _IndicatorTemplate
    tmp = tempfile.mkdtemp(prefix="halo_rs_w_", dir=os.environ.get("HALO_RS_TMP"))            # HALO_RS_TMP=/dev/shm: the same work without the disk-backed file system
    rng = np.random.default_rng(0)
    om = np.full((H, W), 255, np.int64); gt = rng.integers(0, O, (H, W)).astype(np.int64)
    act = np.zeros((H, W), np.bool_); sel = np.zeros((H, W), np.bool_)
    k = 2331
    picks = np.stack([rng.integers(0, H, k), rng.integers(0, W, k), rng.random(k)], 1).astype(np.float64)
    tpl = _IndicatorTemplate.get((H, W))
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= N:
                return
            _hostlib.retire_image(os.path.join(tmp, f"m{i}.png"), os.path.join(tmp, f"i{i}.pth"), om, gt, picks, k, 1, act, sel,
                                  None if os.environ.get("HALO_RS_NO_INDICATOR") else tpl, compose_mask_radius=5)
    runs = []
    for _ in range(REPEATS + 1):
        nxt[0] = 0
        if os.environ.get("HALO_RS_FRESH"):          # every round into NEW files (the default: rewritten in place, as the rounds of a run do)
            for f in os.listdir(tmp):
                os.remove(os.path.join(tmp, f))
        th = [threading.Thread(target=work) for _ in range(n_threads)]
        t0 = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        runs.append(time.perf_counter() - t0)
    shutil.rmtree(tmp, ignore_errors=True)
    runs = sorted(runs[1:])
    return runs[len(runs) // 2] / N * 1e3


if not os.environ.get("HALO_RS_SWEEP"):
    for nt in (1, 2, 4, 8, 16):
        print(f"host floor: {nt:2d} writer thread(s) alone, no GPU in the loop (halo_retire_image x {N}): {writers_alone(nt):6.2f} ms/image", flush=True)
