"""Profiling aid: 16-image selection alone (phase A) and beside the streaming feature kernel (phase B), for a rocprofv3 kernel
trace.  `python tools/prof_select16.py summarize <trace dir>` prints the per-kernel medians of each phase (the phases are
separated by a 0.5 s pause)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(d):
    import collections, csv, glob, statistics as st
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    sel = [r for r in rows if "k_sel_" in r["Kernel_Name"] or "k_greedy" in r["Kernel_Name"]]
    # the range records of the RANGED runs are prepared outside the timed call (k_sel_zero + k_sel_range back to back, no
    # k_sel_hist1 behind them): drop that pair
    keep = []
    i = 0
    while i < len(sel):
        n0 = sel[i]["Kernel_Name"]
        n1 = sel[i + 1]["Kernel_Name"] if i + 1 < len(sel) else ""
        n2 = sel[i + 2]["Kernel_Name"] if i + 2 < len(sel) else ""
        if "k_sel_zero" in n0 and "k_sel_range" in n1 and "k_sel_hist1" not in n2:
            i += 2
            continue
        keep.append(sel[i]); i += 1
    sel = keep
    # split at the biggest pause between selection kernels
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), i) for i, (a, b) in enumerate(zip(sel[:-1], sel[1:]))]
    cut = max(gaps)[1] + 1
    for name, part in (("alone", sel[:cut]), ("beside streaming", sel[cut:])):
        per = collections.defaultdict(list)
        for r in part:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        ncall = max(len(v) for v in per.values())
        # wall per call: first selection kernel start -> last end, per group of one call's launches
        print("== %s (%d calls)" % (name, ncall))
        tot = 0.0
        for k, v in sorted(per.items(), key=lambda kv: -st.median(kv[1]) * len(kv[1])):
            med = st.median(v)
            tot += med * len(v) / ncall
            print("  %-46s x%.1f  median %8.1f us" % (k, len(v) / ncall, med))
        print("  sum of medians per call: %.1f us" % tot)


if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    summarize(sys.argv[2]); sys.exit(0)

import torch
import halo_amd; halo_amd.configure(hw_queues=2)      # before the first HIP call: the acquisition's measured optimum
from halo_amd.core.active.build import greedy_select
from halo_amd.core.active.floating_region import score_maps
dev = torch.device("cuda:0")
H, W, n, B = 1024, 2048, 2331, 16
g = torch.Generator(device=dev).manual_seed(3)
base = torch.randn((B, H // 4, W // 4), generator=g, device=dev, dtype=torch.float64)
score0 = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=True)[0].contiguous()
gt = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
feat = torch.randn((4, 256, H, W), device=dev, dtype=torch.float64) * 0.01
logit = torch.randn((4, 19, H, W), device=dev)
s2 = torch.cuda.Stream(dev)
hi = torch.cuda.Stream(dev, priority=-1)


from halo_amd import _lib
from halo_amd.core.active.floating_region import new_score_range
RANGED = os.environ.get("RANGED", "1") == "1"       # the pipeline's case: the scorer supplies the score maps' value range


def run(loaded):
    sc = score0.clone()
    rec = None
    if RANGED:
        rec = new_score_range(B, dev)
        _lib.check(_lib.lib().halo_score_range(_lib.ptr(sc), _lib.dtype_code(sc), B, H, W, _lib.ptr(rec), _lib.stream_ptr(dev)), "halo_score_range")
    act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
    am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if loaded:
        with torch.cuda.stream(s2):
            for _ in range(10):
                score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
        time.sleep(0.01)
    t0 = time.perf_counter()
    with torch.cuda.stream(hi):
        greedy_select(sc, n, 1, 5, act, sel, am, gt, score_range=rec)
    hi.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return dt


run(False)
print("alone", ["%.2f" % run(False) for _ in range(6)])
time.sleep(0.5)
run(True)
print("beside", ["%.2f" % run(True) for _ in range(6)])
