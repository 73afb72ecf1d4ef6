"""Randomised differential test of the selector alone on one MI355X against the CPU oracle, biased towards what the value-binned
sweep finds hard: plateaus of exact ties (quantised maps, clipped maps, constant regions along borders), near-ties, -inf holes,
NaN / +inf, tiny and oblong maps, every mask radius the sweep serves and some it does not.  Picks, masks and the mutated map must be
the oracle's bit for bit whatever the hand-over counters say; the counters are tallied.  `python tests/fuzz_select.py [n] [seed]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import halo_amd  # noqa: F401
from halo_amd import _lib
from halo_amd.core.active.build import greedy_select
from oracle import halo_oracle as ho

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(SEED)
tally = {}
t0 = time.time()
for i in range(N):
    H, W = int(rng.integers(6, 260)), int(rng.integers(6, 420))
    if rng.random() < 0.1:
        H, W = int(rng.integers(1, 8)), int(rng.integers(1, 600))
    B = int(rng.choice([1, 1, 2, 5]))
    mrad = int(rng.choice([1, 2, 3, 5, 5, 5, 9, 14, 0, 16]))
    arad = int(rng.choice([0, 1, 1, 2]))
    n = int(rng.choice([1, 7, 60, 300, 900]))
    dt = np.float64 if rng.random() < 0.6 else np.float32
    maps = []
    for b in range(B):
        base = ho.bilinear(rng.standard_normal((1, max(2, H // 6), max(2, W // 6))), (H, W))[0]
        kind = rng.choice(["quant", "clip_top", "clip_low", "const_regions", "near", "noise_quant", "smooth", "two"])
        if kind == "quant":
            lv = int(rng.integers(1, 30)); m = np.round(base * lv) / lv
        elif kind == "clip_top":
            m = np.minimum(base, np.quantile(base, rng.uniform(0.3, 0.95)))
        elif kind == "clip_low":
            m = np.maximum(base, np.quantile(base, rng.uniform(0.1, 0.9)))
        elif kind == "const_regions":
            m = base.copy()
            for _ in range(int(rng.integers(1, 5))):
                y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
                m[y0:y0 + int(rng.integers(1, H + 1)), x0:x0 + int(rng.integers(1, W + 1))] = float(rng.choice([2.0, -0.5, 0.0, -0.0]))
        elif kind == "near":
            m = np.where(base > 0, 1.0 + 1e-13 * rng.standard_normal((H, W)), base)
        elif kind == "noise_quant":
            m = np.round(rng.standard_normal((H, W)) * 2) / 2
        elif kind == "two":
            m = np.where(rng.random((H, W)) < 0.5, 1.0, 0.0)
        else:
            m = base
        m = m.astype(dt)
        if rng.random() < 0.35:
            m[rng.random((H, W)) < rng.uniform(0.05, 0.7)] = -np.inf
        if rng.random() < 0.05:
            m[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.nan
        if rng.random() < 0.05:
            m[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.inf
        maps.append(m)
    s0 = np.ascontiguousarray(np.stack(maps))
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    prior = rng.random((B, H, W)) < 0.02
    s = torch.from_numpy(s0).to(dev).clone()
    act = torch.from_numpy(prior).to(dev).clone(); sel = torch.zeros_like(act)
    am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    hov = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    picks, npk = greedy_select(s, n, arad, mrad, act, sel, am, torch.from_numpy(gt).to(dev), handover=hov)
    desc = dict(i=i, H=H, W=W, B=B, mrad=mrad, arad=arad, n=n, dt=str(np.dtype(dt)))
    hv = hov.cpu().numpy()
    for b in range(B):
        so = s0[b].copy(); a_o = prior[b].copy(); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
        _, _, _, _, po = ho.select_pixels_to_label(so, n, arad, mrad, a_o, s_o, m_o, gt[b], True)
        k = int(npk[b])
        pk = picks[b, :k].cpu().numpy()
        ok = k == len(po) and pk.shape == po.shape and np.array_equal(pk.view(np.int64), np.ascontiguousarray(po).view(np.int64))
        ok = ok and np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(sel[b].cpu().numpy(), s_o) and np.array_equal(am[b].cpu().numpy(), m_o)
        got = s[b].cpu().numpy()
        ok = ok and bool(np.all((got == so) | (np.isnan(got) & np.isnan(so))))
        if not ok:
            print("MISMATCH", dict(desc, b=b, k=k, want=len(po), reason=_lib.SWEEP_REASONS[int(hv[b, 0])])); sys.exit(1)
        r = _lib.SWEEP_REASONS[int(hv[b, 0])]
        tally[r] = tally.get(r, 0) + 1
    if i % 50 == 0:
        print("case %d ok %s" % (i, desc), flush=True)
print("fuzz_select: %d cases, seed %d, %.0f s: HIP == oracle bit for bit (picks, masks, mutated maps); sweep outcomes per image %s" % (N, SEED, time.time() - t0, tally))
