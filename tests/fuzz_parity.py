"""Randomised differential test on one MI355X: HIP path vs the CPU oracle, bit for bit, over random shapes / branches / window
sizes / padding modes / dtypes / low-res geometries and modes / selection parameters -- the configurations the fixed parity tests
do not enumerate.  Test infrastructure (it drives the CPU oracle, so it lives under tests/); not collected by pytest (minutes): `python tests/fuzz_parity.py [n_cases] [seed]`, one line per case, a summary at the end;
exit code 1 on the first mismatch (the case's parameters are printed so that it can be replayed with the same seed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import halo_amd  # noqa: F401
from halo_amd.core.active.build import acquire_batch, acquire_batch_lowres, greedy_select
from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
from oracle import halo_oracle as ho

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(SEED)
UNC = ["entropy", "pixel_entropy", "oracle_acc", "none", "hyperbolic"]
PUR = ["ripu", "oracle_ripu", "hyper", "none", "radius", "euc_norm"]
PAD = ["zeros", "zeros", "zeros", "reflect", "replicate", "circular"]


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.int64 if a.dtype == np.float64 else np.int32), b.view(np.int64 if b.dtype == np.float64 else np.int32)) or \
            (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]) and
             np.array_equal(np.signbit(a[~np.isnan(a)]), np.signbit(b[~np.isnan(b)])))
    return np.array_equal(a, b)


def one_case(i):
    O = int(rng.choice([19, 19, 16, 7, 2, 33]))
    C = int(rng.choice([1, 3, 8, 20, 42, 64]))
    lowres = rng.random() < 0.4
    H, W = int(rng.integers(5, 70)), int(rng.integers(5, 140))
    if rng.random() < 0.08:                                          # a few larger maps: several tiles / strips / bins
        H, W = int(rng.integers(70, 300)), int(rng.integers(140, 600))
    if rng.random() < 0.5:
        W = W // 4 * 4 + 4                                       # the aligned fast paths
    B = int(rng.integers(1, 4))
    unc, pur, pad = str(rng.choice(UNC)), str(rng.choice(PUR)), str(rng.choice(PAD))
    norm = bool(rng.random() < 0.6)
    size = int(rng.choice([1, 3, 3, 3, 5, 7]))
    if pad == "reflect" and size // 2 >= min(H, W):
        size = 3
    K = int(rng.choice([2, 10, 100, 100, 17, 300, 4100]))          # (bins on either side of the class sum's flush boundaries: 16, 256, 4096)
    f32 = rng.random() < 0.25
    c = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
    psize = 3 if pur == "hyper" else size
    desc = dict(i=i, O=O, C=C, H=H, W=W, B=B, unc=unc, pur=pur, pad=pad, norm=norm, size=size, K=K, f32=f32, c=c, lowres=lowres)
    gt = rng.integers(0, O, (B, H, W)).astype(np.int64)
    gt[rng.random((B, H, W)) < 0.05] = 255
    act = rng.random((B, H, W)) < 0.03
    scale = float(rng.choice([0.05, 0.3, 1.5]))
    if lowres:
        hl, wl = int(rng.integers(1, H + 1)), int(rng.integers(1, W + 1))
        hf, wf = int(rng.integers(1, H + 1)), int(rng.integers(1, W + 1))
        mode = "exact" if (f32 or rng.random() < 0.5) else "gram"
        if rng.random() < 0.35:
            # head-like geometry: the embedding upsampled x3 ... x8 from an even-width grid, enough channels for several LDS
            # chunks with a partial last one -- the LDS-DMA kernels with compile-time window geometry (4 and 8 pixels per lane)
            f = float(rng.choice([3.0, 4.0, 5.0, 6.4, 8.0]))
            if H < 40:
                H = int(rng.integers(40, 200))
            if W < 70:
                W = int(rng.integers(70, 400))
            hf, wf = max(2, int(round(H / f))), max(2, int(round(W / f)) // 2 * 2)
            C = int(rng.integers(5, 71))
            gt = rng.integers(0, O, (B, H, W)).astype(np.int64)
            act = rng.random((B, H, W)) < 0.03
            hl, wl = min(hl, H), min(wl, W)
            desc.update(H=H, W=W, C=C)
        desc.update(hl=hl, wl=wl, hf=hf, wf=wf, mode=mode)
        logit_lr = (rng.standard_normal((B, O, hl, wl)) * 2).astype(np.float32)
        emb_lr = ho.expmap((rng.standard_normal((B, C, hf, wf)) * scale).astype(np.float32), c, dim=1)
        if f32:
            emb_lr = emb_lr.astype(np.float32)
        try:
            got = score_maps_lowres(t(logit_lr), t(emb_lr), (H, W), unc, pur, norm, t(gt), ksize=size, purity_size=psize, K=K, c=c,
                                    active=t(act), mode=mode, padding_mode=pad)
        except halo_amd._lib.HaloUnsupported:
            return "declined", desc
        for b in range(B):
            lg = ho.bilinear(logit_lr[b:b + 1], (H, W))
            need = pur in ("hyper", "radius", "euc_norm")
            raw = None
            em = None
            if need and mode == "gram" and not f32:
                raw = ho.gram_radius(emb_lr[b], (H, W), "euc_norm" if pur == "euc_norm" else "radius", c)
            elif need:
                em = ho.bilinear(emb_lr[b:b + 1], (H, W))
            so, io, uo = ho.floating_region_score(lg, em, unc, pur, norm, gt[b], size=size, purity_type=pur, K=K, c=c, impurity_raw=raw,
                                                  padding_mode=pad)
            so = so.copy(); so[act[b]] = -np.inf
            if not (same(got[0][b].cpu().numpy(), so) and same(got[1][b].cpu().numpy(), io) and same(got[2][b].cpu().numpy(), uo)):
                return "MISMATCH(lowres maps, image %d)" % b, desc
        score_dev = got[0]
    else:
        logit = (rng.standard_normal((B, O, H, W)) * 2).astype(np.float32)
        emb = ho.expmap((rng.standard_normal((B, C, H, W)) * scale).astype(np.float32), c, dim=1)
        if f32:
            emb = emb.astype(np.float32)
        if rng.random() < 0.1:
            logit[0, :, 0, 0] = np.nan
        if rng.random() < 0.1:
            logit[-1, 0, -1, -1] = np.inf
        got = score_maps(t(logit), t(emb), unc, pur, norm, t(gt), size=size, purity_size=psize, K=K, c=c, active=t(act), padding_mode=pad)
        for b in range(B):
            so, io, uo = ho.floating_region_score(logit[b:b + 1], emb[b:b + 1], unc, pur, norm, gt[b], size=size, purity_type=pur, K=K, c=c,
                                                  padding_mode=pad)
            so = so.copy(); so[act[b]] = -np.inf
            if not (same(got[0][b].cpu().numpy(), so) and same(got[1][b].cpu().numpy(), io) and same(got[2][b].cpu().numpy(), uo)):
                return "MISMATCH(maps, image %d)" % b, desc
        score_dev = got[0]
    # selection on the device's own score map (bit-equal to the oracle's by now)
    n = int(rng.integers(1, max(2, H * W // 20)))
    arad, mrad = int(rng.choice([0, 1, 1, 2])), int(rng.choice([0, 1, 3, 5, 5, 9, 16]))
    desc.update(n=n, arad=arad, mrad=mrad)
    method = str(rng.choice(["auto", "auto", "serial"]))
    a_d, s_d = t(act).clone(), torch.zeros((B, H, W), dtype=torch.bool, device=dev)
    am_d = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    sc = score_dev.clone()
    picks, npk = greedy_select(sc, n, arad, mrad, a_d, s_d, am_d, t(gt), method=method)
    for b in range(B):
        so = score_dev[b].cpu().numpy().copy()
        a_o, s_o, am_o = act[b].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
        _, _, _, _, p_o = ho.select_pixels_to_label(so, n, arad, mrad, a_o, s_o, am_o, gt[b], True)
        k = int(npk[b])
        if k != len(p_o) or not same(picks[b, :k].cpu().numpy(), p_o) or not np.array_equal(a_d[b].cpu().numpy(), a_o) \
                or not np.array_equal(s_d[b].cpu().numpy(), s_o) or not np.array_equal(am_d[b].cpu().numpy(), am_o) \
                or not same(sc[b].cpu().numpy(), so):
            return "MISMATCH(selection %s, image %d: %d vs %d picks)" % (method, b, k, len(p_o)), desc
    return "ok", desc


t0 = time.time()
counts = {}
for i in range(N):
    res, desc = one_case(i)
    counts[res.split("(")[0]] = counts.get(res.split("(")[0], 0) + 1
    if res.startswith("MISMATCH"):
        print(res, desc, flush=True)
        print("seed", SEED, "case", i)
        sys.exit(1)
    if i % 20 == 0:
        print("case %d %s %s" % (i, res, {k: desc[k] for k in ("O", "C", "H", "W", "B", "unc", "pur", "pad", "size", "lowres")}), flush=True)
print("fuzz_parity: %d cases, seed %d, %s, %.0f s: HIP == oracle bit for bit (maps, picks, masks)" % (N, SEED, counts, time.time() - t0))
