"""Numpy model of the value-binned selector (halo_amd/csrc/halo_select_binned.hip) -- TEST INFRASTRUCTURE.

It restates the kernels' host-visible logic step by step (range -> coarse histogram -> threshold and
sub-bin layout -> fine bins (BIN_CAP slots each) -> sweep over the bins with a pick grid -> bail conditions) so that the
ALGORITHM can be checked against the CPU oracle without a GPU: same candidate bound, same bin
arithmetic (IEEE double, truncation), same bail rules.  The product never imports this file.
"""
import numpy as np

NB1 = 2048
SW_SURV = 256
PL_KEYS = 4       # distinct keys of a full bin the sweep walks before it hands over
BIN_CAP = 128     # slots per fine bin (k_sel_place): a fuller bin is a plateau of ties -- walked in position order if its candidates
                  # all carry one key, otherwise the sweep stops in front of it and hands the image over from there


def order_key(v):
    """uint64 keys: value order, -0 == +0, NaN on top (halo_select_common.hpp)."""
    v = np.asarray(v, dtype=np.float64).copy()
    nan = np.isnan(v)
    v[v == 0.0] = 0.0
    u = v.view(np.uint64)
    neg = (u >> np.uint64(63)).astype(bool)
    k = np.where(neg, ~u, u | np.uint64(1 << 63))
    k[nan] = np.uint64(0xffffffffffffffff)
    return k


KEY_NEG_INF = np.uint64(0x000fffffffffffff)
KEY_POS_INF = np.uint64(0xfff0000000000000)


def _hit(grid, y, x, cs, mrad):
    cy, cx = y // cs, x // cs
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            p = grid.get((cy + a, cx + b))
            if p is not None and abs(p[0] - y) <= mrad and abs(p[1] - x) <= mrad:
                return True
    return False


def binned_select(score, n_regions, mrad, target=64, captot=None):
    """-> (status 'done'|'bail', picks [(h, w)], stats).  `score` (H,W) float32|float64 is not modified."""
    H, W = score.shape
    n = min(int(n_regions), H * W)
    v = score.astype(np.float64)
    key = order_key(v)
    bad = key >= KEY_POS_INF
    ok = ~bad & (key != KEY_NEG_INF)
    stats = {"ncand": 0, "bins": 0, "reason": ""}
    if n == 0:
        return "done", [], stats
    win = (2 * mrad + 1) ** 2
    kneed = min(win * n, H * W)
    if captot is None:
        captot = min(max(2 * kneed, 65536), H * W)
    nvalid = int(ok.sum())
    rng_ok = (not bad.any()) and nvalid > 0
    if rng_ok:
        lo, hi = v[ok].min(), v[ok].max()
        with np.errstate(all="ignore"):
            scale = np.float64(NB1) / (hi - lo)
        rng_ok = bool(hi > lo and scale > 0.0 and scale < 1.0e300)
    if not rng_ok:
        stats["reason"] = "range"
        return "bail", [], stats
    t = (v - lo) * scale
    with np.errstate(invalid="ignore"):
        j = np.minimum(np.where(ok, t, 0.0).astype(np.int64), NB1 - 1)
    c = np.bincount(j[ok], minlength=NB1)
    m = (c + target - 1) // target
    S = np.cumsum(c[::-1])[::-1]                     # values in bins >= j
    M = np.concatenate([np.cumsum(m[::-1])[::-1][1:], [0]])   # fine bins above bin j
    ge = np.nonzero(S >= kneed)[0]
    t1 = int(ge.max()) if len(ge) else 0
    truncated = False
    if S[t1] > captot:
        t1 += 1
        truncated = True
    cand = ok & (j >= t1)
    ys, xs = np.nonzero(cand)
    jj = j[cand]
    mj = m[jj]
    s = np.minimum(((t[cand] - jj) * mj).astype(np.int64), mj - 1)
    f = M[jj] + (mj - 1 - s)
    kk = key[cand]
    pos = (xs.astype(np.int64) << 16) | ys
    stats["ncand"] = int(cand.sum())
    full = set(np.nonzero(np.bincount(f) > BIN_CAP)[0].tolist()) if len(f) else set()
    order = np.argsort(f, kind="stable")
    f, kk, pos, ys, xs = f[order], kk[order], pos[order], ys[order], xs[order]
    # sweep
    cs = mrad + 1
    grid = {}
    picks = []
    i = 0
    N = len(f)
    while i < N:
        e = i
        while e < N and f[e] == f[i]:
            e += 1
        if int(f[i]) in full:
            # a bin that ran out of slots: its distinct keys in descending order, each one's candidates in POSITION order (smallest
            # w, then smallest h: ties) -- the kernel walks the map's columns once per key, here the bin's candidates are sorted; more
            # than PL_KEYS distinct keys (dense near-ties) hand the image over from where the walk stands
            stats["plateaus"] = stats.get("plateaus", 0) + 1
            keys_desc = sorted(set(int(q) for q in kk[i:e]), reverse=True)
            for npass, kval in enumerate(keys_desc):
                if npass == PL_KEYS:
                    stats["reason"] = "overflow"
                    return "bail", picks, stats
                by_pos = sorted((q for q in range(i, e) if int(kk[q]) == kval), key=lambda q: int(pos[q]))
                for c0 in range(0, len(by_pos), SW_SURV):
                    piece = by_pos[c0:c0 + SW_SURV]
                    alive = [q for q in piece if not _hit(grid, int(ys[q]), int(xs[q]), cs, mrad)]
                    while alive:
                        best = min(alive, key=lambda q: int(pos[q]))
                        y, x = int(ys[best]), int(xs[best])
                        grid[(y // cs, x // cs)] = (y, x)
                        picks.append((y, x))
                        if len(picks) >= n:
                            return "done", picks, stats
                        alive = [q for q in alive if not (abs(int(ys[q]) - y) <= mrad and abs(int(xs[q]) - x) <= mrad)]
            i = e
            continue
        stats["bins"] += 1
        surv = []
        for q in range(i, e):
            y, x = int(ys[q]), int(xs[q])
            cy, cx = y // cs, x // cs
            hit = False
            for a in (-1, 0, 1):
                for b in (-1, 0, 1):
                    p = grid.get((cy + a, cx + b))
                    if p is not None and abs(p[0] - y) <= mrad and abs(p[1] - x) <= mrad:
                        hit = True
            if not hit:
                surv.append(q)
        if len(surv) > SW_SURV:
            stats["reason"] = "survivors"
            return "bail", picks, stats
        alive = list(surv)
        while alive:
            best = max(alive, key=lambda q: (int(kk[q]), -int(pos[q])))
            y, x = int(ys[best]), int(xs[best])
            assert (y // cs, x // cs) not in grid
            grid[(y // cs, x // cs)] = (y, x)
            picks.append((y, x))
            if len(picks) >= n:
                return "done", picks, stats
            alive = [q for q in alive if not (abs(int(ys[q]) - y) <= mrad and abs(int(xs[q]) - x) <= mrad)]
        i = e
    if truncated:
        stats["reason"] = "exhausted"
        return "bail", picks, stats
    return "done", picks, stats
