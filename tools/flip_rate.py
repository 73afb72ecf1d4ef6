#!/usr/bin/env python3
"""Mask-exact rate of the CPU oracle against the REFERENCE at the headline size.

Build container only: imports /root/reference through tests/golden/make_fixtures.import_reference() and runs its
FloatingRegionScore.forward + select_pixels_to_label on tests/fullsize_inputs.build(seed, ...) -- then the oracle on
the same arrays -- and reports, per image: masks equal?, picks that differ, score max |d|, and how many pixels of
the three maps differ in their BITS.

    python tools/flip_rate.py --branch hyper --seeds 0-49 [--channels 64] [--mods peaked] > profiles/r06_flip_rate_hyper.txt

The reference's pick table is read off its own score map by the oracle's selector and CHECKED against the masks the
reference's select_pixels_to_label left behind (the reference keeps no table, core/active/build.py:37-62).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import fullsize_inputs as fi          # noqa: E402
import oracle.halo_oracle as ho       # noqa: E402

_REF = None


def reference():
    global _REF
    if _REF is None:
        from make_fixtures import import_reference
        cfg, hyp, fr, ab = import_reference()
        cfg.MODEL.CURVATURE = 1.0
        _REF = (cfg, hyp, fr, ab)
    return _REF


def run_reference(inp, branch, O=19, n=None):
    """-> dict(score, impurity, uncertainty, active, selected, active_mask, picks): the reference's own outputs
    (tests/golden/make_fixtures.py:run_reference_fullsize, the function that produced tests/golden/fullsize_picks.npz)."""
    from make_fixtures import run_reference_fullsize
    cfg, hyp, fr, ab = reference()
    cfg.MODEL.NUM_CLASSES = O
    return run_reference_fullsize(fr, ab, inp, branch, O=O, n=n)


def run_oracle(inp, branch, O=19, n=None):
    unc, pur, norm, mrad, K = fi.BRANCHES[branch]
    H, W = inp["gt"].shape
    n = fi.n_regions(H, W) if n is None else n
    score, imp, uncm = ho.floating_region_score(inp["logit"], decoder_out=inp["embed"], unc_type=unc, pur_type=pur,
                                                normalize=norm, ground_truth=inp["gt"], size=3, purity_type=pur, K=K)
    a, s, m = inp["prior"].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
    score[a] = -np.inf
    s0 = score.copy()
    _, _, _, _, picks = ho.select_pixels_to_label(score, n, 1, mrad, a, s, m, inp["gt"], return_picks=True)
    return dict(score=s0, impurity=imp, uncertainty=uncm, active=a, selected=s, active_mask=m, picks=picks)


def bits_differ(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return int((~((a == b) | ((a != a) & (b != b)))).sum())


def compare(ref, ora):
    fin = np.isfinite(ref["score"]) & np.isfinite(ora["score"])
    dmax = float(np.abs(ref["score"][fin].astype(np.float64) - ora["score"][fin].astype(np.float64)).max()) if fin.any() else 0.0
    n = min(len(ref["picks"]), len(ora["picks"]))
    pd = int((ref["picks"][:n, :2] != ora["picks"][:n, :2]).any(axis=1).sum()) + abs(len(ref["picks"]) - len(ora["picks"]))
    first = int(np.argmax((ref["picks"][:n, :2] != ora["picks"][:n, :2]).any(axis=1))) if pd else -1
    return dict(masks_equal=bool(np.array_equal(ref["active_mask"], ora["active_mask"]) and np.array_equal(ref["selected"], ora["selected"])
                                 and np.array_equal(ref["active"], ora["active"])),
                picks_differ=pd, first_diff=first, selected_px_differ=int((ref["selected"] != ora["selected"]).sum()),
                score_max_abs=dmax, score_bits=bits_differ(ref["score"], ora["score"]),
                imp_bits=bits_differ(ref["impurity"], ora["impurity"]), unc_bits=bits_differ(ref["uncertainty"], ora["uncertainty"]))


def parse_seeds(s):
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--branch", default="hyper", choices=sorted(fi.BRANCHES))
    ap.add_argument("--seeds", default="0-12")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--mods", default="")
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    mods = tuple(m for m in a.mods.split("+") if m)
    print(f"# oracle vs reference (torch {torch.__version__}, ATen {torch.backends.cpu.get_cpu_capability()}, "
          f"MKL_ENABLE_INSTRUCTIONS={os.environ.get('MKL_ENABLE_INSTRUCTIONS', 'unset: the best the CPU has')}), branch {a.branch} "
          f"{fi.BRANCHES[a.branch]}, {a.height}x{a.width}, C={a.channels}, mods={mods or '-'}, f32 embedding={a.f32}", flush=True)
    ok = tot = 0
    for seed in parse_seeds(a.seeds):
        t0 = time.time()
        inp = fi.build(seed, C=a.channels, H=a.height, W=a.width, mods=mods, f32_embed=a.f32)
        r = compare(run_reference(inp, a.branch), run_oracle(inp, a.branch))
        tot += 1
        ok += r["masks_equal"]
        print(f"seed {seed:3d}  masks_equal {r['masks_equal']!s:5}  picks_differ {r['picks_differ']:4d} (first {r['first_diff']:4d})  "
              f"selected_px_differ {r['selected_px_differ']:3d}  score max|d| {r['score_max_abs']:.2e}  bits differ: score {r['score_bits']:7d} "
              f"impurity {r['imp_bits']:7d} uncertainty {r['unc_bits']:7d}   [{time.time() - t0:.0f} s]", flush=True)
    print(f"# masks equal: {ok} / {tot}", flush=True)


if __name__ == "__main__":
    main()
