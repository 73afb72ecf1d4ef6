"""Child process of tests/test_sanitizers.py: the oracle's C (a -fsanitize=address,undefined build named by HALO_ORACLE_LIB) driven
through oracle/halo_oracle.py over the golden vectors -- every scoring branch, both selection rounds, the head ops, the resize, the
Gram twin, the padding modes and ragged / degenerate shapes -- with numpy only (the process runs under LD_PRELOAD=libasan).
The VALUES are checked by tests/test_oracle_golden.py in an ordinary process; here the memory accesses are."""
import glob
import os
import sys

import numpy as np

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
import oracle.halo_oracle as ho      # noqa: E402

assert os.environ.get("HALO_ORACLE_LIB")
COMBOS = {"halo": ("entropy", "radius"), "ripu": ("entropy", "ripu"), "hyper": ("entropy", "hyper"), "hyperK10": ("entropy", "hyper"),
          "none_radius": ("none", "radius"), "pixent_euc": ("pixel_entropy", "euc_norm"), "oracle": ("oracle_acc", "oracle_ripu"),
          "ent_none": ("entropy", "none"), "vestigial": ("hyperbolic", "ripu")}
n = 0
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "case_*.npz"))):
    d = np.load(f)
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    emb = ho.expmap(d["z"], 1.0, dim=1)
    lg = ho.hypermlr(emb, d["P_MLR"], d["A_MLR"], 1.0)
    ho.dist0(emb, 1.0, dim=1)
    ho.bilinear(lg.astype(np.float32), (H, W))
    ho.bilinear(emb, (H, W))
    ho.gram_radius(emb, (H, W))
    for tag in sorted({k.split("__")[0] for k in d.files if k.endswith("__score")}):
        unc, pur = COMBOS[tag]
        mrad, K, norm = (int(v) for v in d[tag + "__params"])
        s, i, u = ho.floating_region_score(d["logit"], d["embed"], unc, pur, bool(norm), d["gt"], size=3, purity_type=pur, K=K)
        act, sel, am = d["prior_active"].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
        for _ in range(2):
            sc = s.copy()
            sc[act] = -np.inf
            ho.select_pixels_to_label(sc, int(d["meta_n_regions"][0]), 1, mrad, act, sel, am, d["gt"])
        n += 1
# wider windows, every padding mode, ragged and degenerate shapes, many classes
rng = np.random.default_rng(1)
for (H, W, C, O) in [(1, 1, 1, 1), (1, 7, 2, 3), (9, 1, 3, 2), (33, 47, 6, 11), (16, 16, 4, 40)]:
    logit = rng.standard_normal((1, O, H, W)).astype(np.float32)
    emb = rng.standard_normal((1, C, H, W)) * 0.2
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    gt[rng.random((H, W)) < 0.1] = 255
    for size in (1, 3, 5):
        for mode in ("zeros", "reflect", "replicate", "circular"):
            # torch refuses a reflect pad >= the dimension and a circular pad > it (the product's host checks the same); the 'hyper'
            # purity window is 3 x 3 whatever `size` is (floating_region.py:54-55)
            if mode in ("reflect", "circular") and (max(size, 3) // 2 >= min(H, W)):
                continue
            for unc, pur, K in (("entropy", "radius", 100), ("entropy", "ripu", 100), ("oracle_acc", "hyper", 300), ("entropy", "hyper", 5000)):
                ho.floating_region_score(logit, emb, unc, pur, True, gt, size=size, purity_type=pur, K=K, padding_mode=mode)
                ho.floating_region_score(logit, emb.astype(np.float32), unc, pur, False, gt, size=size, purity_type=pur, K=K, padding_mode=mode)
                n += 2
    sc = rng.standard_normal((H, W))
    sc[rng.random((H, W)) < 0.1] = np.nan
    act, sel, am = np.zeros((H, W), bool), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
    ho.select_pixels_to_label(sc, H * W + 3, 2, 7, act, sel, am, gt)             # runs to exhaustion, windows larger than the image
x = rng.standard_normal((5, 7))
ho.logmap(ho.expmap(x.astype(np.float32)), 1.0)
ho.dist(ho.expmap(x.astype(np.float32)), ho.expmap(x[::-1].astype(np.float32)))
ho.sum_dim0(rng.standard_normal((5000, 3, 5)).astype(np.float32))
ho.logf(np.array([0.0, -1.0, np.inf, np.nan, 1e-45, 1e-38, 1.0, 3e38], np.float32))
ho.expf(np.array([0.0, -1e30, 1e30, np.inf, -np.inf, np.nan, -103.9, 88.8], np.float32))
print("driver ok: %d scoring calls" % n)
