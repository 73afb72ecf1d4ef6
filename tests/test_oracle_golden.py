"""The CPU oracle against vectors produced by the reference's own code (CPU, no GPU).

Bars (BASELINE.json north_star): float maps within 1e-4 (observed <= 2e-6), selected
pixel indices / masks bit-exact, over every (uncertainty, purity) branch of
FloatingRegionScore.forward, two selection rounds each.
"""
import types

import os

import numpy as np
import pytest

from conftest import COMBOS, GOLDEN, all_case_combos, case_files, max_abs_diff
from oracle import halo_oracle as ho

TOL = 1e-4          # the contract
TIGHT = 5e-6        # what the restatement actually achieves on these vectors


@pytest.mark.parametrize("case", [f.split("/")[-1][:-4] for f in case_files()])
def test_head_pieces(golden, case):
    d = golden(case)
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    e = ho.expmap(d["z"], 1.0, dim=1)                      # classifier.py:553
    assert max_abs_diff(e, d["embed_lr"]) < 1e-14
    lg = ho.hypermlr(d["embed_lr"], d["P_MLR"], d["A_MLR"])  # classifier.py:554
    assert max_abs_diff(lg, d["logit_lr64"]) < 1e-11
    assert np.abs(lg.astype(np.float32) - d["logit_lr"]).max() < 1e-5
    r = ho.dist0(d["embed_lr"], 1.0, dim=1)
    assert max_abs_diff(r, d["radius_lr"]) < 1e-13
    up = ho.bilinear(d["logit_lr"], (H, W))                # build.py:123-125
    assert max_abs_diff(up, d["logit"]) < 4e-6
    if d["embed"].dtype == np.float64:
        upe = ho.bilinear(d["embed_lr"], (H, W))           # build.py:133-135
        assert max_abs_diff(upe, d["embed"]) < 1e-15


@pytest.mark.parametrize("case,tag", all_case_combos())
def test_score_and_selection(golden, case, tag):
    d = golden(case)
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    n = int(d["meta_n_regions"][0])
    unc, pur = COMBOS[tag]
    mrad, K, norm = (int(v) for v in d[tag + "__params"])
    s, i, u = ho.floating_region_score(d["logit"], d["embed"], unc, pur, bool(norm), d["gt"],
                                       size=3, purity_type=pur, K=K)
    assert s.dtype == d[tag + "__score"].dtype and i.dtype == d[tag + "__impurity"].dtype
    # 'hyper' (the default purity, defaults.py:69) included: quantised-radius bins could flip on a 1-ulp
    # difference of the normalised radius, but on the committed vectors NO bin flips -- every map and every
    # mask below is compared for every branch, with no escape hatch
    assert max_abs_diff(i, d[tag + "__impurity"]) < TIGHT
    assert max_abs_diff(s, d[tag + "__score"]) < TIGHT
    assert max_abs_diff(u, d[tag + "__uncertainty"]) < TIGHT

    act = d["prior_active"].copy()
    sel = np.zeros((H, W), bool)
    am = np.full((H, W), 255, np.int64)
    for rnd in ("r1", "r2"):
        sc = s.copy()
        sc[act] = -np.inf                                   # build.py:146
        _, _, _, _, picks = ho.select_pixels_to_label(sc, n, 1, mrad, act, sel, am, d["gt"], True)
        ref = d[f"{tag}__{rnd}_picks"]
        assert len(picks) == len(ref)
        assert np.array_equal(picks[:, :2], ref[:, :2]), "selected pixel order differs"
        if len(ref):
            assert max_abs_diff(picks[:, 2], ref[:, 2]) < TIGHT
        assert np.array_equal(act, d[f"{tag}__{rnd}_active"])
        assert np.array_equal(sel, d[f"{tag}__{rnd}_selected"])
        assert np.array_equal(am, d[f"{tag}__{rnd}_active_mask"])
        if rnd == "r1":
            assert max_abs_diff(sc, d[f"{tag}__r1_score"]) < TIGHT


@pytest.mark.parametrize("case,tag", all_case_combos())
def test_selection_on_reference_score(golden, case, tag):
    """Selector alone: fed the reference's own score map it must reproduce the reference's
    masks exactly (integer work, no float slack)."""
    d = golden(case)
    H, W, _, _ = (int(v) for v in d["meta_HWCO"])
    n = int(d["meta_n_regions"][0])
    mrad = int(d[tag + "__params"][0])
    act = d["prior_active"].copy()
    sel = np.zeros((H, W), bool)
    am = np.full((H, W), 255, np.int64)
    for rnd in ("r1", "r2"):
        sc = d[tag + "__score"].copy()
        sc[act] = -np.inf
        _, _, _, _, picks = ho.select_pixels_to_label(sc, n, 1, mrad, act, sel, am, d["gt"], True)
        ref = d[f"{tag}__{rnd}_picks"]
        assert np.array_equal(picks[:, :2], ref[:, :2])
        a = picks[:, 2]
        assert np.array_equal(np.isnan(a), np.isnan(ref[:, 2]))
        assert np.array_equal(a[~np.isnan(a)], ref[:, 2][~np.isnan(a)])
        assert np.array_equal(act, d[f"{tag}__{rnd}_active"])
        assert np.array_equal(sel, d[f"{tag}__{rnd}_selected"])
        assert np.array_equal(am, d[f"{tag}__{rnd}_active_mask"])


def test_hypermapper_lastdim(golden):
    d = golden("hypermapper")
    for c in (1.0, 0.5):
        t = f"c{c}"
        assert max_abs_diff(ho.expmap(d[t + "__x"], c), d[t + "__expmap"]) < 1e-14
        assert max_abs_diff(ho.logmap(d[t + "__expmap"], c), d[t + "__logmap"]) < 1e-12
        assert max_abs_diff(ho.dist0(d[t + "__expmap"], c), d[t + "__dist0"]) < 1e-12
        # near the ball boundary artanh amplifies rounding by 1/(1-z^2) ~ 1e7
        assert max_abs_diff(ho.dist(d[t + "__expmap"], d[t + "__y_h"], c), d[t + "__dist"]) < 1e-8
        # float32 points (round 4: artanh's logs in the input dtype, as geoopt's stereographic/math.py): the rounding of 1 +- z
        # to float32 is shared with the reference; what differs is torch's float32 log against the oracle's recipe (<= 1 ulp of
        # each log value) and, at the clamp, 1 / (1 - z^2) ~ 4e6 times one ulp of the float32 norm
        got, want = ho.dist0(d[t + "__x_f32"], c), d[t + "__dist0_f32"]
        assert got.dtype == np.float32 and want.dtype == np.float32
        z = np.minimum(np.linalg.norm(d[t + "__x_f32"].astype(np.float64), axis=1) * np.sqrt(c), 1 - 2.0 ** -23)
        tol = 4e-7 + 3 * 6e-8 / (1 - z * z) / np.sqrt(c)
        assert np.all(np.abs(got.astype(np.float64) - want) <= tol), float(np.abs(got.astype(np.float64) - want).max())


@pytest.mark.parametrize("lowres_mode", ["exact", "gram"])
def test_region_selection_driver(golden, lowres_mode):
    """RegionSelection (build.py:71-186), two rounds over a 3-image pool, including the
    budget formula, the prior-pick masking and the uint8 mask the reference saved as PNG.
    'gram': the radius through the Gram form (the product's default low-res mode for float64 embeddings) must
    reproduce the reference's files as well."""
    d = golden("region_selection")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1,
                                     MASK_RADIUS_K=5, BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100))
    state = [dict(active=np.zeros((H, W), bool), selected=np.zeros((H, W), bool),
                  origin_mask=np.full((H, W), 255, np.int64)) for _ in range(3)]
    for rnd in (1, 2):
        imgs = [dict(logit_lr=d[f"img{i}__logit_lr"], embed_lr=d[f"img{i}__embed_lr"],
                     origin_label=d[f"img{i}__gt"], **state[i]) for i in range(3)]
        res = ho.region_selection(cfg, imgs, lowres_mode=lowres_mode)
        for i, (mask, act, sel, picks) in enumerate(res):
            assert np.array_equal(mask, d[f"r{rnd}_img{i}__mask_png"])
            assert np.array_equal(act, d[f"r{rnd}_img{i}__active"])
            assert np.array_equal(sel, d[f"r{rnd}_img{i}__selected"])
            state[i] = dict(active=act, selected=sel, origin_mask=mask.astype(np.int64))


def test_helper_methods(golden):
    """compute_pixel_entropy / compute_region_uncertainty / quantize_uncert_map /
    compute_region_impurity (floating_region.py:70-127) against the reference's direct outputs."""
    d = golden("helpers")
    assert max_abs_diff(ho.softmax(d["logit"]), d["p"]) < 1e-7
    for key, (unc, size, box) in {"pixel_entropy": ("pixel_entropy", 3, False), "ru_entropy_k3": ("entropy", 3, True),
                                  "ru_entropy_k5": ("entropy", 5, True), "ru_oracle_acc": ("oracle_acc", 3, True),
                                  "ru_none": ("none", 3, False), "ru_hyperbolic": ("hyperbolic", 3, True)}.items():
        o = ho.uncertainty_from_probs(d["p"], unc, d["gt"], size, box)
        assert o.shape == d[key].shape and max_abs_diff(o, d[key]) < 5e-6, key
    assert np.array_equal(ho.quantize_uncert_map(d["embed"], 100), d["quantized"])
    i, c = ho.region_impurity(d["quantized"], 100, 3)
    assert max_abs_diff(i, d["imp_hyper"]) < 1e-6 and np.array_equal(c, d["cnt_hyper"])
    i, c = ho.region_impurity(d["argmax"], 19, 5)
    assert max_abs_diff(i, d["imp_ripu_k5"]) < 1e-6 and np.array_equal(c, d["cnt_ripu_k5"])


def test_gram_radius_tracks_upsample_then_reduce():
    """The Gram form of the low-res radius (oracle twin of k_gram_lr + k_radius_gram) against the reference order
    (bilinear upsample, then dist0 / norm over the channels): the same number rounded differently.  Smooth embeddings,
    projected (boundary) vectors, zero vectors, clamped edge taps, non-integer magnifications."""
    rng = np.random.default_rng(11)
    for (C, h, w, H, W) in ((8, 8, 16, 32, 64), (24, 9, 7, 40, 45), (64, 12, 20, 77, 128), (5, 1, 6, 4, 24), (3, 4, 4, 4, 4)):
        z = (rng.standard_normal((1, C, h, w)) * 0.1).astype(np.float32)
        z[0, :, 0, 0] = 0.0
        if h > 2 and w > 3:
            z[0, :, 1:3, 1:4] *= 300.0                   # tanh clamp + projection: vectors on the ball's boundary
        emb = ho.expmap(z, 1.0, dim=1)
        up = ho.bilinear(emb, (H, W))
        for mode, ref in (("radius", ho.dist0(up, 1.0, dim=1)[0]), ("euc_norm", np.sqrt((up[0] ** 2).sum(0)))):
            g = ho.gram_radius(emb, (H, W), mode, 1.0)
            assert g.shape == (H, W)
            tol = 1e-12 if mode == "radius" else 1e-13
            assert np.nanmax(np.abs(g - ref) / np.maximum(1.0, np.abs(ref))) < tol, (C, h, w, H, W, mode)
    # as the impurity of the scorer: same maps to ~1e-15, same picks
    logit = rng.standard_normal((1, 19, 16, 32)).astype(np.float32)
    emb = ho.expmap((rng.standard_normal((1, 16, 16, 32)) * 0.1).astype(np.float32), 1.0, dim=1)
    up, lg = ho.bilinear(emb, (64, 128)), ho.bilinear(logit, (64, 128))
    a = ho.floating_region_score(lg, up, "entropy", "radius", True, None, size=3, purity_type="radius")
    b = ho.floating_region_score(lg, None, "entropy", "radius", True, None, size=3, purity_type="radius",
                                 impurity_raw=ho.gram_radius(emb, (64, 128)))
    assert a[0].dtype == b[0].dtype == np.float64 and np.abs(a[0] - b[0]).max() < 1e-13 and np.array_equal(a[2], b[2])
    for pur in ("hyper",):
        a = ho.floating_region_score(lg, up, "entropy", pur, True, None, size=3, purity_type=pur, K=20)
        b = ho.floating_region_score(lg, None, "entropy", pur, True, None, size=3, purity_type=pur, K=20,
                                     impurity_raw=ho.gram_radius(emb, (64, 128)))
        assert np.mean(a[1] != b[1]) < 0.01            # a radius on a bin edge may land in the neighbouring bin


@pytest.mark.parametrize("shape", [(16, 12, 20, 48, 80), (256, 12, 20, 45, 77), (7, 9, 9, 64, 64), (64, 16, 32, 64, 128)])
def test_gram_radius_is_guarded_against_cancellation(shape):
    """neighbouring low-res vectors v and -v (1 - eps), eps 1e-1 .. 1e-12, and opposing vectors on the ball's
    boundary.  Unguarded, the 10-term Gram form loses the squared norm of the interpolated vector to ~1e-16 max||v||^2 (a norm
    error up to 1e-8 where the true norm is near zero).  Guarded (a pixel whose terms cancel below 2^-10 of their magnitudes is
    evaluated in the exact order, oracle/halo_oracle.c:halo_o_gram_radius = k_radius_gram): against the EXACT order
    (ho.bilinear, then the norm) the NORM tanh(r / 2) agrees to 6.5e-11 relative by construction (squared norm: 4 C u 2^10 =
    1.2e-10 at C = 256) -- asserted here at 1e-11, observed 2e-13 -- and the selection made from either map is the same."""
    from conftest import opposing_neighbours_embedding
    C, h, w, H, W = shape
    emb = opposing_neighbours_embedding(C, h, w, seed=C)
    up = ho.bilinear(emb, (H, W))
    r_exact = ho.dist0(up, 1.0, dim=1)[0]
    r_gram = ho.gram_radius(emb[0], (H, W), "radius", 1.0)
    n_exact, n_gram = np.tanh(r_exact / 2), np.tanh(r_gram / 2)
    assert n_exact.min() < 1e-3 and n_exact.max() > 0.999                  # near-vanishing AND boundary pixels are present
    rel = np.abs(n_gram - n_exact) / np.maximum(n_exact, 1e-300)
    assert rel.max() <= 1e-11, rel.max()
    assert (r_gram == r_exact).mean() > 0.3                                # the guarded pixels ARE the exact order, bit for bit
    e_exact = np.sqrt((up[0] ** 2).sum(0))
    e_gram = ho.gram_radius(emb[0], (H, W), "euc_norm", 1.0)
    assert np.max(np.abs(e_gram - e_exact) / np.maximum(e_exact, 1e-300)) <= 1e-11
    # the driver on such an image: same files / picks in both modes
    rng = np.random.default_rng(5)
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1,
                                     MASK_RADIUS_K=5, BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100))
    im = dict(logit_lr=rng.standard_normal((1, 19, h, w)).astype(np.float32), embed_lr=emb,
              origin_label=rng.integers(0, 19, (H, W)).astype(np.int64), origin_mask=np.full((H, W), 255, np.int64),
              active=np.zeros((H, W), bool), selected=np.zeros((H, W), bool))
    (m_e, a_e, s_e, p_e), = ho.region_selection(cfg, [dict(im)], lowres_mode="exact")
    (m_g, a_g, s_g, p_g), = ho.region_selection(cfg, [dict(im)], lowres_mode="gram")
    assert len(p_e) > 0 and np.array_equal(p_e[:, :2], p_g[:, :2]) and np.array_equal(m_e, m_g) and np.array_equal(a_e, a_g)
    assert np.max(np.abs(p_e[:, 2] - p_g[:, 2])) <= 1e-10


@pytest.mark.parametrize("mode", ["reflect", "replicate", "circular"])
def test_padding_modes_vs_reference(golden, mode):
    """FloatingRegionScore(padding_mode=...) -- forwarded to the two nn.Conv2d box filters in the reference
    (floating_region.py:49,63), default 'zeros' in every caller of its tree -- against the reference's own outputs
    (tests/golden/padding.npz: the reference's class run with each mode): maps within 5e-6, the quantised / arg-max
    branches' counts exactly k*k everywhere (padded taps are image pixels), first-round picks identical."""
    d = golden("padding")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    for tag, unc, pur in (("halo", "entropy", "radius"), ("ripu5", "entropy", "ripu"), ("hyperK10", "entropy", "hyper"),
                          ("oracle", "oracle_acc", "oracle_ripu")):
        key = f"{mode}__{tag}"
        size, K, norm = (int(v) for v in d[key + "__params"])
        s, i, u = ho.floating_region_score(d["logit"], d["embed"], unc, pur, bool(norm), d["gt"], size=size, purity_type=pur, K=K,
                                           padding_mode=mode)
        assert s.dtype == d[key + "__score"].dtype
        assert max_abs_diff(u, d[key + "__uncertainty"]) < 5e-6 and max_abs_diff(i, d[key + "__impurity"]) < 5e-6, key
        assert max_abs_diff(s, d[key + "__score"]) < 5e-6, key
        act = d["prior_active"].copy(); sel = np.zeros((H, W), bool); am = np.full((H, W), 255, np.int64)
        s[act] = -np.inf
        _, _, _, _, picks = ho.select_pixels_to_label(s, 12, 1, 3, act, sel, am, d["gt"], True)
        assert np.array_equal(picks[:, :2], d[key + "__picks"][:, :2]), key
        assert np.array_equal(am, d[key + "__active_mask"]), key
    p = ho.softmax(d["logit"][0])
    assert max_abs_diff(ho.uncertainty_from_probs(p, "entropy", None, 5, True, padding_mode=mode), d[f"{mode}__region_unc_k5"]) < 2.5e-5    # 25-tap float32 sums up to ~20: a few ulps of 16 (1.9e-6)
    imp, cnt = ho.region_impurity(p.argmax(0), O, 5, padding_mode=mode)
    assert max_abs_diff(imp, d[f"{mode}__imp_k5"]) < 1e-6 and np.array_equal(cnt, d[f"{mode}__cnt_k5"]) and float(cnt.min()) == 25.0


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("dtype,shape", [("float64", (1, 6, 160, 320, 1024, 2048)), ("float64", (2, 5, 23, 37, 147, 231)),
                                         ("float64", (1, 64, 40, 80, 256, 512)), ("float32", (1, 19, 640, 1280, 1024, 2048)),
                                         ("float32", (2, 19, 37, 53, 101, 203)), ("float64", (1, 3, 8, 8, 1, 1)),
                                         ("float32", (1, 3, 1, 1, 7, 5)), ("float64", (1, 3, 7, 9, 7, 9))])
def test_bilinear_is_torchs_cpu_kernel_bit_for_bit(dtype, shape, threads):
    """The contract's bilinear (columns first, rows second, each p*q + r*s as fma(p, q, r*s); round 4) against the call the
    reference makes, F.interpolate(mode='bilinear', align_corners=True) (build.py:123-135), on this host's CPU: the same bits at
    the shapes the path runs (low-res head outputs -> label size) and the fixture shapes.  Needs an FMA host (ATen's
    vectorised kernel contracts; the oracle is built with -mfma when /proc/cpuinfo has it)."""
    import torch
    import torch.nn.functional as F
    if " fma" not in open("/proc/cpuinfo").read():
        pytest.skip("no FMA on this host: ATen's kernel rounds every product")
    B, C, h, w, H, W = shape
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        x = torch.randn((B, C, h, w), dtype=getattr(torch, dtype), generator=torch.Generator().manual_seed(3))
        want = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True).numpy()
    finally:
        torch.set_num_threads(old)
    got = ho.bilinear(x.numpy(), (H, W))
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_mid_size_fixture_above_atens_onednn_switch_is_matched_bitwise_up_to_the_logarithm():
    """tests/golden/mid_112x192_c8_o19.npz (21504 pixels: above the 20480 at which ATen hands the 3 x 3 box convolution to oneDNN, like
    every production size; the cases A-D sit below, on the im2col + MKL sgemm path whose summation order is MKL's): the oracle's x4
    resize is torch's bit for bit (digest), the window-histogram impurity of `ripu` / `hyper` is the reference's bit for bit, and the
    box-summed entropy differs from the reference's in at most a handful of pixels by one ulp -- the closed-source logarithm
    (oracle/halo_oracle_math.h).  Picks and scores as everywhere."""
    import hashlib
    d = np.load(os.path.join(GOLDEN, "mid_112x192_c8_o19.npz"))
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    assert H * W > 20480
    logit, embed = ho.bilinear(d["logit_lr"], (H, W)), ho.bilinear(d["embed_lr"], (H, W))
    hsh = hashlib.sha256()
    hsh.update(logit.tobytes())
    hsh.update(embed.tobytes())
    assert np.array_equal(np.frombuffer(hsh.digest(), np.uint8), d["resized_digest"]), "the oracle's resize is not torch's bit for bit"
    for tag, (unc, pur) in {"halo": ("entropy", "radius"), "ripu": ("entropy", "ripu"), "hyper": ("entropy", "hyper")}.items():
        mrad, K, norm = (int(v) for v in d[tag + "__params"])
        s, i, u = ho.floating_region_score(logit, embed, unc, pur, bool(norm), d["gt"], size=3, purity_type=pur, K=K)
        nd = int((u != d[tag + "__uncertainty"]).sum())
        assert nd <= 8 and np.abs(u - d[tag + "__uncertainty"]).max() <= 3e-7, (tag, nd)
        if pur != "radius":
            assert np.array_equal(i, d[tag + "__impurity"]), tag
            assert int((s != d[tag + "__score"]).sum()) <= 8, tag
        assert np.abs(s.astype(np.float64) - d[tag + "__score"].astype(np.float64)).max() < 1e-6, tag
        act, sel, am = d["prior_active"].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
        sc = s.copy()
        sc[act] = -np.inf
        _, _, _, _, picks = ho.select_pixels_to_label(sc, 40, 1, mrad, act, sel, am, d["gt"], return_picks=True)
        assert np.array_equal(picks[:, :2], d[tag + "__picks"][:, :2]), tag
