"""Parity of the HIP path (called through the Python host -> C ABI) on a real MI355X.

Three anchors:
  * golden vectors produced by the reference's own code (tests/golden/*.npz): float maps within
    1e-4 (we assert 5e-6), selected indices and masks bit-exact;
  * the CPU oracle on the same inputs: BIT-EXACT maps and picks (both sides implement the same
    numeric contract), at fixture sizes, at BASELINE config 1 (256x512, C=64) and at the full
    1024x2048, C=256 size;
  * size-independent properties of the selection at full size.
"""
import math
import os
import tempfile
import types

import numpy as np
import pytest
import torch

from conftest import COMBOS, GOLDEN, all_case_combos, case_files, max_abs_diff

pytestmark = pytest.mark.gpu

TOL = 5e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from halo_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


class _env_set:
    """context manager: os.environ entries set for the block, restored after"""
    def __init__(self, env):
        self.env, self.old = env, {}

    def __enter__(self):
        import os
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _lr_mode():
    """the low-res mode RegionSelection / score_maps_lowres run in by default ('exact' unless HALO_LOWRES says otherwise):
    the oracle driver is asked for the same evaluation order, so the comparisons stay bit-exact"""
    from halo_amd.core.active.floating_region import lowres_mode
    return lowres_mode(None)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a == b) | both_nan))


# ------------------------------------------------------------------ head pieces
@pytest.mark.parametrize("case", [f.split("/")[-1][:-4] for f in case_files()])
def test_head_pieces_vs_reference_and_oracle(golden, case, dev):
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners
    from oracle import halo_oracle as ho
    d = golden(case)
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    mapper = HyperMapper(c=1.0)
    with torch.no_grad():
        e = mapper.expmap(t(d["z"], dev), dim=1)
        assert e.dtype == torch.float64
        assert max_abs_diff(e.cpu().numpy(), d["embed_lr"]) < 1e-14
        mlr = HyperMLR(C, O, c=1.0).to(dev)
        mlr.P_MLR.copy_(t(d["P_MLR"], dev))
        mlr.A_MLR.copy_(t(d["A_MLR"], dev))
        lg = mlr(t(d["embed_lr"], dev))
        assert lg.dtype == torch.float64
        assert max_abs_diff(lg.cpu().numpy(), d["logit_lr64"]) < 1e-10
        lg32 = mlr._hyper_logits(t(d["embed_lr"], dev), out_dtype=torch.float32)
        assert np.abs(lg32.cpu().numpy() - d["logit_lr"]).max() < 1e-5
        r = mapper.poincare_distance_origin(t(d["embed_lr"], dev), dim=1)
        assert max_abs_diff(r.cpu().numpy(), d["radius_lr"]) < 1e-13
        assert bits_equal(r.cpu().numpy(), ho.dist0(d["embed_lr"], 1.0, dim=1))
        up = bilinear_align_corners(t(d["logit_lr"], dev), (H, W))
        assert max_abs_diff(up.cpu().numpy(), d["logit"]) < 4e-6
        assert bits_equal(up.cpu().numpy(), ho.bilinear(d["logit_lr"], (H, W)))
        upe = bilinear_align_corners(t(d["embed_lr"], dev), (H, W))
        assert bits_equal(upe.cpu().numpy(), ho.bilinear(d["embed_lr"], (H, W)))


def test_hypermapper_lastdim_api(golden, dev):
    from halo_amd.core.utils.hyperbolic import HyperMapper
    d = golden("hypermapper")
    for c in (1.0, 0.5):
        m = HyperMapper(c=c)
        k = f"c{c}"
        xh = m.expmap(t(d[k + "__x"], dev))
        assert max_abs_diff(xh.cpu().numpy(), d[k + "__expmap"]) < 1e-14
        assert max_abs_diff(m.logmap(t(d[k + "__expmap"], dev)).cpu().numpy(), d[k + "__logmap"]) < 1e-12
        assert max_abs_diff(m.poincare_distance_origin(t(d[k + "__expmap"], dev)).cpu().numpy(), d[k + "__dist0"]) < 1e-12
        assert max_abs_diff(m.poincare_distance(t(d[k + "__expmap"], dev), t(d[k + "__y_h"], dev)).cpu().numpy(),
                            d[k + "__dist"]) < 1e-8
        assert max_abs_diff(m.expmap2(t(d[k + "__x"], dev).double()).cpu().numpy(), d[k + "__expmap2"]) < 1e-12
        assert max_abs_diff(m.cosine_distance(t(d[k + "__x"][6:], dev), t(d[k + "__y_h"][6:], dev).float()).cpu().numpy(),
                            d[k + "__cosine"]) < 1e-5


def test_hypermetrics_vs_reference(golden, dev):
    """HyperMetrics.compute (hyperbolic.py:191-228; dead in the reference tree, kept for API completeness) against the
    reference's own outputs: exponential maps and the Poincare distance through the HIP kernels."""
    from halo_amd.core.utils.hyperbolic import HyperMetrics
    d = golden("hypermapper")
    for c in (1.0, 0.5):
        k = f"hm_c{c}__"
        met = HyperMetrics(c=c).compute(t(d[k + "x"], dev), t(d[k + "y"], dev))
        assert set(met) == {"mse", "cosine_dist", "radius_x", "radius_y", "ang_e", "poincare_dist"}
        for key, tol in (("mse", 1e-7), ("cosine_dist", 1e-5), ("radius_x", 1e-12), ("radius_y", 1e-12), ("ang_e", 1e-9), ("poincare_dist", 1e-8)):
            got = met[key].cpu().numpy()
            assert got.dtype == d[k + key].dtype and got.shape == d[k + key].shape, key
            assert max_abs_diff(got, d[k + key]) < tol, (key, c)


# ------------------------------------------------------------------ score + selection on the golden vectors
@pytest.mark.parametrize("case,tag", all_case_combos())
def test_score_and_selection_golden(golden, case, tag, dev):
    from halo_amd.core.active.build import select_pixels_to_label
    from halo_amd.core.active.floating_region import FloatingRegionScore
    from oracle import halo_oracle as ho
    d = golden(case)
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    n = int(d["meta_n_regions"][0])
    unc, pur = COMBOS[tag]
    mrad, K, norm = (int(v) for v in d[tag + "__params"])
    frs = FloatingRegionScore(in_channels=O, size=3, purity_type=pur, K=K).to(dev)
    with torch.no_grad():
        s, i, u = frs(t(d["logit"], dev), decoder_out=t(d["embed"], dev), unc_type=unc, pur_type=pur,
                      normalize=bool(norm), ground_truth=t(d["gt"], dev))
    sn, inn, un = s.cpu().numpy(), i.cpu().numpy(), u.cpu().numpy()
    assert sn.shape == (H, W) and sn.dtype == d[tag + "__score"].dtype and inn.dtype == d[tag + "__impurity"].dtype
    # (1) bit-exact against the oracle
    so, io, uo = ho.floating_region_score(d["logit"], d["embed"], unc, pur, bool(norm), d["gt"], size=3,
                                          purity_type=pur, K=K)
    assert bits_equal(un, uo), "uncertainty differs from the oracle"
    assert bits_equal(inn, io), "impurity differs from the oracle"
    assert bits_equal(sn, so), "score differs from the oracle"
    # (2) within tolerance of the reference's vectors
    assert max_abs_diff(un, d[tag + "__uncertainty"]) < TOL
    assert max_abs_diff(inn, d[tag + "__impurity"]) < TOL        # 'hyper' included: zero quantiser bin flips on these vectors
    assert max_abs_diff(sn, d[tag + "__score"]) < TOL
    # (3) selection: two rounds, masks bit-exact against the reference
    act = t(d["prior_active"], dev).clone()
    sel = torch.zeros((H, W), dtype=torch.bool, device=dev)
    am = torch.full((H, W), 255, dtype=torch.int64, device=dev)
    gt = t(d["gt"], dev)
    for rnd in ("r1", "r2"):
        sc = s.clone()
        sc[act] = -float("inf")
        select_pixels_to_label(sc, n, 1, mrad, act, sel, am, gt)
        assert np.array_equal(act.cpu().numpy(), d[f"{tag}__{rnd}_active"])
        assert np.array_equal(sel.cpu().numpy(), d[f"{tag}__{rnd}_selected"])
        assert np.array_equal(am.cpu().numpy(), d[f"{tag}__{rnd}_active_mask"])
        if rnd == "r1":
            assert max_abs_diff(sc.cpu().numpy(), d[f"{tag}__r1_score"]) < TOL


@pytest.mark.parametrize("case,tag", all_case_combos())
def test_selector_on_reference_scores(golden, case, tag, dev):
    """Integer work only: fed the reference's score map, picks and masks must be identical."""
    from halo_amd.core.active.build import greedy_select
    d = golden(case)
    H, W, _, _ = (int(v) for v in d["meta_HWCO"])
    n = int(d["meta_n_regions"][0])
    mrad = int(d[tag + "__params"][0])
    act = t(d["prior_active"], dev).clone()[None]
    sel = torch.zeros((1, H, W), dtype=torch.bool, device=dev)
    am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
    gt = t(d["gt"], dev)[None]
    for rnd in ("r1", "r2"):
        sc = t(d[tag + "__score"], dev).clone()[None]
        sc[act] = -float("inf")
        picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, gt)
        ref = d[f"{tag}__{rnd}_picks"]
        k = int(npk[0])
        assert k == len(ref)
        p = picks[0, :k].cpu().numpy()
        assert np.array_equal(p[:, :2], ref[:, :2])
        assert bits_equal(p[:, 2], ref[:, 2])
        assert np.array_equal(act[0].cpu().numpy(), d[f"{tag}__{rnd}_active"])
        assert np.array_equal(sel[0].cpu().numpy(), d[f"{tag}__{rnd}_selected"])
        assert np.array_equal(am[0].cpu().numpy(), d[f"{tag}__{rnd}_active_mask"])


def test_select_accepts_cpu_indicators_like_the_reference_call_site(golden, dev):
    """build.py:115-120: score/active_mask/ground_truth on the device, active/selected on the CPU."""
    from halo_amd.core.active.build import select_pixels_to_label
    d = golden("case_b_64x128_c16_o19")
    sc = t(d["halo__score"], dev).clone()
    act = torch.from_numpy(d["prior_active"].copy())
    sel = torch.zeros(64, 128, dtype=torch.bool)
    am = torch.full((64, 128), 255, dtype=torch.int64, device=dev)
    sc[act.to(dev)] = -float("inf")
    out = select_pixels_to_label(sc, 60, 1, 5, act, sel, am, t(d["gt"], dev))
    assert out[1] is act and out[2] is sel and not act.is_cuda
    assert np.array_equal(act.numpy(), d["halo__r1_active"])
    assert np.array_equal(sel.numpy(), d["halo__r1_selected"])
    assert np.array_equal(am.cpu().numpy(), d["halo__r1_active_mask"])


# ------------------------------------------------------------------ seeded inputs vs the oracle at BASELINE sizes
def _synthetic(H, W, C, O, seed, dtype=np.float64):
    """Smooth maps like the real pipeline's: low-res latent -> oracle head -> x4 upsample."""
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(seed)
    z = (rng.standard_normal((1, C, H // 4, W // 4)) * 0.1).astype(np.float32)
    emb_lr = ho.expmap(z, 1.0, dim=1)
    bound = 1.0 / math.sqrt(C)
    P = rng.uniform(-bound, bound, (O, C))
    A = rng.uniform(-bound, bound, (O, C))
    logit_lr = ho.hypermlr(emb_lr, P, A, 1.0).astype(np.float32)
    logit = ho.bilinear(logit_lr, (H, W))
    emb = ho.bilinear(emb_lr, (H, W)).astype(dtype)
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    gt[rng.random((H, W)) < 0.05] = 255
    return logit, emb, gt


def _run_vs_oracle(dev, H, W, C, O, seed, unc, pur, norm, n, mrad, fdtype=np.float64, K=100, methods=("auto", "serial")):
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    logit, emb, gt = _synthetic(H, W, C, O, seed, fdtype)
    so, io, uo = ho.floating_region_score(logit, emb, unc, pur, norm, gt, size=3, purity_type=pur, K=K)
    with torch.no_grad():
        s, i, u = score_maps(t(logit, dev), t(emb, dev), unc, pur, norm, t(gt, dev)[None], size=3, K=K)
    assert bits_equal(u[0].cpu().numpy(), uo)
    assert bits_equal(i[0].cpu().numpy(), io)
    assert bits_equal(s[0].cpu().numpy(), so)
    act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
    _, _, _, _, picks_o = ho.select_pixels_to_label(so.copy(), n, 1, mrad, act_o, sel_o, am_o, gt, True)
    for method in methods:                       # the value-binned sweep (+ serial behind it) and the serial kernel alone
        act = torch.zeros((1, H, W), dtype=torch.bool, device=dev)
        sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        sc = s.clone()
        picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, t(gt, dev)[None], method=method)
        k = int(npk[0])
        assert k == len(picks_o), method
        assert bits_equal(picks[0, :k].cpu().numpy(), picks_o), "selection order differs from the oracle (%s)" % method
        assert np.array_equal(act[0].cpu().numpy(), act_o), method
        assert np.array_equal(sel[0].cpu().numpy(), sel_o), method
        assert np.array_equal(am[0].cpu().numpy(), am_o), method
        so2 = so.copy(); so2[act_o] = -np.inf
        assert bits_equal(sc[0].cpu().numpy(), so2), method                 # the score map is mutated like the reference's
    return picks[0, :k].cpu().numpy(), act[0].cpu().numpy(), sel[0].cpu().numpy()


@pytest.mark.parametrize("unc,pur,norm,mrad", [("entropy", "radius", True, 5), ("entropy", "ripu", False, 3),
                                               ("entropy", "hyper", True, 5), ("pixel_entropy", "euc_norm", True, 5),
                                               ("oracle_acc", "oracle_ripu", False, 3)])
def test_config1_256x512_c64_bit_exact_vs_oracle(dev, unc, pur, norm, mrad):
    """BASELINE.json configs[0]: 256x512, C=64, 19 classes; n = ceil(131072*0.01/9) = 146."""
    _run_vs_oracle(dev, 256, 512, 64, 19, 1234, unc, pur, norm, 146, mrad)


def test_config1_float32_features(dev):
    _run_vs_oracle(dev, 256, 512, 64, 19, 77, "entropy", "radius", True, 146, 5, fdtype=np.float32)


def test_odd_sizes_take_the_scalar_paths(dev):
    """H*W not a multiple of the vector width, class count without a specialised kernel."""
    _run_vs_oracle(dev, 36, 52, 5, 7, 5, "entropy", "radius", True, 12, 5)                   # hw % 4 == 0 only
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(3)
    H, W, C, O = 33, 47, 6, 11                                                               # hw odd
    logit = rng.standard_normal((1, O, H, W)).astype(np.float32)
    emb = (rng.standard_normal((1, C, H, W)) * 0.2)
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    for unc, pur in (("entropy", "radius"), ("entropy", "ripu"), ("oracle_acc", "hyper")):
        so, io, uo = ho.floating_region_score(logit, emb, unc, pur, True, gt, size=3, purity_type=pur, K=13)
        s, i, u = score_maps(t(logit, dev), t(emb, dev), unc, pur, True, t(gt, dev)[None], size=3, K=13)
        assert bits_equal(s[0].cpu().numpy(), so) and bits_equal(i[0].cpu().numpy(), io) and bits_equal(u[0].cpu().numpy(), uo)


@pytest.mark.parametrize("K", [16, 17, 100, 256, 300, 5000])
def test_histogram_class_sum_follows_torchs_cascade_for_any_bin_count(dev, K):
    """compute_region_impurity sums K one-hot channels with torch.sum, whose accumulator cascade flushes every 16 / 256 / 4096 terms
    (oracle: ATen's loop literally; device: the flush decided from the class index alone, ClassSum in halo_score.hip): both window
    kernels against the oracle bit for bit with bins on either side of every flush boundary."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(K)
    H, W, C, O = 40, 72, 6, 19
    logit = rng.standard_normal((1, O, H, W)).astype(np.float32)
    emb = rng.standard_normal((1, C, H, W)) * rng.uniform(0.01, 0.6, (1, 1, H, W))          # radii spread over the whole bin range
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    for size, env in ((3, {}), (3, {"HALO_IMPURITY_GENERIC": "1"}), (5, {})):
        so, io, uo = ho.floating_region_score(logit, emb, "entropy", "hyper", True, gt, size=size, purity_type="hyper", K=K)
        # (the module forces the 'hyper' purity window to 3 x 3 whatever `size` is, floating_region.py:54-55; score_maps takes it explicitly)
        s, i, u = _with_env(env, lambda: score_maps(t(logit, dev), t(emb, dev), "entropy", "hyper", True, t(gt, dev)[None], size=size, purity_size=3, K=K))
        assert bits_equal(i[0].cpu().numpy(), io) and bits_equal(u[0].cpu().numpy(), uo) and bits_equal(s[0].cpu().numpy(), so), (K, size, env)


def test_wider_windows(dev):
    """RADIUS_K = 2 (5x5 windows) and a 7x7 entropy window: the generic window paths."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    logit, emb, gt = _synthetic(64, 96, 8, 19, 9)
    for size in (5, 7):
        for unc, pur in (("entropy", "ripu"), ("entropy", "radius")):
            so, io, uo = ho.floating_region_score(logit, emb, unc, pur, True, gt, size=size, purity_type=pur)
            s, i, u = score_maps(t(logit, dev), t(emb, dev), unc, pur, True, t(gt, dev)[None], size=size)
            assert bits_equal(s[0].cpu().numpy(), so) and bits_equal(i[0].cpu().numpy(), io) and bits_equal(u[0].cpu().numpy(), uo)


def test_full_size_1024x2048_c256(dev):
    """BASELINE.json configs[1] shape, one image: bit-exact vs the oracle, plus properties."""
    H, W, n, mrad = 1024, 2048, 2331, 5
    picks, act, sel = _run_vs_oracle(dev, H, W, 256, 19, 1234, "entropy", "radius", True, n, mrad)
    assert len(picks) == n == math.ceil(H * W * 0.01 / 9)                     # build.py:148-150
    v = picks[:, 2]
    assert np.all(v[:-1] >= v[1:]), "greedy picks must be non-increasing in score"
    hw_ = picks[:, :2].astype(np.int64)
    # no two picks within the suppression window of each other
    order = np.lexsort((hw_[:, 1], hw_[:, 0]))
    p = hw_[order]
    for j in range(1, 12):
        dh = np.abs(p[j:, 0] - p[:-j, 0]); dw = np.abs(p[j:, 1] - p[:-j, 1])
        assert not np.any((dh <= mrad) & (dw <= mrad))
    assert sel.sum() <= 9 * n and sel.sum() >= 4 * n
    assert np.all(act[sel]), "selected pixels lie inside suppressed windows"
    assert act.sum() <= 121 * n


def _fullsize_tags():
    import sys
    sys.path.insert(0, GOLDEN)
    from make_fixtures import FULLSIZE
    return sorted(FULLSIZE)


@pytest.mark.parametrize("tag", _fullsize_tags())
def test_hip_reproduces_the_references_full_size_tables(dev, tag):
    """Reference-held golden at BASELINE size: the HIP scorer + selector on the inputs the REFERENCE ran on in
    the build container (tests/fullsize_inputs.build: numpy stream + oracle C, the digest is checked) against the reference's OWN
    pick table, mask digest and sampled maps (tests/golden/fullsize_picks.npz) -- not against the oracle."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fullsize_inputs as fi
    from make_fixtures import FULLSIZE, SAMPLE_STRIDE, mask_digest
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    d = np.load(os.path.join(GOLDEN, "fullsize_picks.npz"))
    seed, C, branch, mods, f32 = FULLSIZE[tag]
    unc, pur, norm, mrad, K = fi.BRANCHES[branch]
    inp = fi.build(seed, C=C, mods=mods, f32_embed=f32)
    assert fi.digest(inp).encode() == d[tag + "__digest"].tobytes(), "the inputs are not the ones the reference ran on"
    H, W = inp["gt"].shape
    n = fi.n_regions(H, W)
    with torch.no_grad():
        s, i, u = score_maps(t(inp["logit"], dev), t(inp["embed"], dev), unc, pur, norm, t(inp["gt"], dev)[None], size=3, K=K)
    act = t(inp["prior"], dev)[None].clone()
    sel = torch.zeros_like(act)
    am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
    sc = s.clone()
    sc[act] = -float("inf")                                                          # build.py:146
    s0 = sc[0].cpu().numpy().copy()
    picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, t(inp["gt"], dev)[None])
    k = int(npk[0])
    ref = d[tag + "__picks"]
    got = picks[0, :k].cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :2], ref[:, :2]), "%d of %d picks differ from the REFERENCE's" % (int((got[:, :2] != ref[:, :2]).any(axis=1).sum()), len(ref))
    assert np.abs(got[:, 2] - ref[:, 2]).max() < 1e-4
    res = dict(active=act[0].cpu().numpy(), selected=sel[0].cpu().numpy(), active_mask=am[0].cpu().numpy())
    assert np.array_equal(mask_digest(res), d[tag + "__mask_digest"]), "active / selected / active_mask differ from the REFERENCE's"
    for key, arr in (("score", s0), ("impurity", i[0].cpu().numpy()), ("uncertainty", u[0].cpu().numpy())):
        g_, w_ = arr.ravel()[::SAMPLE_STRIDE], d[f"{tag}__{key}_sample"]
        fin = np.isfinite(w_)
        assert g_.dtype == w_.dtype and np.array_equal(fin, np.isfinite(g_)), key
        assert np.abs(g_[fin].astype(np.float64) - w_[fin].astype(np.float64)).max() < 1e-4, key
        if key == "impurity" and branch in ("ripu", "hyper"):
            assert np.array_equal(g_, w_), "the window-histogram impurity is the reference's bit for bit"


def test_mid_size_fixture_hip_vs_reference_bitwise_up_to_the_logarithm(dev):
    """tests/golden/mid_112x192_c8_o19.npz (above ATen's 20480-pixel switch to oneDNN, like every production size): the HIP maps
    against the REFERENCE's -- impurity of `ripu` / `hyper` bit for bit, box-summed entropy equal except for a handful of pixels by
    one ulp (torch.log is MKL's closed-source vsLn; the contract's logf is the correctly rounded value), the device resize torch's
    bit for bit, picks exact.  (tests/test_oracle_golden.py states the same for the oracle.)"""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    import hashlib
    d = np.load(os.path.join(GOLDEN, "mid_112x192_c8_o19.npz"))
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    logit = bilinear_align_corners(t(d["logit_lr"], dev), (H, W))
    embed = bilinear_align_corners(t(d["embed_lr"], dev), (H, W))
    hsh = hashlib.sha256()
    hsh.update(logit.cpu().numpy().tobytes())
    hsh.update(embed.cpu().numpy().tobytes())
    assert np.array_equal(np.frombuffer(hsh.digest(), np.uint8), d["resized_digest"]), "the device resize is not torch's CPU kernel bit for bit"
    for tag, (unc, pur) in {"halo": ("entropy", "radius"), "ripu": ("entropy", "ripu"), "hyper": ("entropy", "hyper")}.items():
        mrad, K, norm = (int(v) for v in d[tag + "__params"])
        with torch.no_grad():
            s, i, u = score_maps(logit, embed, unc, pur, bool(norm), t(d["gt"], dev)[None], size=3, K=K)
        un, inn, sn = u[0].cpu().numpy(), i[0].cpu().numpy(), s[0].cpu().numpy()
        nd = int((un != d[tag + "__uncertainty"]).sum())
        assert nd <= 8 and np.abs(un - d[tag + "__uncertainty"]).max() <= 3e-7, (tag, nd)
        if pur != "radius":
            assert np.array_equal(inn, d[tag + "__impurity"]), tag
            assert int((sn != d[tag + "__score"]).sum()) <= 8, tag
        act = t(d["prior_active"], dev)[None].clone()
        sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        sc = s.clone()
        sc[act] = -float("inf")
        picks, npk = greedy_select(sc, 40, 1, mrad, act, sel, am, t(d["gt"], dev)[None])
        assert np.array_equal(picks[0, :int(npk[0]), :2].cpu().numpy(), d[tag + "__picks"][:, :2]), tag


def test_batched_equals_per_image_and_is_deterministic(dev):
    from halo_amd.core.active.build import acquire_batch
    B, H, W, C, O = 3, 128, 256, 16, 19
    data = [_synthetic(H, W, C, O, 100 + b) for b in range(B)]
    logit = torch.cat([t(d[0], dev) for d in data]); emb = torch.cat([t(d[1], dev) for d in data])
    gt = torch.stack([t(d[2], dev) for d in data])

    def run(lo, em, g):
        b = lo.shape[0]
        act = torch.zeros((b, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((b, H, W), 255, dtype=torch.int64, device=dev)
        picks, npk = acquire_batch(lo, em, g, act, sel, am, unc_type="entropy", pur_type="radius", normalize=True,
                                   n_regions=37, active_radius=1, mask_radius=5)
        return picks.cpu().numpy(), npk.cpu().numpy(), act.cpu().numpy(), am.cpu().numpy()

    full = run(logit, emb, gt)
    again = run(logit, emb, gt)
    for a, b_ in zip(full, again):
        assert np.array_equal(a, b_), "two runs on the same input differ"
    for b in range(B):
        one = run(logit[b:b + 1], emb[b:b + 1], gt[b:b + 1])
        assert np.array_equal(one[0][0], full[0][b]) and np.array_equal(one[2][0], full[2][b])


def test_exhaustion_nan_and_prior_mask_edge_cases(dev):
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    H, W = 24, 40
    rng = np.random.default_rng(0)
    for kind in ("ties", "nan", "all_masked", "neg_zero", "posinf", "constant", "few_pickable"):
        sc = rng.standard_normal((H, W))
        if kind == "ties":
            sc = np.round(sc * 2) / 2                       # many exact ties -> (min w, then min h) rule
        if kind == "nan":
            sc[5, 7] = np.nan; sc[3, 30] = np.nan           # NaN wins torch.max; first column first
        if kind == "all_masked":
            sc[:] = -np.inf
        if kind == "neg_zero":
            sc = np.where(rng.random((H, W)) < 0.5, -0.0, 0.0)
        if kind == "posinf":
            sc[4, 4] = np.inf; sc[20, 33] = np.inf          # +inf is a legitimate maximum (the binned sweep hands over)
        if kind == "constant":
            sc[:] = 0.75
        if kind == "few_pickable":
            sc[:] = -np.inf; sc[3:9, 4:30] = rng.standard_normal((6, 26))   # fewer pickable pixels than regions: stop at -inf
        for dt in (np.float64, np.float32):
            s0 = sc.astype(dt)
            act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
            gt = rng.integers(0, 19, (H, W)).astype(np.int64)
            so = s0.copy()
            _, _, _, _, po = ho.select_pixels_to_label(so, 500, 1, 5, act_o, sel_o, am_o, gt, True)
            for method in ("auto", "serial"):
                s = t(s0, dev)[None].clone()
                act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
                am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
                picks, npk = greedy_select(s, 500, 1, 5, act, sel, am, t(gt, dev)[None], method=method)
                k = int(npk[0])
                assert k == len(po), (kind, method)
                assert bits_equal(picks[0, :k].cpu().numpy(), po), (kind, method)
                assert bits_equal(s[0].cpu().numpy(), so), (kind, method)
                assert np.array_equal(act[0].cpu().numpy(), act_o) and np.array_equal(am[0].cpu().numpy(), am_o), (kind, method)


# ------------------------------------------------------------------ the driver
class _Fake(torch.nn.Module):
    def __init__(self, outs=None):
        super().__init__()
        self.outs, self.i = outs, 0

    def forward(self, x, size=None):
        if self.outs is None:
            return x
        o = self.outs[self.i % len(self.outs)]
        self.i += 1
        return o


@pytest.mark.parametrize("lowres", ["exact", "gram"])
def test_region_selection_driver_two_rounds(golden, dev, monkeypatch, lowres):
    """RegionSelection (build.py:71-186) through its PNG / torch.save persistence, two rounds -- the files the
    REFERENCE wrote for the same inputs, with the fused low-res scorer in its exact mode and (HALO_LOWRES=gram) in the
    Gram mode, whose radius map is rounded differently."""
    monkeypatch.setenv("HALO_LOWRES", lowres)
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    d = golden("region_selection")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs_")
    for i in range(3):
        Image.fromarray(np.full((H, W), 255, dtype=np.uint8)).save(os.path.join(tmp, f"m{i}.png"))
        torch.save({"active": torch.tensor([0], dtype=torch.bool), "selected": torch.tensor([0], dtype=torch.bool)},
                   os.path.join(tmp, f"i{i}.pth"))

    def loader():
        out = []
        for i in range(3):
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            a, s = ind["active"], ind["selected"]
            mask = torch.from_numpy(np.array(Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8)).long()
            if a.size() == (1,):
                a = torch.zeros(H, W, dtype=torch.bool); s = torch.zeros(H, W, dtype=torch.bool)
            out.append({"img": torch.zeros(1, 3, H // 2, W // 2), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
                        "origin_mask": mask[None], "origin_label": torch.from_numpy(d[f"img{i}__gt"])[None],
                        "size": torch.tensor([[H, W]]), "active": a[None], "selected": s[None],
                        "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"img{i}"]})
        return out

    for rnd in (1, 2):
        clf = _Fake([(t(d[f"img{i}__logit_lr"], dev), t(d[f"img{i}__embed_lr"], dev)) for i in range(3)])
        RegionSelection(cfg, _Fake(), clf, loader(), rnd)
        for i in range(3):
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            assert not ind["active"].is_cuda and ind["active"].dtype == torch.bool
            png = Image.open(os.path.join(tmp, f"m{i}.png"))
            assert png.mode == "L"
            assert np.array_equal(np.array(png, dtype=np.uint8), d[f"r{rnd}_img{i}__mask_png"])
            assert np.array_equal(ind["active"].numpy(), d[f"r{rnd}_img{i}__active"])
            assert np.array_equal(ind["selected"].numpy(), d[f"r{rnd}_img{i}__selected"])


# ------------------------------------------------------------------ helper methods and head tails
def test_helper_methods_vs_reference_and_oracle(golden, dev):
    from halo_amd.core.active.floating_region import FloatingRegionScore
    from oracle import halo_oracle as ho
    d = golden("helpers")
    f_hyp = FloatingRegionScore(in_channels=19, size=3, purity_type="hyper", K=100)
    f_rip = FloatingRegionScore(in_channels=19, size=5, purity_type="ripu")
    p, logit, gt = t(d["p"], dev), t(d["logit"], dev), t(d["gt"], dev)
    cases = {"pixel_entropy": lambda: f_hyp.compute_pixel_entropy(p),
             "ru_entropy_k3": lambda: f_hyp.compute_region_uncertainty("entropy", logit, p),
             "ru_entropy_k5": lambda: f_rip.compute_region_uncertainty("entropy", logit, p),
             "ru_oracle_acc": lambda: f_hyp.compute_region_uncertainty("oracle_acc", logit, p, ground_truth=gt),
             "ru_none": lambda: f_hyp.compute_region_uncertainty("none", logit, p),
             "ru_hyperbolic": lambda: f_hyp.compute_region_uncertainty("hyperbolic", logit, p)}
    ora = {"pixel_entropy": ("pixel_entropy", 3, False), "ru_entropy_k3": ("entropy", 3, True),
           "ru_entropy_k5": ("entropy", 5, True), "ru_oracle_acc": ("oracle_acc", 3, True),
           "ru_none": ("none", 3, False), "ru_hyperbolic": ("hyperbolic", 3, True)}
    for key, fn in cases.items():
        out = fn().cpu().numpy()
        assert out.shape == d[key].shape and out.dtype == np.float32
        assert max_abs_diff(out, d[key]) < TOL, key
        u, size, box = ora[key]
        assert bits_equal(out, ho.uncertainty_from_probs(d["p"], u, d["gt"], size, box)), key
    q = f_hyp.quantize_uncert_map(t(d["embed"], dev))
    assert q.dtype == torch.int64 and np.array_equal(q.cpu().numpy(), d["quantized"])
    imp, cnt = f_hyp.compute_region_impurity(q, 100)
    assert max_abs_diff(imp.cpu().numpy(), d["imp_hyper"]) < 1e-6 and np.array_equal(cnt.cpu().numpy(), d["cnt_hyper"])
    assert bits_equal(imp.cpu().numpy(), ho.region_impurity(d["quantized"], 100, 3)[0])
    imp, cnt = f_rip.compute_region_impurity(t(d["argmax"], dev), 19)
    assert max_abs_diff(imp.cpu().numpy(), d["imp_ripu_k5"]) < 1e-6 and np.array_equal(cnt.cpu().numpy(), d["cnt_ripu_k5"])
    with pytest.raises(RuntimeError):
        f_hyp.compute_region_impurity(t(d["argmax"], dev), 19)      # 100-channel purity window, 19-class labels


def test_head_tails(golden, dev):
    """classifier.py:364-379 (v2: resize logits AND embedding) and :552-558 (v3+: logits only)."""
    from halo_amd.core.models.classifier import hyper_head_tail
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners
    d = golden("case_b_64x128_c16_o19")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    mlr = HyperMLR(C, O, c=1.0).to(dev)
    with torch.no_grad():
        mlr.P_MLR.copy_(t(d["P_MLR"], dev)); mlr.A_MLR.copy_(t(d["A_MLR"], dev))
    with torch.no_grad():
        out, emb = hyper_head_tail(t(d["z"], dev), HyperMapper(1.0), mlr, size=(H, W), resize_embed=True)
        out3, emb3 = hyper_head_tail(t(d["z"], dev), HyperMapper(1.0), mlr, size=(H, W))
    assert out.dtype == torch.float32 and emb.dtype == torch.float64
    assert max_abs_diff(out.cpu().numpy(), d["logit"]) < 1e-5
    assert max_abs_diff(emb.cpu().numpy(), d["embed"]) < 1e-14
    assert emb3.shape[-2:] == (H // 4, W // 4) and max_abs_diff(emb3.cpu().numpy(), d["embed_lr"]) < 1e-14
    assert np.array_equal(out3.cpu().numpy(), out.cpu().numpy())
    # training mode (grad enabled, parameters require grad): same values, differentiable
    out_t, emb_t = hyper_head_tail(t(d["z"], dev), HyperMapper(1.0), mlr, size=(H, W))
    assert out_t.requires_grad and np.abs(out_t.detach().cpu().numpy() - d["logit"]).max() < 1e-5
    out_t.sum().backward()
    assert mlr.P_MLR.grad is not None and torch.isfinite(mlr.P_MLR.grad).all()
    with pytest.raises(NotImplementedError, match="inference-only"):
        bilinear_align_corners(t(d["embed_lr"], dev).requires_grad_(True), (H, W))   # refuses to detach silently


@pytest.mark.parametrize("c", [1.0, 0.7])
def test_logmap_and_distance_gradients_vs_reference_autograd(dev, c):
    """hyperbolic.py:51-83 under the reference's autograd (tests/golden/grads.npz, `ops_c*`): forward values are the HIP kernels'
    (equal to the no-grad call bit for bit), input gradients equal geoopt's -- row 4 sits on the projection limit."""
    from halo_amd.core.utils.hyperbolic import HyperMapper
    g = np.load(os.path.join(GOLDEN, "grads.npz"))
    tag = f"ops_c{c}"
    m = HyperMapper(c)
    x, y = t(g[tag + "__x"], dev), t(g[tag + "__y"], dev)
    W1, W2 = t(g[tag + "__W1"], dev), t(g[tag + "__W2"], dev)

    GTOL = 1e-9     # row 4: 1 / (1 - z^2) ~ 5e4 amplifies the last-bit differences between the device's and the host's log1p / norm

    def rel(a, ref):
        return float(np.abs(a.cpu().numpy() - ref).max() / np.abs(ref).max())

    a = x.clone().requires_grad_(True)
    out = m.logmap(a)
    assert out.requires_grad and torch.equal(out.detach(), m.logmap(x))
    (out * W1).sum().backward()
    assert rel(a.grad, g[tag + "__g_logmap"]) < GTOL
    a, b = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    out = m.poincare_distance(a, b)
    assert torch.equal(out.detach(), m.poincare_distance(x, y))
    (out * W2).sum().backward()
    assert rel(a.grad, g[tag + "__g_dist_x"]) < GTOL and rel(b.grad, g[tag + "__g_dist_y"]) < GTOL
    a = x.clone().requires_grad_(True)
    out = m.poincare_distance(a, y)                       # one side only
    (out * W2).sum().backward()
    assert rel(a.grad, g[tag + "__g_dist_x"]) < GTOL
    a = x.clone().requires_grad_(True)
    out = m.poincare_distance_origin(a)
    assert torch.equal(out.detach(), m.poincare_distance_origin(x))
    (out * W2).sum().backward()
    assert rel(a.grad, g[tag + "__g_dist0"]) < GTOL


def test_select_randomized_shapes_and_radii(dev):
    """Many small maps: sizes that are not multiples of the 16x32 tile, 1-pixel-wide maps, mask
    windows spanning many tiles (radius up to 40), active radius up to 3, heavy ties, both dtypes."""
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(2024)
    for trial in range(90):
        H = int(rng.integers(1, 80)); W = int(rng.integers(1, 120))
        mrad = int(rng.choice([0, 1, 2, 5, 9, 17, 40])); arad = int(rng.integers(0, 4))
        n = int(rng.integers(1, 60))
        dt = np.float32 if trial % 2 else np.float64
        sc = rng.standard_normal((H, W))
        if trial % 3 == 0:
            sc = np.round(sc * 3) / 3
        if trial % 7 == 0:
            sc[rng.random((H, W)) < 0.3] = -np.inf
        if trial % 5 == 1 and H > 4 and W > 4:             # smooth maps (clustered picks, neighbours with close values)
            sc = ho.bilinear(rng.standard_normal((1, (H + 3) // 4, (W + 3) // 4)), (H, W))[0]
        s0 = sc.astype(dt)
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        prior = rng.random((H, W)) < 0.05
        act_o = prior.copy(); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
        so = s0.copy()
        _, _, _, _, po = ho.select_pixels_to_label(so, n, arad, mrad, act_o, sel_o, am_o, gt, True)
        for method in ("auto", "serial"):
            s = t(s0, dev)[None].clone()
            act = t(prior, dev)[None].clone(); sel = torch.zeros_like(act)
            am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
            picks, npk = greedy_select(s, n, arad, mrad, act, sel, am, t(gt, dev)[None], method=method)
            k = int(npk[0])
            msg = f"trial {trial}: {H}x{W} mrad {mrad} arad {arad} n {n} {dt.__name__} {method}"
            assert k == len(po), msg
            assert bits_equal(picks[0, :k].cpu().numpy(), po), msg
            assert bits_equal(s[0].cpu().numpy(), so), msg
            assert np.array_equal(act[0].cpu().numpy(), act_o), msg
            assert np.array_equal(sel[0].cpu().numpy(), sel_o), msg
            assert np.array_equal(am[0].cpu().numpy(), am_o), msg


def test_large_map_uses_bigger_tiles(dev):
    """2048x4096 exceeds the 16x32 tile table (LDS) and must switch tile shape transparently."""
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(5)
    H, W, n = 2048, 4096, 300
    s0 = rng.standard_normal((H, W)).astype(np.float32)
    gt = np.zeros((H, W), np.int64)
    act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
    so = s0.copy()
    _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, 5, act_o, sel_o, am_o, gt, True)
    s = t(s0, dev)[None].clone()
    act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
    am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
    picks, npk = greedy_select(s, n, 1, 5, act, sel, am, t(gt, dev)[None])
    assert int(npk[0]) == n and bits_equal(picks[0].cpu().numpy(), po)
    assert np.array_equal(act[0].cpu().numpy(), act_o)


def test_views_and_batch_strides(dev):
    """A batch that is a strided view of a larger resident pool (batch stride != C*H*W)."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    H, W, C, O = 32, 64, 8, 19
    data = [_synthetic(H, W, C, O, 300 + b) for b in range(4)]
    pool_logit = torch.cat([t(d[0], dev) for d in data]); pool_emb = torch.cat([t(d[1], dev) for d in data])
    lv, ev = pool_logit[::2], pool_emb[::2]                       # images 0 and 2, batch stride doubled
    assert not lv.is_contiguous()
    s, i, u = score_maps(lv, ev, "entropy", "radius", True, None, size=3)
    for j, b in enumerate((0, 2)):
        so, io, uo = ho.floating_region_score(data[b][0], data[b][1], "entropy", "radius", True, None, size=3, purity_type="radius")
        assert bits_equal(s[j].cpu().numpy(), so) and bits_equal(u[j].cpu().numpy(), uo)
    # channels-last / permuted inputs are made contiguous by the host, results unchanged
    perm = pool_emb[:1].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    s2, _, _ = score_maps(pool_logit[:1], perm, "entropy", "radius", True, None, size=3)
    assert bits_equal(s2[0].cpu().numpy(), ho.floating_region_score(data[0][0], data[0][1], "entropy", "radius", True, None, size=3, purity_type="radius")[0])


# ------------------------------------------------------------------ fused upsample -> score (low-res sources)
@pytest.mark.parametrize("geom", [((16, 32), (16, 32), (64, 128)),        # x4, same grid for logits and embedding
                                  ((40, 80), (10, 20), (64, 128)),        # the real pipeline's ratios: x1.6 and x6.4
                                  ((23, 37), (9, 14), (50, 77)),          # odd everything
                                  ((64, 128), (64, 128), (64, 128)),      # identity resize
                                  ((5, 7), (3, 4), (96, 130)),            # large magnification
                                  ((32, 48), (32, 48), (64, 96)),         # x2: four output rows span three source rows (generic loop)
                                  ((21, 33), (21, 33), (64, 100))])       # x3: every row-sharing pattern, partial tiles
@pytest.mark.parametrize("unc,pur", [("entropy", "radius"), ("entropy", "hyper"), ("entropy", "ripu"),
                                      ("oracle_acc", "euc_norm")])
def test_lowres_sources_equal_upsample_then_score(dev, geom, unc, pur):
    from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    from oracle import halo_oracle as ho
    (hl, wl), (hf, wf), (H, W) = geom
    rng = np.random.default_rng(hl * 1000 + hf)
    B, C, O = 2, 12, 19
    logit_lr = rng.standard_normal((B, O, hl, wl)).astype(np.float32)
    emb_lr = ho.expmap((rng.standard_normal((B, C, hf, wf)) * 0.3).astype(np.float32), 1.0, dim=1)
    gt = rng.integers(0, O, (B, H, W)).astype(np.int64)
    act = rng.random((B, H, W)) < 0.03
    lg, em = t(logit_lr, dev), t(emb_lr, dev)
    a = score_maps_lowres(lg, em, (H, W), unc, pur, True, t(gt, dev), ksize=3, K=50, active=t(act, dev), mode="exact")
    b = score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(em, (H, W)), unc, pur, True, t(gt, dev),
                   size=3, K=50, active=t(act, dev))
    for x, y in zip(a, b):
        assert bits_equal(x.cpu().numpy(), y.cpu().numpy())
    # and against the oracle's upsample + score for image 0
    so, io, uo = ho.floating_region_score(ho.bilinear(logit_lr[:1], (H, W)), ho.bilinear(emb_lr[:1], (H, W)), unc, pur, True,
                                          gt[0], size=3, purity_type=pur, K=50)
    so[act[0]] = -np.inf
    assert bits_equal(a[0][0].cpu().numpy(), so) and bits_equal(a[1][0].cpu().numpy(), io) and bits_equal(a[2][0].cpu().numpy(), uo)


@pytest.mark.parametrize("geom", [((16, 32), (16, 32), (64, 128)), ((40, 80), (10, 20), (64, 128)), ((23, 37), (9, 14), (50, 77)),
                                  ((64, 128), (64, 128), (64, 128)), ((5, 7), (3, 4), (96, 130)), ((32, 48), (32, 48), (64, 96))])
@pytest.mark.parametrize("pur", ["radius", "euc_norm", "hyper"])
def test_lowres_gram_mode_tracks_the_exact_mode(dev, geom, pur):
    """mode='gram' (per-cell Gram terms, 10 products per output pixel instead of C) is the same number as
    the exact fused path rounded differently: the radius / norm maps agree to 1e-12 (float64; tolerance stated here),
    the uncertainty map -- untouched by the mode -- bit for bit, and the selection made from either score map is the same."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    (hl, wl), (hf, wf), (H, W) = geom
    rng = np.random.default_rng(hl * 77 + hf)
    B, C, O = 2, 24, 19
    logit_lr = rng.standard_normal((B, O, hl, wl)).astype(np.float32)
    emb_lr = ho.expmap((rng.standard_normal((B, C, hf, wf)) * 0.3).astype(np.float32), 1.0, dim=1)
    lg, em = t(logit_lr, dev), t(emb_lr, dev)
    a = score_maps_lowres(lg, em, (H, W), "entropy", pur, False, None, ksize=3, K=50, mode="exact")
    g = score_maps_lowres(lg, em, (H, W), "entropy", pur, False, None, ksize=3, K=50, mode="gram")
    assert bits_equal(a[2].cpu().numpy(), g[2].cpu().numpy())                     # uncertainty: same kernels
    if pur == "hyper":      # quantiser bins of the radius: a bin can only move where the radius sits on a bin edge to 1e-12
        assert (a[1] != g[1]).float().mean().item() < 1e-3
        return
    ia, ig = a[1].cpu().numpy(), g[1].cpu().numpy()
    assert ia.dtype == np.float64 and np.max(np.abs(ia - ig)) <= 1e-12 * max(1.0, np.max(np.abs(ia)))
    n = 40
    for sc in (a[0], g[0]):
        assert sc.dtype == torch.float64
    res = []
    for sc in (a[0].clone(), g[0].clone()):
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        gt = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
        picks, npk = greedy_select(sc, n, 1, 3, act, sel, am, gt)
        res.append((picks[:, :, :2].cpu().numpy().copy(), npk.cpu().numpy().copy(), act.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("geom", [((16, 32), (16, 32), (64, 128)), ((40, 80), (10, 20), (64, 128)), ((23, 37), (9, 14), (50, 77)),
                                  ((64, 128), (64, 128), (64, 128)), ((5, 7), (3, 4), (96, 130)), ((21, 33), (21, 33), (64, 100)),
                                  ((8, 8), (3, 63), (12, 250)), ((8, 8), (2, 64), (9, 257)), ((4, 4), (1, 1), (9, 11)), ((4, 4), (7, 1), (30, 6)),
                                  ((12, 20), (30, 44), (24, 40))])       # the last one DOWN-samples the embedding
@pytest.mark.parametrize("unc,pur,norm", [("entropy", "radius", True), ("pixel_entropy", "euc_norm", True), ("entropy", "hyper", True),
                                           ("entropy", "radius", False)])
def test_lowres_gram_mode_equals_its_oracle_twin_bitwise(dev, geom, unc, pur, norm):
    """Round 3: the 'gram' low-res mode (k_gram_lr + k_radius_gram) has a CPU twin (oracle.halo_oracle.gram_radius: the same
    operations in the same order), so the mode is pinned bit for bit like the exact one -- all three maps, every geometry
    (wave-width edge cases of the 63-column strips, single rows / columns, down-sampling), projected and zero vectors."""
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    (hl, wl), (hf, wf), (H, W) = geom
    rng = np.random.default_rng(hl * 131 + hf * 7 + wf)
    B, C, O = 2, 20, 19
    logit_lr = rng.standard_normal((B, O, hl, wl)).astype(np.float32)
    z = (rng.standard_normal((B, C, hf, wf)) * 0.3).astype(np.float32)
    z[0, :, 0, 0] = 0.0
    z[1, :, hf // 2:, wf // 2:] *= 200.0                      # projected onto the ball's boundary
    emb_lr = ho.expmap(z, 1.0, dim=1)
    gt = rng.integers(0, O, (B, H, W)).astype(np.int64)
    act = rng.random((B, H, W)) < 0.03
    g = score_maps_lowres(t(logit_lr, dev), t(emb_lr, dev), (H, W), unc, pur, norm, t(gt, dev), ksize=3, K=50, active=t(act, dev),
                          mode="gram")
    for b in range(B):
        raw = ho.gram_radius(emb_lr[b], (H, W), "euc_norm" if pur == "euc_norm" else "radius", 1.0)
        so, io, uo = ho.floating_region_score(ho.bilinear(logit_lr[b:b + 1], (H, W)), None, unc, pur, norm, gt[b], size=3,
                                              purity_type=pur, K=50, impurity_raw=raw)
        so = so.copy(); so[act[b]] = -np.inf
        assert bits_equal(g[0][b].cpu().numpy(), so) and bits_equal(g[1][b].cpu().numpy(), io) and bits_equal(g[2][b].cpu().numpy(), uo), b


def test_gram_mode_full_size_vs_oracle(dev):
    """the real boundary at full size -- C = 256 float64 embedding 160x320 and logits 640x1280 -> 1024x2048 --
    on SMOOTH embeddings (a 40x80 latent upsampled x4 before the exponential map, like real feature maps) that include a
    region projected onto the ball's boundary.  Per image: HIP 'gram' == its oracle twin bit for bit (three maps, 2331 picks,
    masks); against the oracle's upsample-then-score ('exact' order, the reference's) the normalised maps agree to 1e-12
    (tolerance stated here; observed 3e-13) and the picks / masks are identical.  The two orders differ by a few 1e-16 in the
    squared norm; artanh amplifies that by 1 / (1 - norm^2), so the invariant is stated on the norm: |tanh(r/2) - tanh(r'/2)|
    <= 1e-14 (observed 2.6e-15; raw radii near 9 differ by up to 2e-12, a vector AT the projection limit would by 2.5e-10).
    HALO_GRAM_IMAGES (default 2) images; tools/r03_gram_fullsize.sh ran 32 (profiles/archive/r03_gram_fullsize.txt)."""
    from halo_amd.core.active.build import acquire_batch_lowres
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    H, W, C, O, n = 1024, 2048, 256, 19, 2331
    n_images = int(os.environ.get("HALO_GRAM_IMAGES", "2"))
    bound = 1.0 / math.sqrt(C)
    worst = 0.0
    for i in range(n_images):
        rng = np.random.default_rng(9000 + i)
        lat = (rng.standard_normal((1, C, 40, 80)) * 0.12).astype(np.float32)
        z = ho.bilinear(lat, (160, 320))
        y0, x0 = int(rng.integers(0, 120)), int(rng.integers(0, 260))
        z[0, :, y0:y0 + 30, x0:x0 + 50] *= np.float32(40.0)             # saturates tanh: projected (boundary) vectors
        emb_lr = ho.expmap(z, 1.0, dim=1)
        logit160 = ho.hypermlr(emb_lr, rng.uniform(-bound, bound, (O, C)), rng.uniform(-bound, bound, (O, C)), 1.0).astype(np.float32)
        logit_lr = ho.bilinear(logit160, (640, 1280))
        gt = rng.integers(0, O, (H, W)).astype(np.int64)
        lg, em = t(logit_lr, dev), t(emb_lr, dev)
        g = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="gram")
        # the twin
        logit_up = ho.bilinear(logit_lr, (H, W))
        sg, ig, ug = ho.floating_region_score(logit_up, None, "entropy", "radius", True, None, size=3, purity_type="radius",
                                              impurity_raw=ho.gram_radius(emb_lr, (H, W)))
        assert bits_equal(g[0][0].cpu().numpy(), sg) and bits_equal(g[1][0].cpu().numpy(), ig) and bits_equal(g[2][0].cpu().numpy(), ug), i
        # the reference's order
        se, ie, ue = ho.floating_region_score(logit_up, ho.bilinear(emb_lr, (H, W)), "entropy", "radius", True, None, size=3,
                                              purity_type="radius")
        worst = max(worst, float(np.abs(sg - se).max()), float(np.abs(ig - ie).max()))
        assert np.abs(sg - se).max() <= 1e-12 and np.abs(ig - ie).max() <= 1e-12 and np.array_equal(ug, ue), (i, worst)
        r_g = score_maps_lowres(lg, em, (H, W), "none", "radius", False, None, ksize=3, mode="gram")[1][0].cpu().numpy()
        r_e = score_maps_lowres(lg, em, (H, W), "none", "radius", False, None, ksize=3, mode="exact")[1][0].cpu().numpy()
        assert np.abs(np.tanh(r_g / 2) - np.tanh(r_e / 2)).max() <= 1e-14, i
        res = []
        for sc in (sg, se):
            a, s_, m = np.zeros((H, W), bool), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
            _, _, _, _, pk = ho.select_pixels_to_label(sc.copy(), n, 1, 5, a, s_, m, gt, True)
            res.append((pk, a, s_, m))
        assert len(res[0][0]) == n and np.array_equal(res[0][0][:, :2], res[1][0][:, :2]), i     # the same pixels in the same order
        for x, y in zip(res[0][1:], res[1][1:]):
            assert np.array_equal(x, y), i
        act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        pk, nk = acquire_batch_lowres(lg, em, (H, W), t(gt, dev)[None], act, sel, am, unc_type="entropy", pur_type="radius", normalize=True,
                                      n_regions=n, active_radius=1, mask_radius=5, lowres_mode="gram")
        assert int(nk[0]) == n and bits_equal(pk[0].cpu().numpy(), res[0][0]), i
        assert np.array_equal(act[0].cpu().numpy(), res[0][1]) and np.array_equal(am[0].cpu().numpy(), res[0][3]), i
    print("gram vs upsample-then-score, %d full-size images: max |map difference| %.3g" % (n_images, worst))


def test_selection_with_the_scorers_range_records(dev):
    """Round 3: the scorer hands the selector the value range of its score maps (free for normalised maps: a product of two
    values in [0, 1]; the exact reduction otherwise), and the sweep skips its own pass over the map.  The binning is monotone
    whatever the bounds, so picks, tables and masks must be bit-identical to the selection that finds the range itself --
    HALO branch, the un-normalised ripu branch, prior-pick masks, a fully masked image, a constant (NaN after normalisation)
    map, float32 and float64 scores, the low-res entry points."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import new_score_range, score_maps, score_maps_lowres
    rng = np.random.default_rng(404)
    H, W, C, O, B, n = 96, 160, 12, 19, 4, 60
    for dt in (np.float64, np.float32):
        logit = rng.standard_normal((B, O, H, W)).astype(np.float32)
        emb = (rng.standard_normal((B, C, H, W)) * 0.05).astype(dt)
        emb[2] = emb[2, :, :1, :1]                                        # constant radius map -> 0/0 = NaN after normalisation
        gt = rng.integers(0, O, (B, H, W)).astype(np.int64)
        act = rng.random((B, H, W)) < 0.05
        act[1] = True                                                     # nothing pickable
        for unc, pur, norm in (("entropy", "radius", True), ("entropy", "ripu", False), ("entropy", "hyper", True), ("entropy", "radius", False)):
            res = []
            for ranged in (False, True):
                rec = new_score_range(B, dev) if ranged else None
                a, sl, am = t(act, dev), torch.zeros((B, H, W), dtype=torch.bool, device=dev), torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
                sc = score_maps(t(logit, dev), t(emb, dev), unc, pur, norm, t(gt, dev), size=3, K=20, active=a, want_maps=False,
                                score_range=rec)[0]
                pk, nk = greedy_select(sc, n, 1, 5, a, sl, am, t(gt, dev), score_range=rec)
                res.append((pk.cpu().numpy(), nk.cpu().numpy(), a.cpu().numpy(), sl.cpu().numpy(), am.cpu().numpy(), sc.cpu().numpy()))
            for x, y in zip(res[0], res[1]):
                assert bits_equal(x, y) if x.dtype.kind == "f" else np.array_equal(x, y), (dt, unc, pur, norm)
            assert res[0][1][1] == 0                                      # the fully masked image picks nothing either way
    # the exact records of an existing map == what the selector finds itself; low-res entry point
    from halo_amd import _lib
    sc = t(rng.standard_normal((2, H, W)), dev)
    rec = new_score_range(2, dev)
    _lib.check(_lib.lib().halo_score_range(_lib.ptr(sc), _lib.dtype_code(sc), 2, H, W, _lib.ptr(rec), _lib.stream_ptr(dev)), "halo_score_range")
    outs = []
    for r_ in (None, rec):
        s2 = sc.clone()
        a, sl, am = (torch.zeros((2, H, W), dtype=torch.bool, device=dev) for _ in range(2)) , None, None
        a, sl = a
        am = torch.full((2, H, W), 255, dtype=torch.int64, device=dev)
        outs.append(greedy_select(s2, n, 1, 5, a, sl, am, t(gt[:2], dev), score_range=r_)[0].cpu().numpy())
    assert bits_equal(outs[0], outs[1])
    # a record that does NOT bound its map: the record of a map in [0, 1] handed over with a map whose values
    # reach slightly below 0 and above 1 (and far outside) -- binning clamps explicitly, picks are those of the unranged path
    base = t(rng.random((2, H, W)), dev)
    rec = new_score_range(2, dev)
    _lib.check(_lib.lib().halo_score_range(_lib.ptr(base), _lib.dtype_code(base), 2, H, W, _lib.ptr(rec), _lib.stream_ptr(dev)), "halo_score_range")
    for spread in (1e-9, 1e-3, 0.5, 40.0):
        wild = base.clone()
        m = t(rng.random((2, H, W)) < 0.15, dev)
        wild[m] = wild[m] * (1 + 2 * spread) - spread                  # into [-spread, 1 + spread]
        wild[:, 3, 5], wild[:, H - 2, W - 7], wild[:, 7, 1] = -spread, 1 + spread, -0.0
        outs = []
        for r_ in (None, rec):
            s2 = wild.clone()
            a, sl = torch.zeros((2, H, W), dtype=torch.bool, device=dev), torch.zeros((2, H, W), dtype=torch.bool, device=dev)
            am = torch.full((2, H, W), 255, dtype=torch.int64, device=dev)
            pk, nk = greedy_select(s2, n, 1, 5, a, sl, am, t(gt[:2], dev), score_range=r_)
            outs.append((pk.cpu().numpy(), nk.cpu().numpy(), a.cpu().numpy(), am.cpu().numpy()))
        assert bits_equal(outs[0][0], outs[1][0]) and all(np.array_equal(x, y) for x, y in zip(outs[0][1:], outs[1][1:])), spread
        assert float(wild.min()) < 0 and float(wild.max()) > 1
    lg, em = t(rng.standard_normal((1, O, 24, 40)).astype(np.float32), dev), t(rng.standard_normal((1, C, 12, 20)) * 0.1, dev)
    outs = []
    for ranged in (False, True):
        rec = new_score_range(1, dev) if ranged else None
        sc = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, want_maps=False, score_range=rec)[0]
        a, sl, am = torch.zeros((1, H, W), dtype=torch.bool, device=dev), torch.zeros((1, H, W), dtype=torch.bool, device=dev), torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        outs.append(greedy_select(sc, n, 1, 5, a, sl, am, t(gt[:1], dev), score_range=rec)[0].cpu().numpy())
    assert bits_equal(outs[0], outs[1])


def test_fused_selector_histogram_is_consumed_once_and_survives_a_changed_map(dev, monkeypatch):
    """Round 4: for normalised maps the combine kernel counts the selector's 2048-bin coarse histogram while it
    writes the score and hands it over behind the range records; the selector skips its own pass over the map.  (1) same picks,
    tables and masks as with HALO_NO_FUSE_HIST=1 (the selector counts) and as without any record; (2) the counts describe the
    map as it was scored: a SECOND selection with the same records (windows now -inf) must not trust them -- the flag is consumed
    by the first; (3) a map the caller changed between scoring and selection (a large region masked by hand, so that fewer
    candidates exist above the threshold than the counts promise) still selects exactly what the unranged path selects: the sweep
    hands an exhausted image over instead of concluding that nothing is left."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import new_score_range, score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(13)
    B, C, O, H, W = 3, 8, 19, 96, 160
    emb = ho.bilinear(ho.expmap((rng.standard_normal((B, C, H // 4, W // 4)) * 0.2).astype(np.float32), 1.0, dim=1), (H, W))
    logit = ho.bilinear(rng.standard_normal((B, O, H // 4, W // 4)).astype(np.float32), (H, W))
    gt = t(rng.integers(0, O, (B, H, W)).astype(np.int64), dev)
    act0 = rng.random((B, H, W)) < 0.02
    lg, em = t(logit, dev), t(emb, dev)

    def run(ranged, n, premask=None, twice=False):
        rec = new_score_range(B, dev) if ranged else None
        act = t(act0, dev).clone()
        sc = score_maps(lg, em, "entropy", "radius", True, None, size=3, active=act, want_maps=False, score_range=rec)[0]
        if premask is not None:
            sc[premask] = -float("inf")
        sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        outs = []
        for _ in range(2 if twice else 1):
            pk, nk = greedy_select(sc, n, 1, 5, act, sel, am, gt, score_range=rec)
            outs.append((pk.cpu().numpy().copy(), nk.cpu().numpy().copy()))
        return outs, act.cpu().numpy(), sel.cpu().numpy(), am.cpu().numpy(), sc.cpu().numpy()

    n = 40
    base = run(False, n)
    fused = run(True, n)
    monkeypatch.setenv("HALO_NO_FUSE_HIST", "1")
    unfused = run(True, n)
    monkeypatch.delenv("HALO_NO_FUSE_HIST")
    for other in (fused, unfused):
        assert bits_equal(base[0][0][0], other[0][0][0]) and np.array_equal(base[0][0][1], other[0][0][1])
        assert all(np.array_equal(x, y) for x, y in zip(base[1:4], other[1:4])) and bits_equal(base[4], other[4])
    assert int(base[0][0][1].min()) == n
    # (2) two selections in a row with the same records
    b2, f2 = run(False, n, twice=True), run(True, n, twice=True)
    for k in range(2):
        assert bits_equal(b2[0][k][0], f2[0][k][0]) and np.array_equal(b2[0][k][1], f2[0][k][1])
    assert all(np.array_equal(x, y) for x, y in zip(b2[1:4], f2[1:4]))
    assert int(b2[0][1][1].min()) > 0                                              # the second round still found regions
    # (3) the caller masks the top 60 % of the rows after scoring: the counts are stale
    pm = torch.zeros((B, H, W), dtype=torch.bool, device=dev)
    pm[:, : int(H * 0.6)] = True
    n3 = 60
    b3, f3 = run(False, n3, premask=pm), run(True, n3, premask=pm)
    assert bits_equal(b3[0][0][0], f3[0][0][0]) and np.array_equal(b3[0][0][1], f3[0][0][1])
    assert all(np.array_equal(x, y) for x, y in zip(b3[1:4], f3[1:4])) and int(b3[0][0][1].min()) > 0


def test_score_and_select_replay_from_a_hip_graph(dev):
    """include/halo_hip.h promises that every call is an asynchronous enqueue with no allocation, no host sync and no global
    state, hence hipGraph-capturable.  Capture one acquisition (score -> range record -> mask -> select, ~12 launches incl. the
    selector's memset) into a graph on static buffers, replay it on new inputs, and compare with the eager path bit for bit."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import new_score_range, score_maps, _workspace
    rng = np.random.default_rng(808)
    B, O, C, H, W, n = 2, 19, 16, 128, 256, 80

    def inputs(seed):
        r = np.random.default_rng(seed)
        return (t(r.standard_normal((B, O, H, W)).astype(np.float32), dev), t(r.standard_normal((B, C, H, W)) * 0.05, dev),
                t(r.integers(0, O, (B, H, W)).astype(np.int64), dev), t(r.random((B, H, W)) < 0.03, dev))

    def eager(lg, em, gt, act0):
        act, sel, am = act0.clone(), torch.zeros_like(act0), torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        rec = new_score_range(B, dev)
        sc = score_maps(lg, em, "entropy", "radius", True, gt, size=3, active=act, want_maps=False, score_range=rec)[0]
        pk, nk = greedy_select(sc, n, 1, 5, act, sel, am, gt, score_range=rec)
        return [x.clone() for x in (sc, pk, nk, act, sel, am)]

    want = [eager(*inputs(s_)) for s_ in (1, 2, 3)]
    # static buffers of the graph
    lg, em, gt, act0 = inputs(1)
    act, sel, am = act0.clone(), torch.zeros_like(act0), torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    score = torch.empty((B, H, W), dtype=torch.float64, device=dev)
    picks = torch.zeros((B, n, 3), dtype=torch.float64, device=dev)
    npk = torch.zeros((B,), dtype=torch.int32, device=dev)
    rec = new_score_range(B, dev)
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        # warm-up on the capture stream: workspaces get allocated, dynamic-LDS limits raised, outside the capture
        score_maps(lg, em, "entropy", "radius", True, gt, size=3, active=act, want_maps=False, out=score, score_range=rec)
        greedy_select(score, n, 1, 5, act, sel, am, gt, out=(picks, npk), score_range=rec)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            score_maps(lg, em, "entropy", "radius", True, gt, size=3, active=act, want_maps=False, out=score, score_range=rec)
            greedy_select(score, n, 1, 5, act, sel, am, gt, out=(picks, npk), score_range=rec)
    for rep, seed in enumerate((1, 2, 3, 2)):
        l2, e2, g2, a2 = inputs(seed)
        lg.copy_(l2); em.copy_(e2); gt.copy_(g2); act.copy_(a2); sel.zero_(); am.fill_(255)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        w = want[seed - 1]
        for name, got, exp in zip(("score", "picks", "n_picked", "active", "selected", "active_mask"), (score, picks, npk, act, sel, am), w):
            same = bits_equal(got.cpu().numpy(), exp.cpu().numpy()) if got.dtype.is_floating_point else torch.equal(got, exp)
            assert same, (rep, seed, name, int((got != exp).sum()))


@pytest.mark.parametrize("shape", [(16, 12, 20, 48, 80), (256, 12, 20, 45, 77), (7, 9, 9, 64, 64), (64, 16, 32, 64, 128), (32, 30, 44, 120, 176)])
def test_gram_mode_is_guarded_against_cancellation(dev, shape):
    """VERDICT r3 #2 on hardware: opposing neighbours (v, -v (1 - eps), eps 1e-1 .. 1e-12) and opposing boundary vectors.
    HIP 'gram' == its oracle twin bit for bit (the guard's exact-order pixels included; even and odd source widths take
    k_gram_lr2 / k_gram_lr); against the EXACT-order oracle (ho.bilinear + floating_region_score) the norm tanh(r/2) agrees
    to 1e-11 relative (bound by construction 6.5e-11), the normalised maps to 1e-10, and picks, masks and pick
    order are identical."""
    from conftest import opposing_neighbours_embedding
    from halo_amd.core.active.build import acquire_batch_lowres
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    C, h, w, H, W = shape
    emb = opposing_neighbours_embedding(C, h, w, seed=C)
    rng = np.random.default_rng(5)
    logit_lr = rng.standard_normal((1, 19, h, w)).astype(np.float32)
    lg, em = t(logit_lr, dev), t(emb, dev)
    r_hip = score_maps_lowres(lg, em, (H, W), "none", "radius", False, None, ksize=3, mode="gram")[1][0].cpu().numpy()
    assert bits_equal(r_hip, ho.gram_radius(emb[0], (H, W), "radius", 1.0))
    e_hip = score_maps_lowres(lg, em, (H, W), "none", "euc_norm", False, None, ksize=3, mode="gram")[1][0].cpu().numpy()
    assert bits_equal(e_hip, ho.gram_radius(emb[0], (H, W), "euc_norm", 1.0))
    up, lgu = ho.bilinear(emb, (H, W)), ho.bilinear(logit_lr, (H, W))
    r_exact = ho.dist0(up, 1.0, dim=1)[0]
    n_exact, n_hip = np.tanh(r_exact / 2), np.tanh(r_hip / 2)
    assert n_exact.min() < 1e-3 and n_exact.max() > 0.999
    assert np.max(np.abs(n_hip - n_exact) / np.maximum(n_exact, 1e-300)) <= 1e-11
    assert (r_hip == r_exact).mean() > 0.3                                  # guarded pixels: the exact order's bits
    g = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="gram")
    so, io, uo = ho.floating_region_score(lgu, up, "entropy", "radius", True, None, size=3, purity_type="radius")
    assert np.max(np.abs(g[0][0].cpu().numpy() - so)) <= 1e-10 and np.max(np.abs(g[1][0].cpu().numpy() - io)) <= 1e-10
    assert bits_equal(g[2][0].cpu().numpy(), uo)
    n = max(4, H * W // 900)
    gt = rng.integers(0, 19, (H, W)).astype(np.int64)
    act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
    _, _, _, _, picks_o = ho.select_pixels_to_label(so.copy(), n, 1, 5, act_o, sel_o, am_o, gt, True)
    for mode in ("gram", "exact"):
        act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        picks, npk = acquire_batch_lowres(lg, em, (H, W), t(gt, dev)[None], act, sel, am, unc_type="entropy", pur_type="radius",
                                          normalize=True, n_regions=n, active_radius=1, mask_radius=5, lowres_mode=mode)
        k = int(npk[0])
        assert k == len(picks_o) > 0 and np.array_equal(picks[0, :k, :2].cpu().numpy(), picks_o[:, :2]), mode
        assert np.array_equal(act[0].cpu().numpy(), act_o) and np.array_equal(am[0].cpu().numpy(), am_o), mode
        if mode == "exact":
            assert bits_equal(picks[0, :k].cpu().numpy(), picks_o)


@pytest.mark.parametrize("C,hf,wf,H,W", [(42, 10, 20, 64, 128), (21, 10, 20, 64, 128), (28, 16, 32, 64, 128), (14, 16, 32, 64, 128),
                                          (20, 22, 44, 64, 128), (10, 22, 44, 64, 128), (63, 160, 320, 1024, 2048)])
def test_lowres_exact_mode_with_whole_channel_chunks(dev, C, hf, wf, H, W):
    """k_feat_reduce_lr_dmaf stages CCF channels per LDS image (21 / 14 / 10 for its 6x8, 7x10, 8x12 window geometries)
    and CCF * window < the DMA units of a wave, so the trailing units used to load channel c0 + CCF -- one plane past the image
    when C is a multiple of CCF, past the TENSOR for the last image of the batch.  They now re-load the chunk's last channel.
    Whole-chunk channel counts at x6.4 / x4 / x2.9, the embedding allocated exactly (no slack behind the last image), results
    bit-identical to upsample-then-score and, for the small ones, to the oracle."""
    from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(C * 7 + hf)
    B = 2 if H * W < 100000 else 1
    emb_lr = ho.expmap((rng.standard_normal((B, C, hf, wf)) * 0.2).astype(np.float32), 1.0, dim=1)
    logit_lr = rng.standard_normal((B, 19, hf, wf)).astype(np.float32)
    lg, em = t(logit_lr, dev), t(emb_lr, dev)
    a = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="exact")
    b = score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(em, (H, W)), "entropy", "radius", True, None, size=3)
    for x, y in zip(a, b):
        assert bits_equal(x.cpu().numpy(), y.cpu().numpy())
    if H * W < 100000:
        so, io, uo = ho.floating_region_score(ho.bilinear(logit_lr[-1:], (H, W)), ho.bilinear(emb_lr[-1:], (H, W)), "entropy", "radius",
                                              True, None, size=3, purity_type="radius")
        assert bits_equal(a[0][-1].cpu().numpy(), so) and bits_equal(a[1][-1].cpu().numpy(), io)


@pytest.mark.parametrize("mode", ["reflect", "replicate", "circular"])
def test_padding_modes_vs_reference_and_oracle(golden, dev, mode):
    """FloatingRegionScore(padding_mode=...) on the device (the generic box / window-histogram kernels with the
    tap index mapped into the image; the 3x3 fast paths are zero-padding only): bit-identical to the oracle, within 5e-6 of
    the reference's own class (tests/golden/padding.npz), identical first-round picks; the helper methods and the batched /
    low-resolution entry points take the mode too; torch's limits on the padding are reported as errors."""
    from halo_amd import _lib
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import FloatingRegionScore, score_maps, score_maps_lowres
    from oracle import halo_oracle as ho
    d = golden("padding")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    lg, em, gt = t(d["logit"], dev), t(d["embed"], dev), t(d["gt"], dev)
    for tag, unc, pur in (("halo", "entropy", "radius"), ("ripu5", "entropy", "ripu"), ("hyperK10", "entropy", "hyper"),
                          ("oracle", "oracle_acc", "oracle_ripu")):
        key = f"{mode}__{tag}"
        size, K, norm = (int(v) for v in d[key + "__params"])
        frs = FloatingRegionScore(in_channels=O, padding_mode=mode, size=size, purity_type=pur, K=K)
        s, i, u = frs(lg, em, unc_type=unc, pur_type=pur, normalize=bool(norm), ground_truth=gt)
        so, io, uo = ho.floating_region_score(d["logit"], d["embed"], unc, pur, bool(norm), d["gt"], size=size, purity_type=pur, K=K,
                                              padding_mode=mode)
        assert bits_equal(s.cpu().numpy(), so) and bits_equal(i.cpu().numpy(), io) and bits_equal(u.cpu().numpy(), uo), key
        assert max_abs_diff(s.cpu().numpy(), d[key + "__score"]) < 5e-6 and max_abs_diff(u.cpu().numpy(), d[key + "__uncertainty"]) < 5e-6, key
        act = t(d["prior_active"], dev)[None].clone(); sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        sc = s[None].clone()
        sc[act] = -float("inf")
        picks, npk = greedy_select(sc, 12, 1, 3, act, sel, am, gt[None])
        assert np.array_equal(picks[0, :int(npk[0]), :2].cpu().numpy(), d[key + "__picks"][:, :2]), key
        assert np.array_equal(am[0].cpu().numpy(), d[key + "__active_mask"]), key
    frs = FloatingRegionScore(in_channels=O, padding_mode=mode, size=5, purity_type="ripu")
    p = torch.softmax(lg[0], dim=0)
    assert max_abs_diff(frs.compute_region_uncertainty("entropy", lg[0], p).cpu().numpy(), d[f"{mode}__region_unc_k5"]) < 2.5e-5
    imp, cnt = frs.compute_region_impurity(p.argmax(dim=0), O)
    assert max_abs_diff(imp.cpu().numpy(), d[f"{mode}__imp_k5"]) < 1e-6 and np.array_equal(cnt.cpu().numpy(), d[f"{mode}__cnt_k5"])
    # batched + low-resolution entry points: same mode, bit-identical to the oracle on the upsampled inputs
    rng = np.random.default_rng(9)
    logit_lr = rng.standard_normal((2, O, 8, 12)).astype(np.float32)
    emb_lr = ho.expmap((rng.standard_normal((2, 6, 8, 12)) * 0.2).astype(np.float32), 1.0, dim=1)
    a = score_maps_lowres(t(logit_lr, dev), t(emb_lr, dev), (32, 48), "entropy", "radius", True, None, ksize=3, mode="exact", padding_mode=mode)
    for b in range(2):
        so, io, uo = ho.floating_region_score(ho.bilinear(logit_lr[b:b + 1], (32, 48)), ho.bilinear(emb_lr[b:b + 1], (32, 48)), "entropy",
                                              "radius", True, None, size=3, purity_type="radius", padding_mode=mode)
        assert bits_equal(a[0][b].cpu().numpy(), so) and bits_equal(a[2][b].cpu().numpy(), uo)
    # torch refuses a reflect padding that is not smaller than the map (and a circular one larger than it)
    tiny = torch.zeros((1, O, 2, 2), device=dev)
    if mode == "reflect":
        with pytest.raises(_lib.HaloHipError, match="reflect"):
            score_maps(tiny, None, "entropy", "ripu", False, None, size=5, padding_mode=mode)
    with pytest.raises(ValueError):
        score_maps(tiny, None, "entropy", "ripu", False, None, size=3, padding_mode="mirror")


@pytest.mark.parametrize("rows", ["2", "4", "8"])
def test_gram_kernel_strip_heights_are_invisible(dev, monkeypatch, rows):
    """k_gram_lr2 lets a wave own 2, 4 or 8 source rows (chosen by the size of the launch; HALO_GRAM_ROWS forces one): same fma
    chains per low-res pixel whatever the strip, so the maps are bit-identical to the oracle twin for every strip height --
    source heights that are no multiple of the strip, shorter than it, a single row, odd widths (the 8-byte kernel) included."""
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    monkeypatch.setenv("HALO_GRAM_ROWS", rows)
    rng = np.random.default_rng(int(rows))
    for (C, hf, wf, H, W) in ((24, 16, 32, 64, 128), (7, 9, 130, 30, 300), (5, 1, 64, 4, 200), (12, 3, 6, 17, 23), (9, 21, 258, 40, 500),
                              (6, 7, 5, 20, 16), (256, 32, 64, 64, 128)):
        emb_lr = ho.expmap((rng.standard_normal((2, C, hf, wf)) * 0.3).astype(np.float32), 1.0, dim=1)
        logit_lr = rng.standard_normal((2, 19, 8, 8)).astype(np.float32)
        g = score_maps_lowres(t(logit_lr, dev), t(emb_lr, dev), (H, W), "none", "radius", False, None, ksize=3, mode="gram")[1].cpu().numpy()
        for b in range(2):
            assert bits_equal(g[b], ho.gram_radius(emb_lr[b], (H, W), "radius", 1.0)), (rows, C, hf, wf, b)


def test_lowres_gram_mode_on_degenerate_grids(dev):
    """single-row / single-column / single-pixel embeddings, odd sizes around the 63-column wave width"""
    from halo_amd.core.active.floating_region import score_maps_lowres
    rng = np.random.default_rng(13)
    for (hf, wf), (H, W) in (((1, 1), (9, 11)), ((1, 5), (8, 40)), ((7, 1), (30, 6)), ((3, 63), (12, 250)), ((2, 64), (9, 257)),
                             ((5, 127), (20, 509)), ((1, 200), (3, 801))):
        lg = t(rng.standard_normal((1, 19, 4, 6)).astype(np.float32), dev)
        em = t(rng.standard_normal((1, 9, hf, wf)) * 0.2, dev)
        a = score_maps_lowres(lg, em, (H, W), "entropy", "radius", False, None, ksize=3, mode="exact")
        g = score_maps_lowres(lg, em, (H, W), "entropy", "radius", False, None, ksize=3, mode="gram")
        ia, ig = a[1].cpu().numpy(), g[1].cpu().numpy()
        assert np.all(np.isfinite(ig)) and np.max(np.abs(ia - ig)) <= 1e-12 * max(1.0, np.max(np.abs(ia))), ((hf, wf), (H, W))


def test_lowres_gram_mode_declines_float32_embeddings_quietly_and_validates_its_name(dev):
    from halo_amd.core.active.floating_region import score_maps_lowres
    rng = np.random.default_rng(12)
    lg = t(rng.standard_normal((1, 19, 20, 30)).astype(np.float32), dev)
    em = t((rng.standard_normal((1, 10, 12, 18)) * 0.2).astype(np.float32), dev)
    a = score_maps_lowres(lg, em, (60, 92), "entropy", "radius", True, None, ksize=3, mode="exact")
    g = score_maps_lowres(lg, em, (60, 92), "entropy", "radius", True, None, ksize=3, mode="gram")     # float32: exact path
    for x, y in zip(a, g):
        assert bits_equal(x.cpu().numpy(), y.cpu().numpy())
    with pytest.raises(ValueError):
        score_maps_lowres(lg, em, (60, 92), "entropy", "radius", True, None, ksize=3, mode="fast")


def test_lowres_sources_f32_embedding_and_other_class_counts(dev):
    from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    rng = np.random.default_rng(8)
    for O in (16, 7):
        lg = t(rng.standard_normal((1, O, 20, 30)).astype(np.float32), dev)
        em = t((rng.standard_normal((1, 10, 12, 18)) * 0.2).astype(np.float32), dev)
        a = score_maps_lowres(lg, em, (60, 92), "entropy", "radius", True, None, ksize=3)
        b = score_maps(bilinear_align_corners(lg, (60, 92)), bilinear_align_corners(em, (60, 92)), "entropy", "radius", True, None, size=3)
        assert a[0].dtype == torch.float32
        for x, y in zip(a, b):
            assert bits_equal(x.cpu().numpy(), y.cpu().numpy())


def test_lowres_downsampling_falls_back_to_explicit_upsample(dev):
    """A source far larger than the target does not fit the LDS window: HaloUnsupported -> explicit path."""
    from halo_amd._lib import HaloUnsupported
    from halo_amd.core.active.build import acquire_batch, acquire_batch_lowres
    from halo_amd.core.active.floating_region import score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    rng = np.random.default_rng(9)
    lg = t(rng.standard_normal((1, 19, 24, 40)).astype(np.float32), dev)
    em = t(rng.standard_normal((1, 4, 700, 1100)) * 0.05, dev)
    with pytest.raises(HaloUnsupported):
        score_maps_lowres(lg, em, (24, 40), "entropy", "radius", True, None, mode="exact")
    gt = t(rng.integers(0, 19, (1, 24, 40)).astype(np.int64), dev)

    def fresh():
        a = torch.zeros((1, 24, 40), dtype=torch.bool, device=dev)
        return a, torch.zeros_like(a), torch.full((1, 24, 40), 255, dtype=torch.int64, device=dev)
    a1, s1, m1 = fresh()
    p1, n1 = acquire_batch_lowres(lg, em, (24, 40), gt, a1, s1, m1, unc_type="entropy", pur_type="radius", normalize=True,
                                  n_regions=5, active_radius=1, mask_radius=5, lowres_mode="exact")
    a2, s2, m2 = fresh()
    p2, n2 = acquire_batch(lg, bilinear_align_corners(em, (24, 40)), gt, a2, s2, m2, unc_type="entropy", pur_type="radius",
                           normalize=True, n_regions=5, active_radius=1, mask_radius=5)
    assert torch.equal(p1, p2) and torch.equal(a1, a2) and torch.equal(m1, m2)


def test_hypermlr_matrix_core_path_matches_valu_path(golden, dev):
    """halo_hypermlr_logits runs its contractions on v_mfma_f64_16x16x4_f64 (<= 32 classes); the VALU
    kernel (one fma chain per pixel/class, HALO_MLR_VALU=1) is the in-library cross-check."""
    import os
    from halo_amd.core.utils.hyperbolic import HyperMLR
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(21)
    for (B, C, O, h, w) in ((1, 64, 19, 20, 36), (2, 256, 19, 16, 24), (1, 10, 16, 7, 9), (1, 33, 3, 5, 5), (1, 8, 32, 4, 13),
                            (3, 64, 19, 9, 7), (1, 256, 19, 33, 66), (2, 20, 24, 1, 1)):
        x = ho.expmap((rng.standard_normal((B, C, h, w)) * 0.2).astype(np.float32), 1.0, dim=1)
        x[0, :, 0, 0] = 0.0
        mlr = HyperMLR(C, O, c=1.0).to(dev)
        want = ho.hypermlr(x, mlr.P_MLR.detach().cpu().numpy(), mlr.A_MLR.detach().cpu().numpy(), 1.0)
        with torch.no_grad():
            os.environ.pop("HALO_MLR_VALU", None)
            a = mlr(t(x, dev)).cpu().numpy()
            a32 = mlr._hyper_logits(t(x, dev), out_dtype=torch.float32).cpu().numpy()
            os.environ["HALO_MLR_VALU"] = "1"
            try:
                v = mlr(t(x, dev)).cpu().numpy()
            finally:
                os.environ.pop("HALO_MLR_VALU", None)
            ch = _with_env({"HALO_MLR_CHUNKED": "1"}, lambda: mlr(t(x, dev)).cpu().numpy())       # round-1 matrix-core kernel
        assert max_abs_diff(a, want) < 1e-11 and max_abs_diff(v, want) < 1e-11 and max_abs_diff(ch, want) < 1e-11
        assert max_abs_diff(a, v) < 1e-12 and max_abs_diff(a, ch) < 1e-12
        assert np.abs(a32 - want.astype(np.float32)).max() < 1e-5
    # 33 classes: more than two column tiles -> VALU kernel
    mlr = HyperMLR(8, 33, c=1.0).to(dev)
    x = ho.expmap((rng.standard_normal((1, 8, 6, 6)) * 0.2).astype(np.float32), 1.0, dim=1)
    with torch.no_grad():
        got = mlr(t(x, dev)).cpu().numpy()
    assert max_abs_diff(got, ho.hypermlr(x, mlr.P_MLR.detach().cpu().numpy(), mlr.A_MLR.detach().cpu().numpy(), 1.0)) < 1e-11


@pytest.mark.parametrize("B,O,h,w,c,out_dtype", [(2, 19, 160, 320, 1.0, "float32"), (1, 19, 6, 10, 1.0, "float32"), (1, 16, 10, 14, 0.7, "float64"),
                                                 (3, 2, 4, 6, 1.0, "float32"), (1, 32, 8, 12, 2.0, "float32"), (1, 24, 9, 14, 1.0, "float64"),
                                                 (1, 19, 1, 2, 1.0, "float32")])
def test_fused_head_tail_equals_the_two_kernel_path_bit_for_bit(dev, B, O, h, w, c, out_dtype):
    """halo_head_tail (expmap -> project -> HyperMLR -> .float() in ONE kernel at the heads' own 64 channels, classifier.py:364-379,
    552-558) against the two calls it replaces: the SAME embedding and the SAME logits, bit for bit -- ordinary pixels, pixels far
    outside the ball (tanh clamp + projection), an exact-origin pixel, pixel counts that leave the last 32-pixel tile ragged,
    2 / 16 / 19 / 24 / 32 classes (all three column-tile counts), float32 and float64 logits, curvatures != 1; and the embedding
    within the oracle's tolerance.  HALO_HEAD_TAIL_SPLIT=1 and shapes the kernel does not serve take the two-kernel path."""
    from halo_amd.core.models.classifier import hyper_head_tail
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, head_tail_fused
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(B * 1000 + O)
    C = 64
    z = (rng.standard_normal((B, C, h, w)) * 0.1).astype(np.float32)
    z[0, :, 0, 0] *= 400.0                       # tanh clamp + project
    if h * w > 6:
        z[0, :, h // 2, w // 3] *= 60.0          # project only
        z[-1, :, h - 1, w - 1] = 0.0             # exact origin
    mapper = HyperMapper(c)
    mlr = HyperMLR(C, O, c=c).to(dev)
    odt = getattr(torch, out_dtype)
    zt = t(z, dev)
    with torch.no_grad():
        fused = head_tail_fused(zt, mlr.P_MLR, mlr.A_MLR, c, odt)
        if O > 24:               # 32 classes: the weight image + eight waves' staging rows exceed the 160 KB of LDS -> the two kernels
            assert fused is None
            o1, e1 = hyper_head_tail(zt, mapper, mlr)
            assert bits_equal(e1.cpu().numpy(), mapper.expmap(zt, dim=1).cpu().numpy())
            return
        assert fused is not None, "the fused kernel serves this shape"
        out_f, emb_f = fused
        emb_s = mapper.expmap(zt, dim=1)
        out_s = mlr._hyper_logits(emb_s, out_dtype=odt)
    assert emb_f.dtype == torch.float64 and out_f.dtype == odt
    assert bits_equal(emb_f.cpu().numpy(), emb_s.cpu().numpy()), "embedding differs from k_expmap0_project's"
    assert bits_equal(out_f.cpu().numpy(), out_s.cpu().numpy()), "logits differ from k_hypermlr_mfma_res's"
    assert np.abs(emb_f.cpu().numpy() - ho.expmap(z, c, dim=1)).max() < 1e-14
    # the head-tail function itself: fused by default, the two kernels under the switch, other channel counts as before
    with torch.no_grad():
        o1, e1 = hyper_head_tail(zt, mapper, mlr)
        o2, e2 = _with_env({"HALO_HEAD_TAIL_SPLIT": "1"}, lambda: hyper_head_tail(zt, mapper, mlr))
        o3, e3 = hyper_head_tail(zt[:, :48].contiguous(), mapper, HyperMLR(48, O, c=c).to(dev))
    assert bits_equal(o1.cpu().numpy(), o2.cpu().numpy()) and bits_equal(e1.cpu().numpy(), e2.cpu().numpy())
    assert e3.shape[1] == 48 and bool(torch.isfinite(o3).all())
    if out_dtype == "float32":
        assert bits_equal(o1.cpu().numpy(), out_f.cpu().numpy())


def test_hypermlr_epilogue_forms_agree(dev):
    """The matrix-core HyperMLR's one-quotient epilogue (default) against the reference-order statement it replaced
    (HALO_MLR_EPI_REF=1: also its rare arm): ordinary embeddings, every x on the ball's boundary, curvatures != 1, an exact-origin
    pixel; NaN / inf in x must come out as the reference-order form propagates them."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
    rng = np.random.default_rng(5)
    for (B, C, O, h, w, c, scale) in ((1, 64, 19, 20, 36, 1.0, 0.1), (2, 64, 19, 16, 24, 1.0, 0.7), (1, 128, 16, 9, 14, 0.5, 0.3),
                                      (1, 256, 20, 8, 8, 2.0, 0.2), (1, 64, 32, 6, 10, 0.1, 1.0)):
        z = (rng.standard_normal((B, C, h, w)) * scale).astype(np.float32)
        z[0, :, 0, 0] = 0.0
        x = HyperMapper(c).expmap(t(z, dev), dim=1).double()
        mlr = HyperMLR(C, O, c=c).to(dev)
        with torch.no_grad():
            a = mlr(x)
            r = _with_env({"HALO_MLR_EPI_REF": "1"}, lambda: mlr(x))
            assert bool(torch.isfinite(a).all())
            assert float((a - r).abs().max()) < 2e-12 * max(1.0, float(r.abs().max())), (B, C, O, c, scale, float((a - r).abs().max()))
            xn = x.clone()
            xn[0, 3, 1, 1] = float("nan"); xn[0, 5, 2, 3] = float("inf")
            an, rn = mlr(xn), _with_env({"HALO_MLR_EPI_REF": "1"}, lambda: mlr(xn))
            assert torch.equal(torch.isnan(an), torch.isnan(rn)) and bool(torch.isnan(an[0, :, 1, 1]).all())
            ok = ~torch.isnan(rn)
            assert float((an[ok] - rn[ok]).abs().max()) < 2e-12 * max(1.0, float(rn[ok].abs().max()))


def test_both_hypermlr_epilogues_feed_the_acquisition_the_same_files(golden, dev):
    """the acquisition's parity "starts from the logits", and in the real pipeline the logits are this package's own
    HyperMLR -- whose matrix-core epilogue (one quotient per logit, refined v_rcp / v_rsq, log-form asinh) is within ~4e-15 of the
    reference-order statement, not bit-equal to it.  Here the head tail runs on the reference's own latents and parameters
    (tests/golden/case_*.npz: z, P_MLR, A_MLR) with BOTH epilogues; the float32 logits must agree with the reference's float32
    logits except where a float64 value sits within the epilogue's error of a float32 rounding boundary (counted, bounded), and the
    masks / indicators / picks that the acquisition derives from each must be the reference's."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps_lowres
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
    for case in ("case_a_32x64_c8_o19", "case_b_64x128_c16_o19"):
        d = golden(case)
        H, W, C, O = (int(v) for v in d["meta_HWCO"])
        n = int(d["meta_n_regions"][0])
        mlr = HyperMLR(C, O, c=1.0).to(dev)
        with torch.no_grad():
            mlr.P_MLR.copy_(t(d["P_MLR"], dev)); mlr.A_MLR.copy_(t(d["A_MLR"], dev))
            emb = HyperMapper(1.0).expmap(t(d["z"], dev), dim=1)
            outs = {"one-quotient": mlr._hyper_logits(emb, out_dtype=torch.float32),
                    "reference-order": _with_env({"HALO_MLR_EPI_REF": "1"}, lambda: mlr._hyper_logits(emb, out_dtype=torch.float32))}
        for name, lg in outs.items():
            nd = int((lg.cpu().numpy() != d["logit_lr"]).sum())
            assert nd <= max(2, lg.numel() // 20000), (case, name, nd)          # float32 roundings of float64 values ~1e-15 apart
            assert np.abs(lg.cpu().numpy() - d["logit_lr"]).max() < 2e-6
            with torch.no_grad():
                s, _, _ = score_maps_lowres(lg, emb, (H, W), "entropy", "radius", True, t(d["gt"], dev)[None], ksize=3, K=100)
            act = t(d["prior_active"], dev)[None].clone()
            sel = torch.zeros_like(act)
            am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
            sc = s.clone()
            sc[act] = -float("inf")
            picks, npk = greedy_select(sc, n, 1, 5, act, sel, am, t(d["gt"], dev)[None])
            assert np.array_equal(picks[0, :int(npk[0]), :2].cpu().numpy(), d["halo__r1_picks"][:, :2]), (case, name)
            assert np.array_equal(act[0].cpu().numpy(), d["halo__r1_active"]) and np.array_equal(sel[0].cpu().numpy(), d["halo__r1_selected"]), (case, name)
            assert np.array_equal(am[0].cpu().numpy(), d["halo__r1_active_mask"]), (case, name)


def test_selection_is_stable_beside_streaming_kernels(dev):
    """The selector's window stores are drained one step late and masked analytically meanwhile
    (halo_select.hip); its loads see HBM latencies several times longer when the feature stream
    saturates the memory system.  Run it repeatedly beside that stream and demand identical picks,
    equal to the oracle's, every time."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    B, H, W, n = 6, 512, 1024, 600
    rng = np.random.default_rng(77)
    base = rng.standard_normal((B, H // 4, W // 4)).astype(np.float64)
    score0 = np.stack([ho.bilinear(base[b][None], (H, W))[0] for b in range(B)])        # smooth maps: clustered picks
    score0[1] = np.round(score0[1] * 8) / 8                                               # one image full of exact ties
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    want = []
    for b in range(B):
        so = score0[b].copy()
        act = np.zeros((H, W), bool); sel = np.zeros((H, W), bool); am = np.full((H, W), 255, np.int64)
        _, _, _, _, p = ho.select_pixels_to_label(so, n, 1, 5, act, sel, am, gt[b], True)
        want.append((p, act, am))
    feat = torch.randn((2, 256, 512, 1024), device=dev, dtype=torch.float64) * 0.01      # 2 GiB streamed per pass
    logit = torch.randn((2, 19, 512, 1024), device=dev)
    side = torch.cuda.Stream(dev)
    gtd, s0 = t(gt, dev), t(score0, dev)
    for rep in range(8):
        sc = s0.clone()
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                score_maps(logit, feat, "entropy", "radius", True, None, want_maps=False)
        picks, npk = greedy_select(sc, n, 1, 5, act, sel, am, gtd)
        torch.cuda.synchronize()
        for b in range(B):
            k = int(npk[b])
            assert k == len(want[b][0]), (rep, b)
            assert bits_equal(picks[b, :k].cpu().numpy(), want[b][0]), (rep, b)
            assert np.array_equal(act[b].cpu().numpy(), want[b][1]) and np.array_equal(am[b].cpu().numpy(), want[b][2]), (rep, b)


def test_region_selection_pipelined_pool_vs_oracle(dev):
    """More images than the in-flight depth, mixed label sizes, non-x4 resize ratios: every mask /
    indicator file equals the oracle driver's result, and all files exist when the call returns."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(31)
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs2_")
    sizes = [(48, 96), (64, 80), (48, 96), (40, 120), (64, 80), (48, 96), (56, 72)]
    items, outs, oracle_in = [], [], []
    for i, (H, W) in enumerate(sizes):
        emb_lr = ho.expmap((rng.standard_normal((1, 8, 10, 20)) * 0.2).astype(np.float32), 1.0, dim=1)
        logit_lr = rng.standard_normal((1, 19, 30, 50)).astype(np.float32)
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        act = rng.random((H, W)) < 0.02
        items.append({"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
                      "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
                      "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None],
                      "selected": torch.zeros(1, H, W, dtype=torch.bool),
                      "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"img{i}"]})
        outs.append((t(logit_lr, dev), t(emb_lr, dev)))
        oracle_in.append(dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act,
                              selected=np.zeros((H, W), bool), origin_mask=np.full((H, W), 255, np.int64)))
    RegionSelection(cfg, _Fake(), _Fake(outs), items, 1, in_flight=2, writer_threads=3)
    want = ho.region_selection(cfg, oracle_in, lowres_mode=_lr_mode())
    for i, (mask, act, sel, _) in enumerate(want):
        png = np.array(Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8)
        ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
        assert np.array_equal(png, mask), i
        assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), i


@pytest.mark.parametrize("staging", ["table", "device"])
@pytest.mark.parametrize("variant", ["loader_batch_3", "wide_labels", "uint8_loader", "oracle_branch"])
def test_region_selection_batched_and_mask_staging_vs_oracle(dev, variant, staging):
    """Round 4 host side: a loader batch > 1 goes through as ONE launch group when its images share a label size (and one by
    one when they do not); with mask_staging="table" (default) the int64 mask / label maps never travel -- the writer thread
    composes the uint8 mask from the pick table -- while "device" DMAs them and lets the selection kernel write the windows (the
    label map also travels when the scorer reads it: oracle_acc / oracle_ripu); a loader that hands uint8 masks is taken as it
    is.  Label values above 255 are included on purpose: only their low byte can reach the uint8 PNG in the reference either
    (build.py:67-68).  Every file equals the oracle driver's result."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(77)
    oracle_branch = variant == "oracle_branch"
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="oracle_acc" if oracle_branch else "entropy", PURITY="oracle_ripu" if oracle_branch else "radius",
                                     NORMALIZE=not oracle_branch, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs4_")
    sizes = [(48, 96)] * 3 + [(64, 80), (48, 96), (64, 80)] + [(40, 120)] * 3            # batches: uniform, mixed, uniform
    per, outs_lr, oracle_in = [], [], []
    for i, (H, W) in enumerate(sizes):
        emb_lr = ho.expmap((rng.standard_normal((1, 8, 10, 20)) * 0.2).astype(np.float32), 1.0, dim=1)
        logit_lr = rng.standard_normal((1, 19, 30, 50)).astype(np.float32)
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        gt[rng.random((H, W)) < 0.05] = 255
        om = np.full((H, W), 255, np.int64)
        prior = rng.random((H, W)) < 0.01
        om[prior] = gt[prior]
        if variant == "wide_labels":
            gt[rng.random((H, W)) < 0.02] += 256                       # only the low byte reaches the file (reference: uint8 cast)
            om[rng.random((H, W)) < 0.02] += 512
        act = rng.random((H, W)) < 0.02
        per.append(dict(om=om, gt=gt, act=act, H=H, W=W, emb=emb_lr, lg=logit_lr))
        oracle_in.append(dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act, selected=np.zeros((H, W), bool), origin_mask=om))
    nb = 3 if variant in ("loader_batch_3", "wide_labels", "oracle_branch") else 1
    items = []
    for k in range(0, len(per), nb):
        grp = per[k:k + nb]
        Hm, Wm = max(g_["H"] for g_ in grp), max(g_["W"] for g_ in grp)
        if len({(g_["H"], g_["W"]) for g_ in grp}) > 1:
            # a mixed-size batch cannot be collated into one tensor by a real loader: hand the driver per-image batches
            for j, g_ in enumerate(grp):
                items.append(([k + j], g_["H"], g_["W"]))
        else:
            items.append((list(range(k, k + len(grp))), Hm, Wm))
    mdt = torch.uint8 if variant == "uint8_loader" else torch.int64
    loader, head_outs = [], []
    for idxs, H, W in items:
        loader.append({"img": torch.zeros(len(idxs), 3, 8, 8),
                       "path_to_mask": [os.path.join(tmp, f"m{i}.png") for i in idxs], "path_to_indicator": [os.path.join(tmp, f"i{i}.pth") for i in idxs],
                       "origin_mask": torch.from_numpy(np.stack([per[i]["om"] for i in idxs])).to(mdt),
                       "origin_label": torch.from_numpy(np.stack([per[i]["gt"] for i in idxs])).to(mdt if not oracle_branch else torch.int64),
                       "size": torch.tensor([[H, W]] * len(idxs)), "active": torch.from_numpy(np.stack([per[i]["act"] for i in idxs])),
                       "selected": torch.zeros(len(idxs), H, W, dtype=torch.bool), "name": [f"img{i}" for i in idxs]})
        head_outs.append((t(np.concatenate([per[i]["lg"] for i in idxs]), dev), t(np.concatenate([per[i]["emb"] for i in idxs]), dev)))
    st = {}
    RegionSelection(cfg, _Fake(), _Fake(head_outs), loader, 1, in_flight=2, writer_threads=3, stats=st, mask_staging=staging)
    assert st["images"] == len(sizes) and st["batches"] == len(items) and st["main_launch_s"] > 0
    want = ho.region_selection(cfg, oracle_in, lowres_mode=_lr_mode())
    for i, (mask, act, sel, _) in enumerate(want):
        png = np.array(Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8)
        ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
        assert np.array_equal(png, mask), (variant, i)
        assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), (variant, i)
        assert ind["active"].dtype == torch.bool and int(sel.sum()) > 0


def test_head_tail_gradients_match_reference_autograd(golden, dev):
    """d loss / d {feat, P_MLR, A_MLR, embed} of the head tail (classifier.py:553-554) from the HIP backward
    kernels + library GEMMs vs the reference's own autograd (tests/golden/grads.npz), including
    tanh-clamped / projected pixels and an exact-origin pixel."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
    d = golden("grads")
    # c64_o19 is the head's own 64 x 19: the fused native backward (halo_hypermlr_backward); the same vectors once more through the
    # term-map path (HALO_MLR_BWD_TERMS=1), which is what the two smaller cases take anyway
    for tag, env in (("c8_o19", {}), ("c16_o16_k07", {}), ("c64_o19", {}), ("c64_o19", {"HALO_MLR_BWD_TERMS": "1"})):
      with _env_set(env):
          c = float(d[tag + "__c"][0])
          z = t(d[tag + "__z"], dev).requires_grad_(True)
          O, C = d[tag + "__P"].shape
          mlr = HyperMLR(C, O, c=c).to(dev)
          with torch.no_grad():
              mlr.P_MLR.copy_(t(d[tag + "__P"], dev)); mlr.A_MLR.copy_(t(d[tag + "__A"], dev))
          embed = HyperMapper(c=c).expmap(z, dim=1)
          embed.retain_grad()
          logits = mlr(embed.double()).float()
          assert max_abs_diff(embed.detach().cpu().numpy(), d[tag + "__embed"]) < 1e-14
          assert np.abs(logits.detach().cpu().numpy() - d[tag + "__logits"]).max() < 1e-5
          loss = (logits * t(d[tag + "__Wt"], dev)).sum() + (embed * t(d[tag + "__Ve"], dev)).sum()
          loss.backward()

          def rel(a, b):
              return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
          assert z.grad.dtype == torch.float32 and mlr.P_MLR.grad.dtype == torch.float64
          assert rel(embed.grad.cpu().numpy(), d[tag + "__g_embed"]) < 1e-10, tag
          assert rel(mlr.P_MLR.grad.cpu().numpy(), d[tag + "__g_P"]) < 1e-10, tag
          assert rel(mlr.A_MLR.grad.cpu().numpy(), d[tag + "__g_A"]) < 1e-10, tag
          assert rel(z.grad.cpu().numpy(), d[tag + "__g_z"]) < 2e-6, tag                     # float32 gradient
          # per-pixel check incl. the clamped / projected / origin pixels
          gz, want = z.grad.cpu().numpy(), d[tag + "__g_z"]
          assert np.abs(gz - want).max() <= 2e-6 * np.abs(want).max() + 1e-7


def test_hypermlr_fused_backward_matches_term_path(dev):
    """halo_hypermlr_backward (three kernels: reverse sweep + d x, d W partials, parameter algebra) against the term-map path
    (reverse-sweep kernel + library GEMMs, HALO_MLR_BWD_TERMS=1) it replaces at the heads' shapes: the same per-element sweep,
    sums in another order.  Shapes: ragged pixel counts (a partial last workgroup, an odd chunk), 1-4 channel blocks, 1-20
    classes, pixels beyond the projection limit, an exact-origin pixel; and twice the same call -> the same bits."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, _HyperMLRFn
    rng = np.random.default_rng(91)
    for (B, C, O, h, w, c) in ((2, 64, 19, 24, 40, 1.0), (1, 128, 19, 9, 7, 1.0), (1, 256, 5, 5, 13, 0.7), (2, 192, 20, 8, 8, 1.0),
                               (3, 64, 1, 1, 3, 1.0), (1, 64, 19, 33, 129, 1.3)):
        z = (rng.standard_normal((B, C, h, w)) * 0.15).astype(np.float32)
        z[0, :, 0, 0] *= 40.0
        z[0, :, 0, 1] = 0.0
        x0 = HyperMapper(c).expmap(t(z, dev), dim=1).double()
        bound = 1.0 / np.sqrt(C)
        P0, A0 = t(rng.uniform(-bound, bound, (O, C)), dev), t(rng.uniform(-bound, bound, (O, C)), dev)
        Wt = t(rng.standard_normal((B, O, h, w)), dev)
        res = []
        # 0, 1: the default twice   2: the pixel kernel's weights from LDS   3: its weights through the scalar cache   4: the term-map path
        # 5: the two separate d x / d W kernels instead of the one-pass kernel
        for env in ({}, {}, {"HALO_MLR_BWD_W": "lds"}, {"HALO_MLR_BWD_W": "scalar"}, {"HALO_MLR_BWD_TERMS": "1"}, {"HALO_MLR_BWD_DXW": "0"}):
            with _env_set(env):
                x, P, A = x0.clone().requires_grad_(True), P0.clone().requires_grad_(True), A0.clone().requires_grad_(True)
                (_HyperMLRFn.apply(x, P, A, c) * Wt).sum().backward()
                res.append([g.grad.cpu().numpy() for g in (x, P, A)])
        for a_, b_ in zip(res[0], res[1]):
            assert np.array_equal(a_, b_), "the fused backward is not deterministic"
        for a_, b_ in zip(res[2], res[3]):
            assert np.array_equal(a_, b_), "the two weight paths of the pixel kernel run the same fma chains: same bits"
        # the two separate d x / d W kernels against the one-pass kernel: the same MFMA chains for d x (same bits), other partial sums for d W
        assert np.array_equal(res[5][0], res[2][0]), "d x of the one-pass kernel differs from the d x kernel's"
        for name, a_, b_ in zip(("gP", "gA"), res[5][1:], res[2][1:]):
            assert np.abs(a_ - b_).max() <= 1e-12 * np.abs(b_).max() + 1e-300, ("two kernels", name, B, C, O, h, w)
        for other, tag in ((2, "weights from LDS"), (4, "term maps")):
            for name, a_, b_ in zip(("gx", "gP", "gA"), res[0], res[other]):
                assert np.isfinite(a_).all(), (name, B, C, O)
                assert np.abs(a_ - b_).max() <= 1e-11 * np.abs(b_).max() + 1e-300, (tag, name, B, C, O, h, w, float(np.abs(a_ - b_).max()), float(np.abs(b_).max()))
        # float32 logits straight from the forward kernel and a float32 gradient straight into the backward (the head's `.float()`
        # fused at both ends): the bits of the float64 route with the cast outside
        W32 = Wt.float()
        xa_, Pa_, Aa_ = x0.clone().requires_grad_(True), P0.clone().requires_grad_(True), A0.clone().requires_grad_(True)
        o32 = _HyperMLRFn.apply(xa_, Pa_, Aa_, c, torch.float32)
        assert o32.dtype == torch.float32
        (o32 * W32).sum().backward()
        xb_, Pb_, Ab_ = x0.clone().requires_grad_(True), P0.clone().requires_grad_(True), A0.clone().requires_grad_(True)
        o64 = _HyperMLRFn.apply(xb_, Pb_, Ab_, c)
        assert torch.equal(o32, o64.float())
        (o64.float() * W32).sum().backward()
        for ga_, gb_ in ((xa_.grad, xb_.grad), (Pa_.grad, Pb_.grad), (Aa_.grad, Ab_.grad)):
            assert torch.equal(ga_, gb_), "float32 gradient route differs from the float64 route"
        # the same input at an address that is 8 but not 16 bytes aligned (a contiguous view one element into a buffer): the
        # 16-byte operand loads of the weight-gradient kernel and of the forward give way to their scalar arms
        buf = torch.empty(x0.numel() + 1, dtype=torch.float64, device=dev)
        buf[1:].copy_(x0.reshape(-1))
        xu = buf[1:].view_as(x0).detach().requires_grad_(True)
        assert xu.data_ptr() % 16 == 8
        Pu, Au = P0.clone().requires_grad_(True), A0.clone().requires_grad_(True)
        (_HyperMLRFn.apply(xu, Pu, Au, c) * Wt).sum().backward()
        for name, a_, b_ in zip(("gx", "gP", "gA"), (xu.grad, Pu.grad, Au.grad), res[0]):
            assert np.abs(a_.cpu().numpy() - b_).max() <= 1e-11 * np.abs(b_).max() + 1e-300, ("unaligned", name, B, C, O, h, w)
    assert _lib_ws_zero_for_unserved_shapes()


def _lib_ws_zero_for_unserved_shapes():
    from halo_amd import _lib
    L = _lib.lib()
    return (L.halo_hypermlr_backward_workspace_bytes(1, 8, 19, 100) == 0 and L.halo_hypermlr_backward_workspace_bytes(1, 64, 21, 100) == 0
            and L.halo_hypermlr_backward_workspace_bytes(1, 320, 19, 100) == 0 and L.halo_hypermlr_backward_workspace_bytes(2, 64, 19, 100) > 0)


def test_empty_and_degenerate_inputs(dev):
    """Empty batch, zero regions, 1x1 / 1xN / Nx1 images."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    s, i, u = score_maps(torch.zeros((0, 19, 8, 8), device=dev), torch.zeros((0, 4, 8, 8), dtype=torch.float64, device=dev),
                         "entropy", "radius", True, None)
    assert s.shape == (0, 8, 8) and s.dtype == torch.float64 and u.shape == (0, 8, 8)
    z = torch.zeros((1, 8, 8), device=dev)
    picks, npk = greedy_select(z.double(), 0, 1, 5, z.bool(), z.bool(), z.long(), z.long())
    assert picks.shape[1] == 0 and int(npk[0]) == 0
    rng = np.random.default_rng(12)
    for (H, W) in ((1, 1), (1, 9), (7, 1), (2, 3)):
        logit = rng.standard_normal((1, 19, H, W)).astype(np.float32)
        emb = rng.standard_normal((1, 5, H, W)) * 0.2
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        for unc, pur in (("entropy", "radius"), ("entropy", "ripu"), ("oracle_acc", "hyper")):
            so, io, uo = ho.floating_region_score(logit, emb, unc, pur, True, gt, size=3, purity_type=pur, K=7)
            s, i, u = score_maps(t(logit, dev), t(emb, dev), unc, pur, True, t(gt, dev)[None], size=3, K=7)
            assert bits_equal(s[0].cpu().numpy(), so) and bits_equal(i[0].cpu().numpy(), io) and bits_equal(u[0].cpu().numpy(), uo), (H, W, unc, pur)


def test_bench_contract_on_a_small_shape(dev):
    """bench.py prints ONE JSON line carrying the contract's keys (run on a tiny shape so it takes seconds)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "64", "--width", "128", "--channels", "16",
                        "--steps", "3", "--warmup", "1", "--batch", "4", "--ring", "4", "--cpu-images", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "images/s" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "images/s" and d["parity_vs_cpu"] is True
    assert d["pipeline_tables_consistent"] is True
    # the oracle checks the TIMED launches' own tables, images spread over the batch, every occurrence in the round
    assert d["parity_images_checked"] >= 1 and d["parity_rows_checked"] >= d["parity_images_checked"]
    sel = d["selection"]
    assert sel["images"] == 12 and 0 <= sel["handed_over"] <= sel["images"] and isinstance(sel["reasons"], dict)
    assert d["roofline"]["traffic"] is None and "12 image evaluations" in d["config"]["workload"]


@pytest.mark.parametrize("mode", ["exact", "gram"])
def test_bench_lowres_source_is_oracle_checked_too(dev, mode):
    """bench.py --source lowres: the timed launches' pick tables against the oracle's upsample-then-score (exact) / Gram twin."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "128", "--width", "256", "--channels", "16",
                        "--steps", "3", "--warmup", "1", "--batch", "4", "--ring", "8", "--cpu-images", "2", "--source", "lowres",
                        "--lr-mode", mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["parity_vs_cpu"] is True and d["parity_images_checked"] == 2 and d["config"]["lowres_mode"] == mode
    assert d["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("data", ["late_round", "saturated", "peaked", "late_round+saturated+peaked", "plateau"])
def test_bench_data_variants_on_a_small_shape(dev, data):
    """bench.py --data: the value distributions that stress the selector (half the map already active, projected radii,
    saturated softmax) -- the timed tables still equal the oracle's and the hand-over counters are reported."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "128", "--width", "256", "--channels", "16",
                        "--steps", "3", "--warmup", "1", "--batch", "4", "--ring", "8", "--cpu-images", "4", "--data", data],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["parity_vs_cpu"] is True and d["parity_images_checked"] == 4 and d["config"]["data"] == data
    assert d["selection"]["images"] == 12 and d["pipeline_tables_consistent"] is True


def test_fused_and_unfused_entropy_paths_agree_bitwise(dev):
    """k_feat_reduce with the fused per-pixel entropy vs the stand-alone k_logit_maps kernel (HALO_NO_FUSE=1)."""
    from halo_amd.core.active.floating_region import score_maps
    for dt in (np.float64, np.float32):
        logit, emb, gt = _synthetic(128, 256, 24, 19, 5, dt)
        args = (t(logit, dev), t(emb, dev), "entropy", "radius", True, None)
        os.environ.pop("HALO_NO_FUSE", None)
        a = score_maps(*args, size=3)
        os.environ["HALO_NO_FUSE"] = "1"
        try:
            b = score_maps(*args, size=3)
        finally:
            os.environ.pop("HALO_NO_FUSE", None)
        for x, y in zip(a, b):
            assert bits_equal(x.cpu().numpy(), y.cpu().numpy())


def test_fused_tail_equals_the_round2_tail_bitwise(dev):
    """Round 3: the 3x3 box sum recomputed inside the combine kernel (k_combine_box3, edge taps through lane shifts) and
    the single finalize launch vs the round-2 sequence k_box3_unc -> 2 x finalize -> k_combine (HALO_NO_FUSE_TAIL=1), and
    both against the oracle: every branch that box-sums, normalised or not, with and without the prior-pick mask,
    widths that are and are not multiples of a wave's 256 pixels (rows that start inside a wave), single-row images."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(77)
    for (H, W, C) in ((64, 256, 8), (40, 72, 8), (33, 132, 5), (1, 8, 4), (3, 4, 4), (128, 1024, 4)):
        for dt in (np.float64, np.float32):
            logit = rng.standard_normal((2, 19, H, W)).astype(np.float32)
            emb = (rng.standard_normal((2, C, H, W)) * 0.05).astype(dt)
            gt = rng.integers(0, 19, (2, H, W)).astype(np.int64)
            act = rng.random((2, H, W)) < 0.1
            for unc, pur, norm in (("entropy", "radius", True), ("entropy", "radius", False), ("entropy", "ripu", False),
                                   ("entropy", "hyper", True), ("oracle_acc", "oracle_ripu", True), ("entropy", "none", True),
                                   ("entropy", "euc_norm", True)):
                for a in (None, act):
                    args = (t(logit, dev), t(emb, dev), unc, pur, norm, t(gt, dev))
                    kw = dict(size=3, K=7, active=None if a is None else t(a, dev))
                    new = score_maps(*args, **kw)
                    old = _with_env({"HALO_NO_FUSE_TAIL": "1"}, lambda: score_maps(*args, **kw))
                    for x, y in zip(new, old):
                        assert bits_equal(x.cpu().numpy(), y.cpu().numpy()), (H, W, dt, unc, pur, norm, a is None)
                    for b in range(2):
                        so, io, uo = ho.floating_region_score(logit[b:b + 1], emb[b:b + 1], unc, pur, norm, gt[b], size=3,
                                                              purity_type=pur, K=7)
                        if a is not None:
                            so = so.copy(); so[a[b]] = -np.inf
                        assert bits_equal(new[0][b].cpu().numpy(), so) and bits_equal(new[1][b].cpu().numpy(), io) and \
                            bits_equal(new[2][b].cpu().numpy(), uo), (H, W, dt, unc, pur, norm, b)


def test_feature_kernel_chunk_map_is_invisible(dev):
    """k_feat_reduce deals its 2 KiB pixel chunks to workgroups XCD-contiguously (granule 256 chunks, halved to fit small maps,
    plain map past the last whole group): the three maps are the same bits as with the plain map (HALO_FEAT_XCD_GRANULE=-1),
    with odd granules, and as the oracle's -- chunk counts that are / are not multiples of 8 granules, a partial last chunk."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(404)
    for (H, W, C) in ((64, 256, 8), (40, 72, 8), (96, 1000, 4), (128, 1024, 4), (1024, 2048, 2), (5, 11, 3)):
        for dt in (np.float64, np.float32):
            logit = rng.standard_normal((2, 19, H, W)).astype(np.float32)
            emb = (rng.standard_normal((2, C, H, W)) * 0.05).astype(dt)
            args = (t(logit, dev), t(emb, dev), "entropy", "radius", True, None)
            new = score_maps(*args, size=3)
            for g in ("-1", "3", "1", "1000000"):
                old = _with_env({"HALO_FEAT_XCD_GRANULE": g}, lambda: score_maps(*args, size=3))
                for x, y in zip(new, old):
                    assert bits_equal(x.cpu().numpy(), y.cpu().numpy()), (H, W, dt, g)
            if H * W <= 128 * 1024:
                so, io, uo = ho.floating_region_score(logit[1:2], emb[1:2], "entropy", "radius", True, None, size=3, purity_type="radius")
                assert bits_equal(new[0][1].cpu().numpy(), so) and bits_equal(new[1][1].cpu().numpy(), io) and bits_equal(new[2][1].cpu().numpy(), uo)


def test_scorer_tail_on_a_second_stream_equals_the_inline_call(dev):
    """halo_score_maps_split: the passes over the inputs on the current stream, everything behind them on `tail_stream` (forked at
    the stop event) -- same bits as the one-stream call, for every branch that reads decoder_out, with the range record, several
    calls in flight on rotating workspaces."""
    from halo_amd import _lib
    from halo_amd.core.active.floating_region import new_score_range, score_maps, score_workspace
    rng = np.random.default_rng(515)
    H, W, C, B = 96, 256, 8, 3
    tail = torch.cuda.Stream(dev, priority=-1)
    L = _lib.lib()
    for dt in (np.float64, np.float32):
        for unc, pur, norm in (("entropy", "radius", True), ("entropy", "hyper", True), ("entropy", "euc_norm", False), ("none", "radius", True)):
            calls = []
            for it in range(3):
                logit = rng.standard_normal((B, 19, H, W)).astype(np.float32)
                emb = (rng.standard_normal((B, C, H, W)) * 0.05).astype(dt)
                act = rng.random((B, H, W)) < 0.1
                calls.append((t(logit, dev), t(emb, dev), t(act, dev)))
            want = [score_maps(lg, em, unc, pur, norm, None, size=3, K=7, active=ac, score_range=new_score_range(B, dev)) for lg, em, ac in calls]
            torch.cuda.synchronize(dev)
            got, keep = [], []
            for lg, em, ac in calls:                       # three calls in flight, each with its own workspace / maps / events
                odt = want[0][0].dtype
                ws = score_workspace(B, H, W, dev)
                maps = (torch.empty((B, H, W), dtype=odt, device=dev), torch.empty((B, H, W), dtype=torch.float32, device=dev))
                ev = (L.halo_event_create(), L.halo_event_create())
                rec = new_score_range(B, dev)
                keep.append((ws, ev, rec))
                got.append(score_maps(lg, em, unc, pur, norm, None, size=3, K=7, active=ac, events=ev, score_range=rec,
                                      tail_stream=tail, workspace=ws, maps=maps))
            tail.synchronize()
            for a, b in zip(want, got):
                for x, y in zip(a, b):
                    assert bits_equal(x.cpu().numpy(), y.cpu().numpy()), (dt, unc, pur, norm)
            for _, ev, _ in keep:
                L.halo_event_destroy(ev[0]); L.halo_event_destroy(ev[1])
    lg, em, ac = calls[0]
    with pytest.raises(AssertionError):
        score_maps(lg, em, "entropy", "radius", True, None, size=3, tail_stream=tail)            # no events / workspace / maps
    with pytest.raises(_lib.HaloHipError):
        ws = score_workspace(B, H, W, dev)
        maps = (torch.empty((B, H, W), dtype=torch.float32, device=dev), torch.empty((B, H, W), dtype=torch.float32, device=dev))
        ev = (L.halo_event_create(), L.halo_event_create())
        score_maps(lg, None, "entropy", "ripu", False, None, size=3, tail_stream=tail, events=ev, workspace=ws, maps=maps)   # no decoder_out pass to fork behind


def test_region_selection_full_size_real_geometry_vs_oracle(dev):
    """The real pipeline's geometry at full label size: logits 640x1280 and a C=64 float64 embedding at
    160x320 resized to 1024x2048 inside the scorer (never materialised on the device), 2331 regions --
    masks and indicators written by RegionSelection equal the oracle driver's (explicit upsample) output."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(2026)
    H, W, C, O = 1024, 2048, 64, 19
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 15000, 30000, 40000, 50000], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs3_")
    emb_lr = ho.expmap((rng.standard_normal((1, C, 160, 320)) * 0.1).astype(np.float32), 1.0, dim=1)
    bound = 1.0 / math.sqrt(C)
    logit160 = ho.hypermlr(emb_lr, rng.uniform(-bound, bound, (O, C)), rng.uniform(-bound, bound, (O, C)), 1.0).astype(np.float32)
    logit_lr = ho.bilinear(logit160, (640, 1280))                      # the v3+ head resizes logits to the input size
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    act = np.zeros((H, W), bool); act[100:140, 300:380] = True
    item = {"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, "m.png")],
            "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
            "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None], "selected": torch.zeros(1, H, W, dtype=torch.bool),
            "path_to_indicator": [os.path.join(tmp, "i.pth")], "name": ["img"]}
    RegionSelection(cfg, _Fake(), _Fake([(t(logit_lr, dev), t(emb_lr, dev))]), [item], 1)
    (mask, a_o, s_o, picks), = ho.region_selection(cfg, [dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act,
                                                            selected=np.zeros((H, W), bool), origin_mask=np.full((H, W), 255, np.int64))],
                                                   lowres_mode=_lr_mode())
    assert len(picks) == 2331
    ind = torch.load(os.path.join(tmp, "i.pth"))
    assert np.array_equal(np.array(Image.open(os.path.join(tmp, "m.png")), dtype=np.uint8), mask)
    assert np.array_equal(ind["active"].numpy(), a_o) and np.array_equal(ind["selected"].numpy(), s_o)


def test_region_selection_writes_the_references_files_at_full_size(dev):
    """Reference-held golden for the path every real RegionSelection call runs (N1 + N2): the HIP driver over the real pipeline's
    geometry (64-channel float64 embedding at 160 x 320, logits at 640 x 1280, labels 1024 x 2048, 2331 regions), two rounds through
    its own PNG / indicator files, against the digests of the files the REFERENCE's RegionSelection wrote for the same arrays in the
    build container (tests/golden/fullsize_driver.npz, tests/golden/make_fixtures.py:gen_fullsize_driver)."""
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, GOLDEN)
    import fullsize_inputs as fi
    from make_fixtures import DRIVER_SEEDS, files_digest
    from halo_amd.core.active.build import RegionSelection
    d = np.load(os.path.join(GOLDEN, "fullsize_driver.npz"))
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs_ref_")
    for seed in DRIVER_SEEDS:
        inp = fi.build_driver_inputs(seed)
        assert fi.digest(inp).encode() == d[f"s{seed}__digest"].tobytes()
        H, W = inp["gt"].shape
        pm, pi = os.path.join(tmp, f"m{seed}.png"), os.path.join(tmp, f"i{seed}.pth")
        mask, act, sel = torch.full((H, W), 255, dtype=torch.int64), torch.zeros(H, W, dtype=torch.bool), torch.zeros(H, W, dtype=torch.bool)
        outs = [(t(inp["logit_lr"], dev), t(inp["embed_lr"], dev))]
        for rnd in (1, 2):
            item = {"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [pm], "origin_mask": mask[None], "origin_label": torch.from_numpy(inp["gt"])[None],
                    "size": torch.tensor([[H, W]]), "active": act[None], "selected": sel[None], "path_to_indicator": [pi], "name": [f"s{seed}"]}
            RegionSelection(cfg, _Fake(), _Fake(outs), [item], rnd)
            png = np.array(Image.open(pm), dtype=np.uint8)
            ind = torch.load(pi)
            act, sel = ind["active"], ind["selected"]
            assert [int(sel.sum()), int(act.sum()), int((png != 255).sum())] == list(d[f"s{seed}__r{rnd}_counts"]), rnd
            assert np.array_equal(files_digest(png, act.numpy(), sel.numpy()), d[f"s{seed}__r{rnd}_files_digest"]), "round %d: not the REFERENCE's files" % rnd
            mask = torch.from_numpy(png).long()                                    # what the loader reads back (cityscapes.py:234)


def test_autograd_gradcheck_float64(dev):
    """Finite-difference check (torch.autograd.gradcheck, float64) of the HIP backward kernels: expmap over the
    last dim and over dim=1 (inside the ball, tanh-clamped + projected, and mixed), HyperMLR w.r.t. x, P, A."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, _HyperMLRFn
    g = torch.Generator(device=dev).manual_seed(4)
    m = HyperMapper(c=0.8)
    x = (torch.randn((5, 7), generator=g, device=dev, dtype=torch.float64) * 0.4).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t_: m.expmap(t_), (x,), eps=1e-6, atol=1e-7, rtol=1e-5)
    xb = (torch.randn((2, 6, 3, 4), generator=g, device=dev, dtype=torch.float64) * 0.3)
    xb[0, :, 0, 0] *= 30.0                                   # projected pixel (not tanh-clamped: ||u|| sqrt(c) < 15)
    xb.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t_: m.expmap(t_, dim=1), (xb,), eps=1e-6, atol=1e-6, rtol=1e-4)
    xc = (torch.randn((3, 5), generator=g, device=dev, dtype=torch.float64) * 40.0).requires_grad_(True)   # clamp + project
    y = m.expmap(xc)
    (gx,) = torch.autograd.grad((y * torch.randn_like(y)).sum(), xc)
    assert torch.isfinite(gx).all() and float(gx.abs().max()) < 1e-1    # saturated: only the direction still matters
    mlr = HyperMLR(6, 5, c=0.8).to(dev)
    xe = m.expmap(torch.randn((2, 6, 3, 3), generator=g, device=dev) * 0.3, dim=1).detach().requires_grad_(True)
    P = mlr.P_MLR.detach().clone().requires_grad_(True)
    A = mlr.A_MLR.detach().clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b, c_: _HyperMLRFn.apply(a, b, c_, 0.8), (xe, P, A), eps=1e-6, atol=1e-6, rtol=1e-4)


def test_training_losses_match_reference_values_and_gradients(golden, dev):
    """NegativeLearningLoss and LocalConsistentLoss ('l1', 'kl'): forward value and gradient w.r.t. the input
    vs the reference's own modules under autograd (tests/golden/losses.npz), incl. the empty-selection case."""
    from halo_amd.core.loss import LocalConsistentLoss, NegativeLearningLoss
    d = golden("losses")
    label = t(d["label"], dev)
    for lt in ("l1", "kl"):
        x = t(d["x"], dev).requires_grad_(True)
        loss = LocalConsistentLoss(19, lt)(x, label)
        assert loss.dtype == torch.float32
        assert abs(loss.item() - float(d[f"lcl_{lt}__loss"][0])) < 2e-6 * max(1.0, abs(float(d[f"lcl_{lt}__loss"][0])))
        (gx,) = torch.autograd.grad(loss, x)
        want = d[f"lcl_{lt}__gx"]
        assert np.abs(gx.cpu().numpy() - want).max() < 2e-5 * np.abs(want).max() + 1e-9, lt
    with torch.no_grad():                                     # no gradient requested: no coefficient tensors
        assert abs(float(LocalConsistentLoss(19, "l1")(t(d["x"], dev), label)) - float(d["lcl_l1__loss"][0])) < 1e-6
    p = t(d["neg__p"], dev).requires_grad_(True)
    loss = NegativeLearningLoss(threshold=0.05)(p)
    assert abs(loss.item() - float(d["neg__loss"][0])) < 2e-6
    (gp,) = torch.autograd.grad(loss, p)
    assert np.abs(gp.cpu().numpy() - d["neg__gp"]).max() < 2e-5 * np.abs(d["neg__gp"]).max()
    x0 = t(d["empty__x"], dev).requires_grad_(True)
    l0 = LocalConsistentLoss(19, "l1")(x0, torch.zeros((1, 8, 8), dtype=torch.int64, device=dev))
    assert bool(torch.isnan(l0)) == bool(d["empty__loss_isnan"][0])
    (g0,) = torch.autograd.grad(l0, x0)
    assert float(g0.abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        LocalConsistentLoss(19, "l2")


def _torch_local_consistent(x, label, kl):
    """plain torch statement of LocalConsistentLoss (core/loss/local_consistent_loss.py:12-17, boundary.py:48-61, 94-103) on x's device"""
    import torch.nn.functional as F
    O = x.shape[1]
    p = torch.softmax(x, dim=1)
    mean = F.conv2d(F.pad(p, (1, 1, 1, 1), mode="replicate"), torch.full((O, 1, 3, 3), 1.0 / 9.0, device=x.device), groups=O)
    l = (p * torch.log(p / (mean + 1e-6) + 1e-6)).sum(dim=1) if kl else (p - mean).abs().sum(dim=1)
    k = torch.tensor([[[[-1., -1., -1.], [-1., 8., -1.], [-1., -1., -1.]]]], device=x.device)
    mask = (F.conv2d(label.float().unsqueeze(1), k, padding=1).long().squeeze(1) != 0) & (label != 255)
    return l[mask].mean()


def test_training_losses_ragged_shapes_and_many_classes(dev):
    """The loss kernels' other arms against a plain torch statement of the same modules under autograd: pixel counts that are not a
    multiple of four (scalar softmax), more than 20 classes (two-pass backward), one class, a single row / column, a label map
    with no boundary at all in one image, ignore labels; NegativeLearningLoss on unaligned views and lengths n % 4 != 0."""
    from halo_amd.core.loss import LocalConsistentLoss, NegativeLearningLoss
    g = torch.Generator(device="cpu").manual_seed(31)
    for (B, O, h, w) in ((2, 19, 24, 40), (1, 19, 7, 9), (2, 21, 6, 10), (1, 3, 1, 17), (1, 5, 13, 1), (2, 1, 5, 6), (1, 20, 16, 12)):
        x0 = (torch.randn((B, O, h, w), generator=g) * 2.0).to(dev)
        label = torch.randint(0, max(2, O), (B, (h + 3) // 4, (w + 3) // 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :h, :w].contiguous()
        label[torch.rand((B, h, w), generator=g) < 0.1] = 255
        if B > 1:
            label[1] = 3                                              # no boundary in the second image
        label = label.to(dev)
        for lt in ("l1", "kl"):
            xb = x0.clone().requires_grad_(True)
            lb = _torch_local_consistent(xb, label, lt == "kl")
            gb = None if bool(torch.isnan(lb)) else torch.autograd.grad(lb, xb)[0]
            for env in ({}, {"HALO_LCL_PLAIN": "1"}):          # the strip-walking forward (default) and the one-pixel-per-thread forward
                with _env_set(env):
                    xa = x0.clone().requires_grad_(True)
                    la = LocalConsistentLoss(O, lt)(xa, label)
                    if gb is None:
                        assert bool(torch.isnan(la))
                        continue
                    assert abs(la.item() - lb.item()) < 3e-6 * max(1.0, abs(lb.item())), (B, O, h, w, lt, env, la.item(), lb.item())
                    (ga,) = torch.autograd.grad(la, xa)
                    assert float((ga - gb).abs().max()) < 3e-5 * float(gb.abs().max()) + 1e-9, (B, O, h, w, lt, env)
    for n, off in ((1000, 0), (1003, 0), (1001, 1), (5, 3), (4096, 2)):
        buf = torch.rand(n + off + 8, generator=g).to(dev) * 0.2
        pa, pb = buf[off:off + n].clone().requires_grad_(True), buf[off:off + n].clone().requires_grad_(True)
        view = buf[off:off + n].detach().requires_grad_(True)       # an unaligned view of the buffer
        la = NegativeLearningLoss(0.05)(view if off else pa)
        mk = (pb < 0.05).detach()
        lb = (-(mk * torch.log(1 - pb + 1e-6))).sum() / mk.sum()
        assert abs(la.item() - lb.item()) < 3e-6, (n, off)
        (ga,) = torch.autograd.grad(la, view if off else pa)
        (gb,) = torch.autograd.grad(lb, pb)
        assert float((ga - gb).abs().max()) < 3e-5 * float(gb.abs().max()), (n, off)


@pytest.mark.parametrize("c", [0.5, 2.0])
def test_score_with_other_curvatures(dev, c):
    """cfg.MODEL.CURVATURE != 1 reaches the scorer through HyperMapper(c) (floating_region.py:68)."""
    from halo_amd.core.active.floating_region import FloatingRegionScore, score_maps
    from halo_amd.core.configs import cfg
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(int(c * 10))
    H, W, C, O = 40, 64, 10, 19
    logit = rng.standard_normal((1, O, H, W)).astype(np.float32)
    emb = ho.expmap((rng.standard_normal((1, C, H, W)) * 0.3).astype(np.float32), c, dim=1)
    for pur in ("radius", "hyper"):
        so, io, uo = ho.floating_region_score(logit, emb, "entropy", pur, True, None, size=3, purity_type=pur, K=20, c=c)
        s, i, u = score_maps(t(logit, dev), t(emb, dev), "entropy", pur, True, None, size=3, K=20, c=c)
        assert bits_equal(s[0].cpu().numpy(), so) and bits_equal(i[0].cpu().numpy(), io)
    old = cfg.MODEL.CURVATURE
    try:
        cfg.MODEL.CURVATURE = c
        frs = FloatingRegionScore(in_channels=O, size=3, purity_type="radius")
        assert frs.mapper.c == c
        s2, _, _ = frs(t(logit, dev), decoder_out=t(emb, dev), unc_type="entropy", pur_type="radius", normalize=True)
        so, _, _ = ho.floating_region_score(logit, emb, "entropy", "radius", True, None, size=3, purity_type="radius", c=c)
        assert bits_equal(s2.cpu().numpy(), so)
    finally:
        cfg.MODEL.CURVATURE = old


def test_calls_are_graph_capturable(dev):
    """include/halo_hip.h promises asynchronous enqueue only (no allocation, no host sync inside a call):
    capture score + select into a HIP graph through torch.cuda.CUDAGraph, replay it on new input values and
    compare with the eager result."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    B, H, W, C, O, n = 2, 64, 128, 8, 19, 12
    d0 = [_synthetic(H, W, C, O, 700 + b) for b in range(B)]
    d1 = [_synthetic(H, W, C, O, 800 + b) for b in range(B)]
    logit = torch.cat([t(x[0], dev) for x in d0]); emb = torch.cat([t(x[1], dev) for x in d0])
    gt = torch.stack([t(x[2], dev) for x in d0])
    score = torch.empty((B, H, W), dtype=torch.float64, device=dev)
    act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
    am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)

    def work():
        act.zero_(); sel.zero_(); am.fill_(255)
        score_maps(logit, emb, "entropy", "radius", True, None, size=3, active=act, want_maps=False, out=score)
        return greedy_select(score, n, 1, 5, act, sel, am, gt)

    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        work()                                                   # warm-up outside capture (workspaces get cached)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        picks_g, npk_g = work()
    logit.copy_(torch.cat([t(x[0], dev) for x in d1])); emb.copy_(torch.cat([t(x[1], dev) for x in d1]))
    gt.copy_(torch.stack([t(x[2], dev) for x in d1]))
    g.replay()
    torch.cuda.synchronize()
    got = (picks_g.clone(), npk_g.clone(), act.clone(), am.clone())
    picks_e, npk_e = work()
    torch.cuda.synchronize()
    assert torch.equal(got[0], picks_e) and torch.equal(got[1], npk_e) and torch.equal(got[2], act) and torch.equal(got[3], am)
    assert int(npk_e.min()) == n


# ------------------------------------------------------------------ round 2: selector methods, config gaps
def _select_vs_oracle(dev, s0, n, arad, mrad, methods=("auto", "serial"), prior=None, tag=""):
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    H, W = s0.shape
    rng = np.random.default_rng(H * 31 + W)
    gt = rng.integers(0, 19, (H, W)).astype(np.int64)
    prior = np.zeros((H, W), bool) if prior is None else prior
    act_o = prior.copy(); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
    so = s0.copy()
    _, _, _, _, po = ho.select_pixels_to_label(so, n, arad, mrad, act_o, sel_o, am_o, gt, True)
    for method in methods:
        s = t(s0, dev)[None].clone()
        act = t(prior, dev)[None].clone(); sel = torch.zeros_like(act)
        am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
        picks, npk = greedy_select(s, n, arad, mrad, act, sel, am, t(gt, dev)[None], method=method)
        k = int(npk[0])
        assert k == len(po), (tag, method, k, len(po))
        assert bits_equal(picks[0, :k].cpu().numpy(), po), (tag, method)
        assert bits_equal(s[0].cpu().numpy(), so), (tag, method)
        assert np.array_equal(act[0].cpu().numpy(), act_o) and np.array_equal(sel[0].cpu().numpy(), sel_o), (tag, method)
        assert np.array_equal(am[0].cpu().numpy(), am_o), (tag, method)
    return po


def test_binned_method_forced_and_geometries_it_declines(dev):
    """method='binned' must do the whole job itself where it applies and say so where it does not."""
    from halo_amd._lib import HaloUnsupported
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(41)
    s0 = ho.bilinear(rng.standard_normal((1, 40, 64)), (160, 256))[0]
    for mrad in (1, 3, 5, 14):
        _select_vs_oracle(dev, s0, 70, 1, mrad, methods=("binned",), tag=f"mrad{mrad}")
    _select_vs_oracle(dev, s0.astype(np.float32), 70, 2, 5, methods=("binned",), tag="f32")
    z = torch.zeros((1, 160, 256), device=dev)
    with pytest.raises(HaloUnsupported):                                  # mask radius above 14: serial kernel only
        greedy_select(t(s0, dev)[None].clone(), 5, 1, 17, z.bool(), z.bool(), z.long(), z.long(), method="binned")
    big = torch.zeros((1, 1024, 2048), device=dev)
    with pytest.raises(HaloUnsupported):                                  # radius 1 at 1024x2048: the pick grid exceeds LDS
        greedy_select(big.double(), 5, 1, 1, big.bool(), big.bool(), big.long(), big.long(), method="binned")


def test_binned_sweep_hand_over_cases_at_scale(dev):
    """512x1024 maps where the sweep must hand (part of) the image to the serial kernel -- plateaus of exact ties,
    NaN, +inf, constant -- or stop early because the pickable pixels run out; and maps it finishes alone."""
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(43)
    H, W, n = 512, 1024, 500
    smooth = ho.bilinear(rng.standard_normal((1, H // 4, W // 4)), (H, W))[0]
    cases = {"smooth": smooth, "noise_f32": rng.standard_normal((H, W)).astype(np.float32),
             "plateaus": np.round(smooth * 4) / 4, "half_masked": np.where(rng.random((H, W)) < 0.5, -np.inf, smooth),
             "nan": smooth.copy(), "posinf": smooth.copy(), "constant": np.full((H, W), 0.5),
             "few_pickable": np.full((H, W), -np.inf), "skewed": np.exp(6.0 * smooth) * 1e-3}
    cases["nan"][100, 200] = np.nan
    cases["posinf"][17, 900] = np.inf
    cases["few_pickable"][40:60, 100:400] = smooth[40:60, 100:400]
    cases["skewed"][5, 5] = 1e6                                            # one outlier stretches the value range
    for tag, s0 in cases.items():
        _select_vs_oracle(dev, np.ascontiguousarray(s0), n, 1, 5, methods=("auto",), tag=tag)
    _select_vs_oracle(dev, smooth, n, 1, 3, methods=("auto",), tag="ripu radius")    # configs/gtav/ripu.yaml: MASK_RADIUS_K 3
    prior = rng.random((H, W)) < 0.1
    sm = smooth.copy(); sm[prior] = -np.inf
    _select_vs_oracle(dev, sm, n, 1, 5, methods=("auto",), prior=prior, tag="second round")


def test_batched_hand_over_more_images_than_resume_workgroups(dev):
    """Behind the sweep the serial kernel runs with two workgroups that walk the images: a batch of seven maps, five of
    which hand over (plateaus, NaN, +inf, constant, plateaus again) in between two the sweep finishes, must come out
    image by image as the oracle's."""
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(45)
    H, W, n, mrad = 256, 384, 300, 5
    smooth = ho.bilinear(rng.standard_normal((1, H // 4, W // 4)), (H, W))[0]
    maps = [smooth.copy(), np.round(smooth * 4) / 4, smooth.copy(), smooth.copy(), np.full((H, W), 0.25), np.round(smooth * 3) / 3,
            ho.bilinear(rng.standard_normal((1, H // 4, W // 4)), (H, W))[0]]
    maps[2][10, 20] = np.nan
    maps[3][200, 300] = np.inf
    B = len(maps)
    s0 = np.ascontiguousarray(np.stack(maps))
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    s = t(s0, dev).clone()
    act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
    am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    picks, npk = greedy_select(s, n, 1, mrad, act, sel, am, t(gt, dev), method="auto")
    for b in range(B):
        so = s0[b].copy()
        act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
        _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, mrad, act_o, sel_o, am_o, gt[b], True)
        k = int(npk[b])
        assert k == len(po) and bits_equal(picks[b, :k].cpu().numpy(), po), b
        assert bits_equal(s[b].cpu().numpy(), so), b
        assert np.array_equal(act[b].cpu().numpy(), act_o) and np.array_equal(sel[b].cpu().numpy(), sel_o), b
        assert np.array_equal(am[b].cpu().numpy(), am_o), b


def test_select_1536x2048_uses_a_tile_table_above_64KiB(dev):
    """1536x2048 needs ~74 KiB of dynamic LDS in the serial kernel (48 KiB at 1024x2048)."""
    rng = np.random.default_rng(47)
    s0 = rng.standard_normal((1536, 2048)).astype(np.float32)
    _select_vs_oracle(dev, s0, 400, 1, 5, methods=("serial", "auto"), tag="1536x2048")


def test_config5_shape_c512_o16_full_size_bit_exact_vs_oracle(dev):
    """BASELINE.json configs[4] workload: C=512, 16 classes (SYNTHIA) at 1024x2048, 2331 regions."""
    _run_vs_oracle(dev, 1024, 2048, 512, 16, 4321, "entropy", "radius", True, 2331, 5, methods=("auto",))


def test_full_size_default_hyper_and_ripu_branches_vs_oracle(dev):
    """The default purity ('hyper', defaults.py:69) and configs/gtav/ripu.yaml at 1024x2048 (C=64 keeps the host side light)."""
    _run_vs_oracle(dev, 1024, 2048, 64, 19, 99, "entropy", "hyper", True, 2331, 5, methods=("auto",))
    _run_vs_oracle(dev, 1024, 2048, 64, 19, 98, "entropy", "ripu", False, 2331, 3, methods=("auto",))


def test_region_selection_deeplab_v2_geometry_full_size(dev):
    """DeepLab-v2 hands over logits AND embedding at the input size (classifier.py:375-377): both 640x1280 -> 1024x2048
    inside the scorer (ratio 1.6 for both), files equal the oracle driver's."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(2027)
    H, W, C, O = 1024, 2048, 64, 19
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 15000, 30000, 40000, 50000], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs4_")
    emb80 = ho.expmap((rng.standard_normal((1, C, 80, 160)) * 0.1).astype(np.float32), 1.0, dim=1)
    bound = 1.0 / math.sqrt(C)
    logit80 = ho.hypermlr(emb80, rng.uniform(-bound, bound, (O, C)), rng.uniform(-bound, bound, (O, C)), 1.0).astype(np.float32)
    logit_lr, emb_lr = ho.bilinear(logit80, (640, 1280)), ho.bilinear(emb80, (640, 1280))      # the v2 head resizes both
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    act = np.zeros((H, W), bool)
    item = {"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, "m.png")],
            "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
            "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None], "selected": torch.zeros(1, H, W, dtype=torch.bool),
            "path_to_indicator": [os.path.join(tmp, "i.pth")], "name": ["img"]}
    RegionSelection(cfg, _Fake(), _Fake([(t(logit_lr, dev), t(emb_lr, dev))]), [item], 1)
    (mask, a_o, s_o, picks), = ho.region_selection(cfg, [dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act,
                                                            selected=np.zeros((H, W), bool), origin_mask=np.full((H, W), 255, np.int64))],
                                                   lowres_mode=_lr_mode())
    assert len(picks) == 2331
    ind = torch.load(os.path.join(tmp, "i.pth"))
    assert np.array_equal(np.array(Image.open(os.path.join(tmp, "m.png")), dtype=np.uint8), mask)
    assert np.array_equal(ind["active"].numpy(), a_o) and np.array_equal(ind["selected"].numpy(), s_o)


def test_narrow_maps_through_the_lowres_scorer(dev):
    """a 64x16-tiled kernel writes more min/max partials than 128-pixel blocks on narrow maps."""
    from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    rng = np.random.default_rng(51)
    for (H, W) in ((4096, 4), (3000, 1), (2, 5000)):
        lg = t(rng.standard_normal((1, 19, max(1, H // 4), max(1, W // 2))).astype(np.float32), dev)
        em = t((rng.standard_normal((1, 6, max(1, H // 8), max(1, W // 2))) * 0.2), dev)
        a = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="exact")
        b = score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(em, (H, W)), "entropy", "radius", True, None, size=3)
        for x, y in zip(a, b):
            assert bits_equal(x.cpu().numpy(), y.cpu().numpy()), (H, W)
        g = score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="gram")
        for x, y in zip(a, g):
            assert max_abs_diff(x.cpu().numpy(), y.cpu().numpy()) < 1e-12, (H, W)


def test_region_selection_reads_the_curvature_of_the_cfg_it_is_given(dev):
    """RegionSelection(cfg, ...) with cfg.MODEL.CURVATURE != 1 and NO halo_amd.core.configs.use()."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from halo_amd.core.configs import cfg as standin
    from oracle import halo_oracle as ho
    assert standin.MODEL.CURVATURE == 1.0
    rng = np.random.default_rng(53)
    H, W, C, O, c = 48, 80, 8, 19, 0.5
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=c),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs5_")
    emb_lr = ho.expmap((rng.standard_normal((1, C, 12, 20)) * 0.4).astype(np.float32), c, dim=1)
    logit_lr = rng.standard_normal((1, O, 24, 40)).astype(np.float32)
    gt = rng.integers(0, O, (H, W)).astype(np.int64)
    item = {"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, "m.png")],
            "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
            "size": torch.tensor([[H, W]]), "active": torch.zeros(1, H, W, dtype=torch.bool), "selected": torch.zeros(1, H, W, dtype=torch.bool),
            "path_to_indicator": [os.path.join(tmp, "i.pth")], "name": ["img"]}
    tables = RegionSelection(cfg, _Fake(), _Fake([(t(logit_lr, dev), t(emb_lr, dev))]), [item], 1, return_tables=True)
    im = dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=np.zeros((H, W), bool), selected=np.zeros((H, W), bool),
              origin_mask=np.full((H, W), 255, np.int64))
    (mask, a_o, s_o, picks), = ho.region_selection(cfg, [im], lowres_mode=_lr_mode())
    (mask1, _, _, picks1), = ho.region_selection(cfg, [im], c=1.0, lowres_mode=_lr_mode())
    assert not np.array_equal(picks, picks1), "the test inputs must tell the two curvatures apart"
    assert np.array_equal(np.array(Image.open(os.path.join(tmp, "m.png")), dtype=np.uint8), mask)
    assert len(tables) == 1 and tables[0][1] == len(picks) and bits_equal(tables[0][0][:len(picks)].cpu().numpy(), picks)
    with pytest.warns(RuntimeWarning, match="VIZ_MASK"):
        cfg.ACTIVE.VIZ_MASK = True
        RegionSelection(cfg, _Fake(), _Fake([(t(logit_lr, dev), t(emb_lr, dev))]), [item], 2)


def _run_script(args, timeout=900, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_bench_under_torchrun_world1_uses_rccl(dev):
    """The multi-GPU launch path on the one GPU there is: torch.distributed.run, world 1, backend nccl (= RCCL):
    process-group init on the device, the per-step all_gather_into_tensor of the pick tables, the MAX all-reduce."""
    import json
    import socket
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = _run_script(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--height", "64", "--width", "128",
                     "--channels", "16", "--steps", "3", "--warmup", "1", "--batch", "4", "--ring", "8", "--cpu-images", "0"],
                    env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "RCCL all-gather" in d["config"]["sharding"] and d["pipeline_tables_consistent"] is True


def test_bench_self_spawn_refuses_more_gpus_than_visible(dev):
    """`python bench.py --gpus N` without a launcher starts its own ranks BEFORE touching the GPU; asking for more
    devices than are visible must fail fast with a clear message (on an 8-GPU node the same call runs 8 ranks)."""
    from conftest import ROOT
    n = torch.cuda.device_count() + 1
    r = _run_script([os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], timeout=300)
    assert r.returncode != 0 and "ROCm device" in (r.stdout + r.stderr)


def test_bench_self_spawn_launches_its_ranks_and_relays_one_line(dev):
    """The code path of `python bench.py --gpus N` (N > 1) on the one GPU there is: HALO_BENCH_SPAWN forces the built-in
    launcher at N = 1 -- a torch.distributed.run CHILD started before the parent touches the GPU, RCCL in the child,
    exactly one JSON line relayed, the child's exit code returned."""
    import json
    from conftest import ROOT
    r = _run_script([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--height", "64", "--width", "128", "--channels", "16", "--steps", "3",
                     "--warmup", "1", "--batch", "4", "--ring", "8", "--cpu-images", "0"], env={"HALO_BENCH_SPAWN": "1"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "RCCL all-gather" in d["config"]["sharding"]


def test_bench_pool_images_and_branches_on_a_small_shape(dev):
    import json
    from conftest import ROOT
    for extra in (["--pool-images", "22"], ["--branch", "ripu"], ["--branch", "hyper"]):
        r = _run_script([os.path.join(ROOT, "bench.py"), "--height", "64", "--width", "128", "--channels", "16", "--steps", "3",
                         "--warmup", "1", "--batch", "4", "--ring", "8", "--cpu-images", "1"] + extra)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["parity_vs_cpu"] is True and d["pipeline_tables_consistent"] is True, extra
        if extra[0] == "--pool-images":
            assert d["config"]["image_evaluations"] == 22 and d["steps"] == 6


def test_head_kernel_variants_agree_bitwise(dev):
    """Round-2 kernels of the head tail against the round-1 kernels they replace (kept behind A/B switches) and the
    oracle: expmap0+project through an LDS tile (single read, hoisted exact division) vs the two-pass plane walk;
    bilinear with LDS-staged taps vs global gathers vs one element per thread."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, bilinear_align_corners
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(61)
    for c in (1.0, 0.6):
        m = HyperMapper(c=c)
        for (B, C, h, w) in ((1, 256, 32, 64), (2, 64, 40, 72), (1, 8, 16, 32), (1, 512, 8, 16), (1, 5, 12, 20), (3, 19, 4, 4), (1, 64, 2, 6)):
            z = (rng.standard_normal((B, C, h, w)) * 0.1).astype(np.float32)
            z[0, :, 0, 0] = 0.0                                      # norm clamp
            z[0, :, 1, 1] *= 400.0                                   # tanh clamp + projection
            z[0, :, 1, 2] *= 30.0                                    # near the ball's boundary
            if h > 2:
                z[-1, :, 2, 3] = 3.0e18                              # huge finite values
            a = m.expmap(t(z, dev), dim=1).cpu().numpy()
            b = _with_env({"HALO_EXPMAP_PLANES": "1"}, lambda: m.expmap(t(z, dev), dim=1).cpu().numpy())
            assert bits_equal(a, b), (c, C, h, w)
            assert max_abs_diff(a, ho.expmap(z, c, dim=1)) < 1e-14
            n = np.sqrt((a * a).sum(axis=1))
            assert n.max() <= (1 - 1e-5) / math.sqrt(c) * (1 + 1e-12)
    for dt in (np.float64, np.float32):
        for (planes, hin, hout) in ((5, (16, 32), (64, 128)), (3, (40, 80), (64, 128)), (2, (10, 20), (64, 128)), (4, (23, 37), (50, 77)),
                                    (1, (64, 128), (64, 128)), (7, (5, 7), (96, 130)), (2, (64, 96), (16, 24)), (9, (33, 65), (129, 1030))):
            src = rng.standard_normal((1, planes) + hin).astype(dt)
            a = bilinear_align_corners(t(src, dev), hout).cpu().numpy()
            b = _with_env({"HALO_BILINEAR_ROWS": "1"}, lambda: bilinear_align_corners(t(src, dev), hout).cpu().numpy())
            c_ = _with_env({"HALO_BILINEAR_FLAT": "1"}, lambda: bilinear_align_corners(t(src, dev), hout).cpu().numpy())
            assert bits_equal(a, b) and bits_equal(a, c_), (dt, planes, hin, hout)
            assert bits_equal(a, ho.bilinear(src, hout)), (dt, planes, hin, hout)


def test_region_impurity_lds_tile_equals_generic_kernel(dev):
    """The 3x3 sliding-window histogram on an LDS label tile vs the generic global re-scan kernel (A/B switch) and
    the oracle, for label maps with many / few distinct classes, odd sizes, 1-pixel-wide maps."""
    from halo_amd.core.active.floating_region import FloatingRegionScore
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(71)
    for (H, W, K) in ((64, 128, 19), (37, 53, 100), (1, 70, 19), (90, 1, 7), (16, 64, 2), (130, 67, 19)):
        f = FloatingRegionScore(in_channels=K, size=3, purity_type="ripu")
        for kind in ("noise", "blobs"):
            lab = rng.integers(0, K, (H, W))
            if kind == "blobs":
                lab = (ho.bilinear(rng.standard_normal((1, max(1, H // 8) + 1, max(1, W // 8) + 1)), (H, W))[0] * 2).astype(np.int64) % K
            lab = lab.astype(np.int64)
            imp, cnt = f.compute_region_impurity(t(lab, dev), K)
            imp_g, cnt_g = _with_env({"HALO_IMPURITY_GENERIC": "1"}, lambda: f.compute_region_impurity(t(lab, dev), K))
            assert bits_equal(imp.cpu().numpy(), imp_g.cpu().numpy()) and bits_equal(cnt.cpu().numpy(), cnt_g.cpu().numpy()), (H, W, K, kind)
            want_i, want_c = ho.region_impurity(lab, K, 3)
            assert bits_equal(imp.cpu().numpy(), want_i) and np.array_equal(cnt.cpu().numpy(), want_c), (H, W, K, kind)


def test_v2_head_class_end_to_end_against_reference_vectors(golden, dev):
    """The drop-in head class (classifier.py:335-379 interface: dict in, (logits, embedding) out, both resized for
    DeepLab-v2) with its dilated branches set to pass the golden latent through: outputs equal the reference's vectors.
    Also: the patched forward on a reference-SHAPED head object (foreign mapper / conv_seg classes) shares parameters."""
    import torch.nn as nn
    from halo_amd.core.models.classifier import ASPP_Classifier_V2_Hyper, _tail_modules, v2_hyper_forward
    from halo_amd.core.utils.hyperbolic import HyperMLR
    d = golden("case_b_64x128_c16_o19")
    H, W, C, O = (int(v) for v in d["meta_HWCO"])
    head = ASPP_Classifier_V2_Hyper(C, [6, 12], [6, 12], O, C).to(dev).eval()
    assert sorted(head.state_dict()) == ["conv2d_list.0.bias", "conv2d_list.0.weight", "conv2d_list.1.bias", "conv2d_list.1.weight",
                                         "conv_seg.A_MLR", "conv_seg.P_MLR"]
    with torch.no_grad():
        for m in head.conv2d_list:
            m.weight.zero_(); m.bias.zero_()
        for c in range(C):
            head.conv2d_list[0].weight[c, c, 1, 1] = 1.0          # branch 0 = identity, branch 1 = 0: embed input = x['out']
        head.conv_seg.P_MLR.copy_(t(d["P_MLR"], dev)); head.conv_seg.A_MLR.copy_(t(d["A_MLR"], dev))
        out, emb = head({"out": t(d["z"], dev)}, size=(H, W))
        out_lr, emb_lr = head({"out": t(d["z"], dev)})
    assert out.dtype == torch.float32 and emb.dtype == torch.float64 and out.shape == (1, O, H, W) and emb.shape == (1, C, H, W)
    assert max_abs_diff(out.cpu().numpy(), d["logit"]) < 1e-5 and max_abs_diff(emb.cpu().numpy(), d["embed"]) < 1e-14
    assert max_abs_diff(emb_lr.cpu().numpy(), d["embed_lr"]) < 1e-14 and np.abs(out_lr.cpu().numpy() - d["logit_lr"]).max() < 1e-5

    class RefMapper:                                   # what a reference-built head holds when its classes were imported earlier
        def __init__(self, c):
            self.c = c

    class RefMLR(nn.Module):
        def __init__(self, src):
            super().__init__()
            self.c, self.K, self.num_classes = src.c, src.K, src.num_classes
            self.P_MLR, self.A_MLR = src.P_MLR, src.A_MLR

    foreign = nn.Module()
    foreign.conv2d_list = head.conv2d_list
    foreign.mapper, foreign.conv_seg = RefMapper(1.0), RefMLR(head.conv_seg)
    with torch.no_grad():
        out2, emb2 = v2_hyper_forward(foreign, {"out": t(d["z"], dev)}, size=(H, W))
    assert torch.equal(out2, out) and torch.equal(emb2, emb)
    mapper, seg = _tail_modules(foreign)
    assert isinstance(seg, HyperMLR) and seg.P_MLR is foreign.conv_seg.P_MLR and "_halo_tail" not in dict(foreign.named_modules())
    assert sorted(foreign.state_dict()) == sorted(head.state_dict())
    # training mode: differentiable through the HIP backward kernels
    head.train()
    out_t, _ = head({"out": t(d["z"], dev).requires_grad_(True)}, size=(H, W))
    out_t.square().mean().backward()
    assert head.conv_seg.P_MLR.grad is not None and head.conv2d_list[0].weight.grad is not None


SHARDED_SCRIPT = r'''
import os, sys, types
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from PIL import Image
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
from halo_amd.core.active.floating_region import lowres_mode
from halo_amd.pool import region_selection_sharded
from oracle import halo_oracle as ho
from test_pool_gloo import _Pool, _cfg
root = {tmp!r}
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))


class Head(torch.nn.Module):            # "classifier": hands back the item's precomputed low-res head outputs
    def forward(self, x, size=None):
        return self.cur["logit_lr"][None].to(dev), self.cur["embed_lr"][None].to(dev)


class Loader:                            # a DataLoader-like object over the pool that tells the head which item is current
    def __init__(self, ds, head):
        self.dataset, self.head = ds, head


pool = _Pool(root, 5)
head = Head()


class Tap(torch.nn.Module):              # "feature extractor": records the batch so Head can return that image's outputs
    def forward(self, x):
        return x


# the driver iterates a DataLoader built by region_selection_sharded over the dataset; route the per-item head outputs
class DS(torch.utils.data.Dataset):
    def __len__(self):
        return len(pool)

    def __getitem__(self, i):
        it = dict(pool[i])
        head.cur = it
        return {{k: v for k, v in it.items() if k not in ("logit_lr", "embed_lr")}}


res = region_selection_sharded(_cfg(), Tap(), head, DS(), 1, loader_kwargs=dict(pin_memory=False))
cfg = _cfg()
want = ho.region_selection(cfg, [dict(logit_lr=it["logit_lr"][None].numpy(), embed_lr=it["embed_lr"][None].numpy(),
                                      origin_label=it["origin_label"].numpy(), origin_mask=it["origin_mask"].numpy(),
                                      active=it["active"].numpy(), selected=it["selected"].numpy()) for it in pool.items],
                           lowres_mode=lowres_mode(None))       # the evaluation order RegionSelection runs by default
assert res["range"] == (0, 5) and res["tables"].is_cuda and res["tables"].shape[0] == 5
for i, (mask, act, sel, picks) in enumerate(want):
    assert np.array_equal(np.array(Image.open(os.path.join(root, "m%d.png" % i))), mask), i
    ind = torch.load(os.path.join(root, "i%d.pth" % i))
    assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), i
    k = int(res["counts"][i])
    assert k == len(picks) and np.array_equal(res["tables"][i, :k].cpu().numpy().view(np.int64), np.ascontiguousarray(picks).view(np.int64)), i
# the opt-in pool-wide budget with the real driver: score + select without writing, ONE all-gather, files from the kept prefixes
import copy
for f in os.listdir(root):
    if f.endswith((".png", ".pth")):
        os.remove(os.path.join(root, f))
G = 8
resg = region_selection_sharded(_cfg(), Tap(), head, DS(), 1, loader_kwargs=dict(pin_memory=False), global_budget=G)
kept = resg["kept"].cpu().numpy()
assert int(kept.sum()) == G and torch.equal(resg["tables"], res["tables"]) and kept.max() > kept.min()
for i, it in enumerate(pool.items):
    H, W = it["origin_label"].shape
    cfg_i = copy.deepcopy(_cfg())
    cfg_i.ACTIVE.SELECT_ITER = [0]
    cfg_i.ACTIVE.BUDGET = max(0.0, (int(kept[i]) - 0.5) * 9.0 / (H * W))
    (mask, act, sel, picks), = ho.region_selection(cfg_i, [dict(logit_lr=it["logit_lr"][None].numpy(), embed_lr=it["embed_lr"][None].numpy(),
                                                                 origin_label=it["origin_label"].numpy(), origin_mask=it["origin_mask"].numpy(),
                                                                 active=it["active"].numpy(), selected=it["selected"].numpy())],
                                                   lowres_mode=lowres_mode(None))
    assert len(picks) == kept[i], i
    assert np.array_equal(np.array(Image.open(os.path.join(root, "m%d.png" % i))), mask), i
    ind = torch.load(os.path.join(root, "i%d.pth" % i))
    assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), i
dist.destroy_process_group()
print("sharded ok")
'''


def test_region_selection_sharded_hip_driver_under_torchrun_world1(dev, tmp_path):
    """halo_amd.pool.region_selection_sharded with the REAL driver (HIP RegionSelection, return_tables) and RCCL: one rank
    under torch.distributed.run; files and the gathered (packed, all-gathered, unpacked) tables equal the oracle's."""
    import socket
    from conftest import ROOT
    script = tmp_path / "sharded.py"
    script.write_text(SHARDED_SCRIPT.format(root=ROOT, tmp=str(tmp_path)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = _run_script(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), str(script)], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_softmax_shared_reciprocal_path_and_plain_division_path_bitwise(dev):
    """softmax's p/s runs through a reciprocal shared by the classes of a pixel when every exp(x - max) is far from the
    denormal range, and through the plain division otherwise: both must give the oracle's (IEEE) bits.  Logit gaps from
    0 to 150 exercise both paths, mixed within and across waves, for the fused, stand-alone and generic-class kernels."""
    from halo_amd.core.active.floating_region import score_maps
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(81)
    H, W, C = 48, 128, 6
    emb = (rng.standard_normal((1, C, H, W)) * 0.2)
    for O in (19, 16, 7):
        logit = rng.standard_normal((1, O, H, W)).astype(np.float32)
        logit[:, :, :, 32:64] *= 30.0                              # gaps around 60-100: some pixels on each side of the switch
        logit[:, :, :, 64:96] *= 80.0                              # far beyond: denormal / zero probabilities
        logit[:, 0, 10, 100:128] = 1e30                            # overflowing differences
        logit[:, 1, 11, 100:128] = -np.inf
        logit[:, 2, 12, 96:128] = np.nan                           # a NaN that is not class 0 hides from the running max / min
        logit[:, 0, 13, 64:80] = np.nan
        logit[:, 3, 14, 0:8] = np.inf
        gt = rng.integers(0, O, (H, W)).astype(np.int64)
        for unc, pur in (("entropy", "radius"), ("entropy", "ripu"), ("oracle_acc", "oracle_ripu")):
            so, io, uo = ho.floating_region_score(logit, emb, unc, pur, False, gt, size=3, purity_type=pur)
            s, i, u = score_maps(t(logit, dev), t(emb, dev), unc, pur, False, t(gt, dev)[None], size=3)
            assert bits_equal(u[0].cpu().numpy(), uo), (O, unc, pur)
            assert bits_equal(i[0].cpu().numpy(), io) and bits_equal(s[0].cpu().numpy(), so), (O, unc, pur)


def test_binned_selection_repeats_identically(dev):
    """The sweep's inner order is not fixed (atomics place candidates inside a bin in arrival order, a bin's picks are
    committed by parallel lanes): 25 repetitions per map -- noise maps, where neighbours within one bin do conflict, smooth
    maps, every mask radius the sweep serves on these sizes -- must all give the oracle's picks and masks."""
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(91)
    for (H, W, mrad, n, kind, dt) in ((96, 160, 1, 300, "noise", np.float64), (96, 160, 2, 200, "noise", np.float32),
                                      (128, 192, 3, 150, "smooth", np.float64), (200, 320, 5, 120, "noise", np.float32),
                                      (200, 320, 9, 60, "smooth", np.float64), (64, 64, 14, 9, "noise", np.float64)):
        s0 = rng.standard_normal((H, W)) if kind == "noise" else ho.bilinear(rng.standard_normal((1, H // 4, W // 4)), (H, W))[0]
        s0 = np.ascontiguousarray(s0.astype(dt))
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        act_o = np.zeros((H, W), bool); sel_o = np.zeros((H, W), bool); am_o = np.full((H, W), 255, np.int64)
        so = s0.copy()
        _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, mrad, act_o, sel_o, am_o, gt, True)
        gtd = t(gt, dev)[None]
        for rep in range(25):
            s = t(s0, dev)[None].clone()
            act = torch.zeros((1, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
            am = torch.full((1, H, W), 255, dtype=torch.int64, device=dev)
            picks, npk = greedy_select(s, n, 1, mrad, act, sel, am, gtd, method="binned")
            k = int(npk[0])
            assert k == len(po) and bits_equal(picks[0, :k].cpu().numpy(), po), (H, W, mrad, kind, rep)
            assert np.array_equal(act[0].cpu().numpy(), act_o) and np.array_equal(am[0].cpu().numpy(), am_o), (H, W, mrad, kind, rep)
            assert bits_equal(s[0].cpu().numpy(), so), (H, W, mrad, kind, rep)


def test_randomised_differential_run_vs_oracle(dev):
    """tests/fuzz_parity.py as a regression: 200 random configurations (shapes, branches, window sizes, padding modes, dtypes,
    low-res geometries and modes, selection parameters, NaN / inf logits) -- HIP == oracle bit for bit in maps, picks, masks.
    (Round 4: the first run found that the pick table reports +0 for a -0.0 score and the canonical NaN for any NaN while the
    oracle kept the raw bits; the table's convention is now written down and the oracle follows it.)"""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "200", "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "HIP == oracle bit for bit" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,shape", [("float64", (1, 12, 160, 320, 1024, 2048)), ("float32", (1, 19, 640, 1280, 1024, 2048)),
                                         ("float64", (2, 5, 23, 37, 147, 231)), ("float32", (2, 19, 37, 53, 101, 203))])
def test_device_resize_is_torchs_cpu_interpolate_bit_for_bit(dev, dtype, shape):
    """The HIP resize against the reference's own call (build.py:123-135), F.interpolate(mode='bilinear', align_corners=True)
    evaluated by torch on this box's CPU: the same bits, at the head-output -> label-size shapes of the path and at ragged
    ones.  (Round 4: the contract's bilinear is ATen's order; the CPU oracle carries the same statement,
    tests/test_oracle_golden.py::test_bilinear_is_torchs_cpu_kernel_bit_for_bit.)"""
    import torch.nn.functional as F
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    if " fma" not in open("/proc/cpuinfo").read():
        pytest.skip("no FMA on this host: ATen's CPU kernel rounds every product")
    B, C, h, w, H, W = shape
    x = torch.randn((B, C, h, w), dtype=getattr(torch, dtype), generator=torch.Generator().manual_seed(11))
    want = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)
    got = bilinear_align_corners(x.to(dev), (H, W)).cpu()
    assert got.dtype == want.dtype and torch.equal(got, want)


@pytest.mark.parametrize("C,hf,wf,H,W", [(40, 160, 320, 1024, 2048), (22, 256, 512, 1024, 2048), (17, 64, 128, 256, 512), (9, 22, 44, 64, 128), (12, 40, 64, 150, 250),
                                          (8, 16, 32, 100, 136), (5, 30, 50, 95, 190), (33, 12, 24, 80, 192)])
def test_lowres_exact_mode_staging_variants_agree_bitwise(dev, C, hf, wf, H, W):
    """The exact low-res embedding pass has four statements of one arithmetic: LDS-DMA with a compile-time window geometry and 8
    pixels per lane (default where a source row is at least 3 output rows tall; row codes of HALO_LR_CODES8, anything else takes
    the in-kernel generic loop), the same with 4 pixels per lane (HALO_LR_PPT4=1), runtime strides (HALO_LR_NOFIXED=1) and
    register staging (HALO_LR_NODMA=1).  Same bits from all of them -- at x6.4 and x4 at FULL size with a partial last chunk (the
    race of round 4 -- NOTES.md -- only fired with a full chip of waves in flight), at a scale of exactly 1/3 (three-row runs,
    three steps inside a lane's 8 pixels), at heights that are no multiple of 32 and widths that are no multiple of 64 -- and
    the same bits as upsample-then-score."""
    from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
    from halo_amd.core.utils.hyperbolic import bilinear_align_corners
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(C * 13 + hf)
    emb_lr = ho.expmap((rng.standard_normal((2, C, hf, wf)) * 0.2).astype(np.float32), 1.0, dim=1)
    lg, em = t(rng.standard_normal((2, 19, hf, wf)).astype(np.float32), dev), t(emb_lr, dev)
    want = score_maps(bilinear_align_corners(lg, (H, W)), bilinear_align_corners(em, (H, W)), "entropy", "radius", True, None, size=3)
    for env in ({}, {"HALO_LR_PPT4": "1"}, {"HALO_LR_NOFIXED": "1"}, {"HALO_LR_NODMA": "1"}):
        got = _with_env(env, lambda: score_maps_lowres(lg, em, (H, W), "entropy", "radius", True, None, ksize=3, mode="exact"))
        torch.cuda.synchronize()
        for x, y in zip(got, want):
            assert bits_equal(x.cpu().numpy(), y.cpu().numpy()), env


def test_the_visualize_wrong_call_pattern(dev, monkeypatch):
    """The path's second caller (core/utils/visualize.py:20-34): `FloatingRegionScore(in_channels, size).cuda()` -- the
    purity window taken from the global cfg -- and three forwards on the head's outputs: ('entropy', 'ripu'), ('hyperbolic',
    'ripu'), ('certainty', 'ripu'), normalised, with a decoder_out the ripu purity never reads.  Under a ripu configuration
    (configs/gtav/ripu.yaml) the two vestigial uncertainty names are zero maps (floating_region.py:83-92), so their normalised
    uncertainty is 0/0 = NaN everywhere: same maps as the oracle, bit for bit.  Under the default cfg (PURITY 'hyper': a 100-bin
    purity window) the reference's depthwise conv rejects the 19-class one-hot; the mirror raises too."""
    from halo_amd.core.active import floating_region as fr
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(5)
    H, W, O, C = 40, 72, 19, 8
    logit = rng.standard_normal((1, O, H, W)).astype(np.float32) * 3
    emb = ho.expmap((rng.standard_normal((1, C, H, W)) * 0.2).astype(np.float32), 1.0, dim=1)
    monkeypatch.setattr(fr.cfg.ACTIVE, "PURITY", "hyper", raising=False)
    with pytest.raises(RuntimeError):
        fr.FloatingRegionScore(in_channels=O, size=3).cuda()(t(logit, dev), decoder_out=t(emb, dev), unc_type="entropy", pur_type="ripu", normalize=True)
    monkeypatch.setattr(fr.cfg.ACTIVE, "PURITY", "ripu", raising=False)
    frs = fr.FloatingRegionScore(in_channels=O, size=3).cuda()
    for unc in ("entropy", "hyperbolic", "certainty"):
        s, i, u = frs(t(logit, dev), decoder_out=t(emb, dev), unc_type=unc, pur_type="ripu", normalize=True)
        so, io, uo = ho.floating_region_score(logit, emb, unc, "ripu", True, None, size=3)
        assert s.shape == (H, W) and s.dtype == torch.float32
        assert bits_equal(s.cpu().numpy(), so) and bits_equal(i.cpu().numpy(), io) and bits_equal(u.cpu().numpy(), uo), unc
        if unc != "entropy":
            assert np.isnan(u.cpu().numpy()).all()


# ------------------------------------------------------------------ the launch shape bench.py times: B = 16 full-size images in ONE call
def _device_pool(dev, B, H, W, C, O, seed, fdtype=torch.float64):
    """B distinct smooth full-size images (latent -> expmap -> HyperMLR -> x4 resize) produced on the device: inputs are inputs,
    whoever made them -- the checked images are copied to the host and handed to the oracle byte for byte."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners
    rng = np.random.default_rng(seed)
    h, w = H // 4, W // 4
    mapper = HyperMapper(c=1.0)
    torch.manual_seed(seed)
    mlr = HyperMLR(C, O, c=1.0).to(dev)
    feat = torch.empty((B, C, H, W), dtype=fdtype, device=dev)
    logit = torch.empty((B, O, H, W), dtype=torch.float32, device=dev)
    feat_lr = torch.empty((B, C, h, w), dtype=torch.float64, device=dev)
    logit_lr = torch.empty((B, O, h, w), dtype=torch.float32, device=dev)
    gt = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    with torch.no_grad():
        for b in range(B):
            z = t((rng.standard_normal((1, C, h, w)) * 0.1).astype(np.float32), dev)
            emb = mapper.expmap(z, dim=1)
            lg = mlr._hyper_logits(emb, out_dtype=torch.float32)
            feat_lr[b:b + 1], logit_lr[b:b + 1] = emb, lg
            logit[b:b + 1] = bilinear_align_corners(lg, (H, W))
            up = bilinear_align_corners(emb, (H, W))
            feat[b:b + 1] = up if fdtype == torch.float64 else up.float()
            del up
            g = rng.integers(0, O, (H, W)).astype(np.int64)
            g[rng.random((H, W)) < 0.05] = 255
            gt[b] = t(g, dev)
    torch.cuda.synchronize(dev)
    return feat, logit, gt, feat_lr, logit_lr


def _check_batch_images_vs_oracle(dev, maps, picks, npk, act, sel, am, sc_after, logit, feat, gt, images, n, mrad, tag):
    """images `images` of one batched call against the oracle, bit for bit: three maps, pick table, masks, mutated score"""
    from oracle import halo_oracle as ho
    H, W = gt.shape[-2:]
    for b in images:
        so, io, uo = ho.floating_region_score(logit[b:b + 1].cpu().numpy(), feat[b:b + 1].cpu().numpy(), "entropy", "radius", True,
                                              gt[b].cpu().numpy(), size=3, purity_type="radius")
        assert bits_equal(maps[2][b].cpu().numpy(), uo), (tag, b, "uncertainty")
        assert bits_equal(maps[1][b].cpu().numpy(), io), (tag, b, "impurity")
        assert bits_equal(maps[0][b].cpu().numpy(), so), (tag, b, "score")
        a_o = np.zeros((H, W), bool); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
        _, _, _, _, pk_o = ho.select_pixels_to_label(so, n, 1, mrad, a_o, s_o, m_o, gt[b].cpu().numpy(), True)
        k = int(npk[b])
        assert k == len(pk_o) == n, (tag, b)
        assert bits_equal(picks[b, :k].cpu().numpy(), pk_o), (tag, b, "picks")
        assert np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(sel[b].cpu().numpy(), s_o), (tag, b, "indicators")
        assert np.array_equal(am[b].cpu().numpy(), m_o), (tag, b, "mask")
        assert bits_equal(sc_after[b].cpu().numpy(), so), (tag, b, "score after selection")     # so was mutated by the oracle's selection


@pytest.mark.parametrize("B,C,O,images", [(16, 256, 19, (0, 7, 8, 15)), (8, 512, 16, (0, 3, 4, 7))])
def test_headline_launch_shape_batched_full_size_vs_oracle(dev, B, C, O, images):
    """what bench.py times is ONE score_maps + ONE greedy_select call over 16 full-size images (1024x2048,
    C = 256 float64: 8.6 G feature elements, the only place where element offsets pass 2^32 -- image 8 starts exactly there, image
    7 ends there; at C = 512 image 4 does).  Images spread over the batch are checked against the oracle bit for bit: maps, the
    2331 picks in order, the masks and the mutated score map (floating_region.py:129-217, build.py:137-160)."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps
    H, W, n, mrad = 1024, 2048, 2331, 5
    feat, logit, gt, _, _ = _device_pool(dev, B, H, W, C, O, 500 + C)
    assert feat.numel() > (1 << 32)
    with torch.no_grad():
        maps = score_maps(logit, feat, "entropy", "radius", True, gt, size=3)
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        sc = maps[0].clone()
        picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, gt)
    torch.cuda.synchronize(dev)
    _check_batch_images_vs_oracle(dev, maps, picks, npk, act, sel, am, sc, logit, feat, gt, images, n, mrad, "B=%d C=%d" % (B, C))


def test_headline_launch_shape_lowres_sources_batched_vs_oracle(dev):
    """The same batch through the RegionSelection boundary (N1): 16 x4 low-res head outputs in ONE score_maps_lowres call (default
    'exact' order) == the oracle's upsample-then-score, images 0 / 7 / 15, plus selection."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.core.active.floating_region import score_maps_lowres
    from oracle import halo_oracle as ho
    B, H, W, C, O, n, mrad = 16, 1024, 2048, 256, 19, 2331, 5
    rng = np.random.default_rng(77)
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
    torch.manual_seed(77)
    mlr = HyperMLR(C, O, c=1.0).to(dev)
    with torch.no_grad():
        z = t((rng.standard_normal((B, C, H // 4, W // 4)) * 0.1).astype(np.float32), dev)
        feat_lr = HyperMapper(c=1.0).expmap(z, dim=1)
        logit_lr = mlr._hyper_logits(feat_lr, out_dtype=torch.float32)
        gt = t(rng.integers(0, O, (B, H, W)).astype(np.int64), dev)
        maps = score_maps_lowres(logit_lr, feat_lr, (H, W), "entropy", "radius", True, gt, ksize=3, mode="exact")
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        sc = maps[0].clone()
        picks, npk = greedy_select(sc, n, 1, mrad, act, sel, am, gt)
    torch.cuda.synchronize(dev)
    for b in (0, 7, 15):
        lg = ho.bilinear(logit_lr[b:b + 1].cpu().numpy(), (H, W))
        em = ho.bilinear(feat_lr[b:b + 1].cpu().numpy(), (H, W))
        so, io, uo = ho.floating_region_score(lg, em, "entropy", "radius", True, gt[b].cpu().numpy(), size=3, purity_type="radius")
        assert bits_equal(maps[0][b].cpu().numpy(), so) and bits_equal(maps[1][b].cpu().numpy(), io) and bits_equal(maps[2][b].cpu().numpy(), uo), b
        a_o = np.zeros((H, W), bool); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
        _, _, _, _, pk_o = ho.select_pixels_to_label(so, n, 1, mrad, a_o, s_o, m_o, gt[b].cpu().numpy(), True)
        assert int(npk[b]) == n == len(pk_o) and bits_equal(picks[b].cpu().numpy(), pk_o), b
        assert np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(am[b].cpu().numpy(), m_o), b


def test_sweep_hand_over_counters(dev):
    """halo_greedy_select_ex's cost counters: which images the value-binned sweep finished, which it handed to the serial kernel,
    why, and after how many of its own picks -- with the picks identical to the oracle's either way.  A plateau of exact ties no
    longer hands the image over (round 4: any full bin did, from pick 0): below the values the picks need it is never reached, and
    reached it is walked in position order through the map itself."""
    from halo_amd import _lib
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(21)
    H, W, n, mrad = 256, 512, 150, 5
    smooth = ho.bilinear(rng.standard_normal((1, H // 4, W // 4)), (H, W))[0] + 1e-4 * rng.standard_normal((H, W))
    floor, cap = np.quantile(smooth, 0.5), np.quantile(smooth, 0.997)
    low = smooth.copy(); low[low < floor] = floor                                   # half the map one value, never reached
    mid = smooth.copy(); mid[(mid > floor) & (mid < cap)] = 0.5 * (floor + cap)      # a plateau the picks must cross
    near = smooth.copy()                                                             # near-ties: distinct keys in one sub-slice of the range
    m_ = (near > floor) & (near < cap)
    near[m_] = 0.5 * (floor + cap) + 1e-13 * rng.standard_normal(int(m_.sum()))
    top = smooth.copy(); top[smooth > np.quantile(smooth, 0.6)] = 7.0                # the plateau IS the top: every pick a tie-break by position
    nan = smooth.copy(); nan[5, 7] = np.nan
    const = np.full((H, W), 0.25)
    maps = [smooth, low, mid, near, top, nan, const]
    B = len(maps)
    s0 = np.ascontiguousarray(np.stack(maps))
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    # a plateau of exact ties reached by the walk is taken in position order by the sweep itself (round 5); only a full bin of MIXED
    # keys still hands over
    want_reason = ["done", "done", "done", "bin_overflow", "done", "bad_values", "bad_values"]
    for method in ("auto", "serial"):
        s = t(s0, dev).clone()
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        ho_t = torch.full((B, 2), -7, dtype=torch.int32, device=dev)
        picks, npk = greedy_select(s, n, 1, mrad, act, sel, am, t(gt, dev), method=method, handover=ho_t)
        hv = ho_t.cpu().numpy()
        for b in range(B):
            so = s0[b].copy()
            a_o = np.zeros((H, W), bool); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
            _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, mrad, a_o, s_o, m_o, gt[b], True)
            k = int(npk[b])
            assert k == len(po) and bits_equal(picks[b, :k].cpu().numpy(), po), (method, b)
            assert np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(am[b].cpu().numpy(), m_o), (method, b)
            reason = _lib.SWEEP_REASONS[int(hv[b, 0])]
            if method == "serial":
                assert reason == "not_run" and hv[b, 1] == 0, b
            else:
                assert reason == want_reason[b], (b, reason, hv[b])
                if reason == "done":
                    assert hv[b, 1] == k
                if reason == "bin_overflow":
                    assert 0 < hv[b, 1] < k, hv[b]          # everything above the plateau came from the sweep
                if reason == "bad_values":
                    assert hv[b, 1] == 0


def test_region_selection_replayed_launch_groups_survive_workspace_growth(dev):
    """RegionSelection replays a (slot, shape)'s launch group from a HIP graph from its third use on.  The recording must own what it
    points to: here a pool of ONE label size runs three rounds through one slot (eager, eager + record, replay), then a batch-3 round of
    a LARGER size grows the stream's cached scratch buffers (round 5: the recording kept the old pointer -- a memory access fault),
    the workspaces are dropped altogether, and the first pool runs again through the replayed group.  Every round's files equal the
    oracle's; HALO_RS_GRAPH=0 gives the same files eagerly."""
    import halo_amd
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(131)
    cfg = types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    tmp = tempfile.mkdtemp(prefix="halo_rs_graph_")

    def pool(n, H, W, tag, nb=1):
        items, outs, oin = [], [], []
        for i in range(n):
            emb_lr = ho.expmap((rng.standard_normal((1, 8, H // 4, W // 4)) * 0.2).astype(np.float32), 1.0, dim=1)
            logit_lr = rng.standard_normal((1, 19, H // 2, W // 2)).astype(np.float32)
            gt = rng.integers(0, 19, (H, W)).astype(np.int64)
            act = rng.random((H, W)) < 0.02
            items.append({"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"{tag}_m{i}.png")],
                          "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
                          "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None], "selected": torch.zeros(1, H, W, dtype=torch.bool),
                          "path_to_indicator": [os.path.join(tmp, f"{tag}_i{i}.pth")], "name": [f"{tag}{i}"]})
            outs.append((t(logit_lr, dev), t(emb_lr, dev)))
            oin.append(dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act, selected=np.zeros((H, W), bool),
                            origin_mask=np.full((H, W), 255, np.int64)))
        if nb > 1:        # the same pool through a loader of batch size nb
            bat = []
            for k in range(0, n, nb):
                grp = items[k:k + nb]
                bat.append({key: (torch.cat([g_[key] for g_ in grp]) if torch.is_tensor(grp[0][key]) else sum((g_[key] for g_ in grp), []))
                            for key in grp[0]})
            outs = [(torch.cat([outs[k + j][0] for j in range(nb)]), torch.cat([outs[k + j][1] for j in range(nb)])) for k in range(0, n, nb)]
            items = bat
        return items, outs, oin

    def check(tag, oin):
        want = ho.region_selection(cfg, oin, lowres_mode=_lr_mode())
        for i, (mask, act, sel, _) in enumerate(want):
            assert np.array_equal(np.array(Image.open(os.path.join(tmp, f"{tag}_m{i}.png")), dtype=np.uint8), mask), (tag, i)
            ind = torch.load(os.path.join(tmp, f"{tag}_i{i}.pth"))
            assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), (tag, i)
            os.remove(os.path.join(tmp, f"{tag}_m{i}.png")); os.remove(os.path.join(tmp, f"{tag}_i{i}.pth"))

    small = pool(5, 48, 96, "s")
    for rnd in range(3):                                              # five images through one slot: eager, record, replay x 3 -- three times
        RegionSelection(cfg, _Fake(), _Fake(small[1]), small[0], 1, in_flight=1, writer_threads=2)
        check("s", small[2])
    big = pool(6, 96, 160, "b", nb=3)                                 # larger maps, three per launch group: the cached scratch grows
    RegionSelection(cfg, _Fake(), _Fake(big[1]), big[0], 1, in_flight=1, writer_threads=2)
    check("b", big[2])
    halo_amd.release_workspaces()                                     # ... and is dropped
    RegionSelection(cfg, _Fake(), _Fake(small[1]), small[0], 1, in_flight=1, writer_threads=2)
    check("s", small[2])
    _with_env({"HALO_RS_GRAPH": "0"}, lambda: RegionSelection(cfg, _Fake(), _Fake(small[1]), small[0], 1, in_flight=1, writer_threads=2))
    check("s", small[2])


def test_region_selection_two_table_widths_through_one_slot_shape(dev):
    """One slot shape (b, H, W) served with TWO pick-table widths in one process (another ACTIVE.BUDGET): the replayed recording of
    the first width copies into ITS pinned table; the writer must read that one and not the buffer the other width's eager launch
    touched last (`buf.out_picks` was a mutable slot attribute the replay never updated -> silently wrong files).
    Alternates the two budgets over seven calls, so that both widths go eager -> recorded -> replayed and interleave."""
    from PIL import Image
    from halo_amd.core.active.build import RegionSelection
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(977)
    tmp = tempfile.mkdtemp(prefix="halo_rs_widths_")
    H, W, n = 48, 96, 4

    def cfg_for(budget):
        return types.SimpleNamespace(
            MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
            ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                         BUDGET=budget, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    items, outs, oin = [], [], []
    for i in range(n):
        emb_lr = ho.expmap((rng.standard_normal((1, 8, H // 4, W // 4)) * 0.2).astype(np.float32), 1.0, dim=1)
        logit_lr = rng.standard_normal((1, 19, H // 2, W // 2)).astype(np.float32)
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        act = rng.random((H, W)) < 0.02
        items.append({"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
                      "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
                      "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None], "selected": torch.zeros(1, H, W, dtype=torch.bool),
                      "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"w{i}"]})
        outs.append((t(logit_lr, dev), t(emb_lr, dev)))
        oin.append(dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act, selected=np.zeros((H, W), bool),
                        origin_mask=np.full((H, W), 255, np.int64)))
    want = {b: ho.region_selection(cfg_for(b), oin, lowres_mode=_lr_mode()) for b in (0.05, 0.15)}
    assert len(want[0.05][0][3]) != len(want[0.15][0][3])                 # two table widths indeed
    for call, budget in enumerate((0.05, 0.15, 0.05, 0.15, 0.05, 0.15, 0.05)):
        RegionSelection(cfg_for(budget), _Fake(), _Fake(outs), items, 1, in_flight=1, writer_threads=2)
        for i, (mask, act, sel, _) in enumerate(want[budget]):
            assert np.array_equal(np.array(Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8), mask), (call, budget, i)
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), (call, budget, i)
            os.remove(os.path.join(tmp, f"m{i}.png")); os.remove(os.path.join(tmp, f"i{i}.pth"))


def test_region_selection_survives_a_failed_recording_of_its_launch_group(dev, monkeypatch):
    """A HIP-graph recording that fails must cost nothing but the replay: the launch group is recorded on a stream of its own
    (build._capture_stream), so the slot's stream -- where this batch's eager run, its copies and its event live -- never enters
    capture mode.  (Round 6: one recording in ~1700 was invalidated by the runtime under GPU_MAX_HW_QUEUES=8; it had been made on
    the slot's own stream, which ROCm 7.2 then left in capture mode for good, and the round died in a writer thread's event wait.)
    Here the second batch's recording is broken on purpose -- a device synchronisation inside the capture -- and the five images
    still give the oracle's files, eagerly; afterwards another launch group of the same process records and replays normally."""
    from PIL import Image
    from halo_amd.core.active import build
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(4242)
    tmp = tempfile.mkdtemp(prefix="halo_rs_badcapture_")
    H, W, n = 40, 88, 5

    def cfg_for(budget):
        return types.SimpleNamespace(
            MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
            ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                         BUDGET=budget, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
    items, outs, oin = [], [], []
    for i in range(n):
        emb_lr = ho.expmap((rng.standard_normal((1, 8, H // 4, W // 4)) * 0.2).astype(np.float32), 1.0, dim=1)
        logit_lr = rng.standard_normal((1, 19, H // 2, W // 2)).astype(np.float32)
        gt = rng.integers(0, 19, (H, W)).astype(np.int64)
        act = rng.random((H, W)) < 0.02
        items.append({"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
                      "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64), "origin_label": torch.from_numpy(gt)[None],
                      "size": torch.tensor([[H, W]]), "active": torch.from_numpy(act)[None], "selected": torch.zeros(1, H, W, dtype=torch.bool),
                      "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"c{i}"]})
        outs.append((t(logit_lr, dev), t(emb_lr, dev)))
        oin.append(dict(logit_lr=logit_lr, embed_lr=emb_lr, origin_label=gt, active=act, selected=np.zeros((H, W), bool),
                        origin_mask=np.full((H, W), 255, np.int64)))

    def check(budget):
        for i, (mask, act, sel, _) in enumerate(ho.region_selection(cfg_for(budget), oin, lowres_mode=_lr_mode())):
            assert np.array_equal(np.array(Image.open(os.path.join(tmp, f"m{i}.png")), dtype=np.uint8), mask), (budget, i)
            ind = torch.load(os.path.join(tmp, f"i{i}.pth"))
            assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), (budget, i)
            os.remove(os.path.join(tmp, f"m{i}.png")); os.remove(os.path.join(tmp, f"i{i}.pth"))

    real = build.acquire_batch_lowres
    broke = []

    def breaks_a_capture(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            broke.append(1)
            torch.cuda.synchronize()                     # not permitted inside a capture: invalidates it and raises
        return real(*a, **k)
    monkeypatch.setattr(build, "acquire_batch_lowres", breaks_a_capture)
    with pytest.warns(RuntimeWarning, match="graph capture of the launch group failed"):
        build.RegionSelection(cfg_for(0.05), _Fake(), _Fake(outs), items, 1, in_flight=1, writer_threads=2)
    assert broke == [1]                                  # one recording was attempted, none after it failed
    check(0.05)
    build.RegionSelection(cfg_for(0.05), _Fake(), _Fake(outs), items, 1, in_flight=1, writer_threads=2)    # the same group again: eager
    assert broke == [1]
    check(0.05)
    monkeypatch.setattr(build, "acquire_batch_lowres", real)
    mine = lambda: [g for sl in build._SLOTS.values() for s_ in sl for k_, g in s_.graphs.items() if k_[:3] == (1, H, W)]
    assert not any(isinstance(g, build._SlotGraph) for g in mine())
    build.RegionSelection(cfg_for(0.15), _Fake(), _Fake(outs), items, 1, in_flight=1, writer_threads=2)    # another group: eager, record, replay x 3
    assert sum(isinstance(g, build._SlotGraph) for g in mine()) == 1
    check(0.15)


def test_more_handed_over_images_than_resume_workgroups(dev):
    """The serial kernel behind the sweep runs one workgroup per image up to 64; 70 maps that ALL hand over (NaN, constant maps) make its workgroups walk more than one image each."""
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(5)
    B, H, W, n, mrad = 70, 48, 64, 6, 3
    maps = []
    for b in range(B):
        m = rng.standard_normal((H, W))
        if b % 2:
            m[:] = 0.5                                                       # a constant map: no value range to bin
        else:
            m[b % H, (7 * b) % W] = np.nan                                   # NaN wins the first arg-max, then the rest is ordinary
        maps.append(m)
    s0 = np.ascontiguousarray(np.stack(maps))
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    s = t(s0, dev).clone()
    act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
    am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
    hov = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    picks, npk = greedy_select(s, n, 1, mrad, act, sel, am, t(gt, dev), handover=hov)
    assert int((hov[:, 0] != 0).sum()) == B
    for b in range(B):
        so = s0[b].copy()
        a_o = np.zeros((H, W), bool); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
        _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, mrad, a_o, s_o, m_o, gt[b], True)
        k = int(npk[b])
        assert k == len(po) and bits_equal(picks[b, :k].cpu().numpy(), po), b
        assert np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(am[b].cpu().numpy(), m_o), b


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_plateaus_of_exact_ties_are_walked_by_the_sweep(dev, dt):
    """Exact-tie plateaus at full width: quantised maps (a dozen plateaus, each thousands of pixels, the picks crossing several),
    a map whose top 40 % is one value, plateaus beside -inf regions and image borders, a batch of them -- picks, masks and the
    mutated map bit for bit the oracle's, every image finished by the sweep (hand-over reason 'done')."""
    from halo_amd import _lib
    from halo_amd.core.active.build import greedy_select
    from oracle import halo_oracle as ho
    rng = np.random.default_rng(77)
    H, W, mrad = 200, 328, 5
    base = ho.bilinear(rng.standard_normal((1, H // 8, W // 8)), (H, W))[0]
    quant = np.round(base * 3) / 3                                                   # ~12 levels
    top = base.copy(); top[base > np.quantile(base, 0.6)] = 2.5
    holes = quant.copy(); holes[rng.random((H, W)) < 0.3] = -np.inf                  # ragged plateaus
    edge = np.full((H, W), -1.0); edge[:, :7] = 1.0; edge[:9, :] = 1.0; edge[-3:, :] = 1.0; edge[:, -4:] = 1.0      # plateaus along the borders
    edge += 1e-9 * 0
    maps = [quant, top, holes, edge, np.where(base > 0, 1.0, base)]
    B = len(maps)
    s0 = np.ascontiguousarray(np.stack(maps).astype(dt))
    gt = rng.integers(0, 19, (B, H, W)).astype(np.int64)
    for n in (40, 700):
        s = t(s0, dev).clone()
        act = torch.zeros((B, H, W), dtype=torch.bool, device=dev); sel = torch.zeros_like(act)
        am = torch.full((B, H, W), 255, dtype=torch.int64, device=dev)
        hov = torch.full((B, 2), -1, dtype=torch.int32, device=dev)
        picks, npk = greedy_select(s, n, 1, mrad, act, sel, am, t(gt, dev), handover=hov)
        hv = hov.cpu().numpy()
        for b in range(B):
            so = s0[b].copy()
            a_o = np.zeros((H, W), bool); s_o = np.zeros((H, W), bool); m_o = np.full((H, W), 255, np.int64)
            _, _, _, _, po = ho.select_pixels_to_label(so, n, 1, mrad, a_o, s_o, m_o, gt[b], True)
            k = int(npk[b])
            assert k == len(po) and bits_equal(picks[b, :k].cpu().numpy(), po), (n, b, k, len(po))
            assert np.array_equal(act[b].cpu().numpy(), a_o) and np.array_equal(am[b].cpu().numpy(), m_o), (n, b)
            assert bits_equal(s[b].cpu().numpy(), so), (n, b)
            assert _lib.SWEEP_REASONS[int(hv[b, 0])] == "done" and hv[b, 1] == k, (n, b, hv[b])
