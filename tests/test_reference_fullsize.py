"""The path against the REFERENCE at the headline size: 1024 x 2048, 19 classes, 2331 regions.

tests/golden/fullsize_picks.npz holds what the reference's own FloatingRegionScore.forward + select_pixels_to_label returned
on tests/fullsize_inputs.build(seed, ...) in the build container (tests/golden/make_fixtures.py:gen_fullsize; re-run and
compared array by array in tests/test_fixtures_reproduce.py): the ordered pick table, a digest of the three mask arrays
and every 1031st pixel of the three maps, for the bench's shape (entropy x radius, C = 256), the stressed variant, `ripu`
with mask radius 3 and the reference's DEFAULT purity `hyper`.

  * test_oracle_reproduces_the_references_full_size_tables    runs wherever the oracle builds (this container, the GPU box)
  * test_live_reference_at_full_size                          the reference itself on seeds the fixture does NOT hold
                                                              (build container only: /root/reference never travels)
  * tests/test_gpu_parity.py::test_hip_reproduces_the_references_full_size_tables     the HIP path against the same tables

Scores within 1e-4 (north_star), picks and masks exact.  How close the maps are BITWISE, and the mask-exact rate over 50
seeds per branch, are measurements: tools/flip_rate.py -> profiles/r06_flip_rate_*.txt.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, GOLDEN)
import fullsize_inputs as fi                                    # noqa: E402
from make_fixtures import FULLSIZE, SAMPLE_STRIDE, mask_digest  # noqa: E402

REF = os.environ.get("HALO_REFERENCE", "/root/reference")


def run_oracle(inp, branch, n=None):
    import oracle.halo_oracle as ho
    unc, pur, norm, mrad, K = fi.BRANCHES[branch]
    H, W = inp["gt"].shape
    n = fi.n_regions(H, W) if n is None else n
    score, imp, uncm = ho.floating_region_score(inp["logit"], decoder_out=inp["embed"], unc_type=unc, pur_type=pur, normalize=norm,
                                                ground_truth=inp["gt"], size=3, purity_type=pur, K=K)
    a, s, m = inp["prior"].copy(), np.zeros((H, W), bool), np.full((H, W), 255, np.int64)
    score[a] = -np.inf                                          # core/active/build.py:146
    s0 = score.copy()
    _, _, _, _, picks = ho.select_pixels_to_label(score, n, 1, mrad, a, s, m, inp["gt"], return_picks=True)
    return dict(score=s0, impurity=imp, uncertainty=uncm, active=a, selected=s, active_mask=m, picks=picks)


def assert_matches_reference(res, ref_picks, ref_samples, ref_digest, what):
    assert res["picks"].shape == ref_picks.shape, what
    assert np.array_equal(res["picks"][:, :2], ref_picks[:, :2]), "%s: %d of %d picks differ from the reference's" % (
        what, int((res["picks"][:, :2] != ref_picks[:, :2]).any(axis=1).sum()), len(ref_picks))
    assert np.abs(res["picks"][:, 2] - ref_picks[:, 2]).max() < 1e-4, what
    assert np.array_equal(mask_digest(res), ref_digest), what + ": active / selected / active_mask differ from the reference's"
    for k in ("score", "impurity", "uncertainty"):
        got, want = res[k].ravel()[::SAMPLE_STRIDE], ref_samples[k]
        assert got.dtype == want.dtype, (what, k)
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got)) and np.abs(got[fin].astype(np.float64) - want[fin].astype(np.float64)).max() < 1e-4, (what, k)


@pytest.mark.parametrize("tag", sorted(FULLSIZE))
def test_oracle_reproduces_the_references_full_size_tables(tag):
    d = np.load(os.path.join(GOLDEN, "fullsize_picks.npz"))
    seed, C, branch, mods, f32 = FULLSIZE[tag]
    inp = fi.build(seed, C=C, mods=mods, f32_embed=f32)
    assert fi.digest(inp).encode() == d[tag + "__digest"].tobytes(), "the inputs are not the ones the reference ran on (numpy stream / oracle changed)"
    res = run_oracle(inp, branch)
    assert [int(res["selected"].sum()), int(res["active"].sum())] == list(d[tag + "__n_selected"])
    assert_matches_reference(res, d[tag + "__picks"], {k: d[f"{tag}__{k}_sample"] for k in ("score", "impurity", "uncertainty")},
                             d[tag + "__mask_digest"], tag)
    if branch in ("ripu", "hyper"):                             # the window-histogram impurity is the reference's bit for bit
        assert np.array_equal(res["impurity"].ravel()[::SAMPLE_STRIDE], d[tag + "__impurity_sample"])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "core", "active")), reason="reference tree not present (it never travels)")
@pytest.mark.parametrize("branch,seed,C,mods", [("hyper", 101, 64, ("late_round",))])
def test_live_reference_at_full_size(branch, seed, C, mods):
    """Seeds the fixture does not hold, the reference run here and now."""
    import torch
    from make_fixtures import import_reference, run_reference_fullsize
    cfg, hyp, fr, ab = import_reference()
    cfg.MODEL.CURVATURE, cfg.MODEL.NUM_CLASSES = 1.0, 19
    torch.set_num_threads(min(8, torch.get_num_threads()))
    inp = fi.build(seed, C=C, mods=mods)
    ref = run_reference_fullsize(fr, ab, inp, branch)
    res = run_oracle(inp, branch)
    fin = np.isfinite(ref["score"])
    assert np.abs(res["score"][fin].astype(np.float64) - ref["score"][fin].astype(np.float64)).max() < 1e-4
    assert np.array_equal(res["picks"][:, :2], ref["picks"][:, :2])
    for k in ("active", "selected", "active_mask"):
        assert np.array_equal(res[k], ref[k]), k
    if branch == "hyper":
        assert np.array_equal(res["impurity"], ref["impurity"])           # bit for bit
        nd = int((res["uncertainty"] != ref["uncertainty"]).sum())
        assert nd < 2048, "uncertainty: %d pixels differ (the closed-source logarithm accounts for ~80)" % nd


def _driver_cfg():
    import types
    return types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))


def test_oracle_driver_reproduces_the_references_region_selection_files_at_full_size():
    """tests/golden/fullsize_driver.npz: digests of the mask PNG and the two indicator maps the REFERENCE's RegionSelection
    (core/active/build.py:71-186, its own F.interpolate calls included) wrote for the real pipeline's geometry -- a 64-channel
    float64 embedding at 160 x 320, logits at 640 x 1280, labels 1024 x 2048 -- over two rounds.  The oracle driver, fed the same
    low-res arrays and carrying its own files from round 1 into round 2, must leave the same bytes."""
    import oracle.halo_oracle as ho
    from make_fixtures import DRIVER_SEEDS, files_digest
    d = np.load(os.path.join(GOLDEN, "fullsize_driver.npz"))
    cfg = _driver_cfg()
    for seed in DRIVER_SEEDS:
        inp = fi.build_driver_inputs(seed)
        assert fi.digest(inp).encode() == d[f"s{seed}__digest"].tobytes()
        H, W = inp["gt"].shape
        mask, act, sel = np.full((H, W), 255, np.int64), np.zeros((H, W), bool), np.zeros((H, W), bool)
        for rnd in (1, 2):
            (m8, act, sel, picks), = ho.region_selection(cfg, [dict(logit_lr=inp["logit_lr"], embed_lr=inp["embed_lr"], origin_label=inp["gt"],
                                                                    active=act, selected=sel, origin_mask=mask)])
            assert len(picks) == 2331
            assert [int(sel.sum()), int(act.sum()), int((m8 != 255).sum())] == list(d[f"s{seed}__r{rnd}_counts"]), rnd
            assert np.array_equal(files_digest(m8, act, sel), d[f"s{seed}__r{rnd}_files_digest"]), "round %d: not the reference's files" % rnd
            mask = m8.astype(np.int64)
