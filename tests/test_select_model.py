"""The value-binned selection ALGORITHM (numpy model of halo_select_binned.hip, tests/select_model.py) against the
CPU oracle's literal restatement of select_pixels_to_label (build.py:27-64): where the model says 'done' its picks
are the oracle's sequence; where it bails, its picks are a prefix of it (the serial kernel continues from there)."""
import numpy as np
import pytest

from select_model import binned_select


def _oracle(score, n, mrad):
    from oracle import halo_oracle as ho
    H, W = score.shape
    act = np.zeros((H, W), bool); sel = np.zeros((H, W), bool); am = np.full((H, W), 255, np.int64)
    _, _, _, _, p = ho.select_pixels_to_label(score.copy(), n, 1, mrad, act, sel, am, np.zeros((H, W), np.int64), True)
    return [(int(r[0]), int(r[1])) for r in p]


def _smooth(rng, H, W, dt):
    from oracle import halo_oracle as ho
    base = rng.standard_normal((1, max(2, H // 4), max(2, W // 4)))
    return ho.bilinear(base, (H, W))[0].astype(dt)


@pytest.mark.parametrize("seed", range(12))
def test_model_equals_oracle_on_random_maps(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(8, 90)), int(rng.integers(8, 130))
    mrad = int(rng.choice([1, 2, 3, 5, 9, 14]))
    n = int(rng.integers(1, 80))
    dt = np.float32 if seed % 2 else np.float64
    sc = _smooth(rng, H, W, dt) if seed % 3 else rng.standard_normal((H, W)).astype(dt)
    if seed % 4 == 0:
        sc[rng.random((H, W)) < 0.2] = -np.inf
    want = _oracle(sc, n, mrad)
    status, picks, st = binned_select(sc, n, mrad, target=int(rng.choice([8, 32, 128])))
    assert status == "done", st
    assert picks == want


def test_candidate_bound_is_tight_enough_and_truncation_bails_with_a_prefix():
    rng = np.random.default_rng(5)
    sc = _smooth(rng, 96, 160, np.float64)
    want = _oracle(sc, 60, 5)
    status, picks, st = binned_select(sc, 60, 5)
    assert status == "done" and picks == want and st["ncand"] >= min(60 * 121, 96 * 160)
    # a threshold bin larger than the staging capacity (a plateau of ties under a few high pixels): the bin is
    # dropped, the candidates above it run out before n picks, and the sweep hands over a correct prefix
    sc = rng.random((96, 160)) * 0.5
    sc[20:70, :] = 0.5
    hot = rng.random((96, 160)) < 0.003
    sc[hot] = 0.5 + 0.5 * rng.random(int(hot.sum()))
    want = _oracle(sc, 60, 5)
    status, picks, st = binned_select(sc, 60, 5, captot=7500)      # kneed 7260 <= captot < 8000 + |hot|
    assert picks == want[:len(picks)]
    assert status == "bail" and st["reason"] == "exhausted" and 0 < len(picks) < 60, (status, st, len(picks))


def test_degenerate_maps_bail_or_finish_correctly():
    rng = np.random.default_rng(6)
    H, W = 40, 64
    for kind in ("nan", "posinf", "const", "all_masked", "plateau", "two_values", "exhaust"):
        sc = rng.standard_normal((H, W))
        n, mrad = 30, 5
        if kind == "nan":
            sc[3, 4] = np.nan
        if kind == "posinf":
            sc[7, 9] = np.inf
        if kind == "const":
            sc[:] = 0.25
        if kind == "all_masked":
            sc[:] = -np.inf
        if kind == "plateau":
            sc[:, :] = np.round(sc)                    # a handful of distinct values: thousands of exact ties per bin
        if kind == "two_values":
            sc[:] = 0.0; sc[::7, ::5] = 1.0
        if kind == "exhaust":
            sc[:] = -np.inf; sc[5:9, 5:30] = rng.standard_normal((4, 25)); n = 200    # fewer pickable pixels than regions
        want = _oracle(sc, n, mrad)
        status, picks, st = binned_select(sc, n, mrad)
        assert picks == want[:len(picks)], kind
        if status == "done":
            assert picks == want, kind
        if kind in ("nan", "posinf", "const", "all_masked"):
            assert status == "bail" and st["reason"] == "range" and not picks, kind
        if kind == "exhaust":
            assert status == "done" and len(picks) == len(want) < n, kind


def test_a_full_bin_only_matters_when_the_sweep_reaches_it():
    """A plateau of exact ties overfills its fine bin.  Below the values the n picks need it is never visited (round 4 handed the
    whole image over untouched as soon as ANY bin from the threshold bin up was full); reached by the walk it is taken in position
    order -- the reference's tie-break -- and the walk goes on below it; only a full bin of MIXED keys (near-ties denser than the
    bins resolve) hands the image over, from there."""
    rng = np.random.default_rng(11)
    H, W, mrad = 96, 160, 3
    sc = _smooth(rng, H, W, np.float64) + 1e-3 * rng.standard_normal((H, W))
    floor = np.quantile(sc, 0.45)
    low = sc.copy(); low[low < floor] = floor                 # 45 % of the map is one value, at the BOTTOM of the candidates
    n = 12
    want = _oracle(low, n, mrad)
    status, picks, st = binned_select(low, n, mrad, captot=H * W)      # the plateau's bin is inside the threshold bin's range
    assert status == "done" and picks == want and "plateaus" not in st, st
    cap = np.quantile(sc, 0.97)
    mid = sc.copy(); mid[(mid > floor) & (mid < cap)] = 0.5 * (floor + cap)      # a plateau the picks must cross
    for n in (120, 400):                                       # ends inside the plateau / goes on below it
        want = _oracle(mid, n, mrad)
        status, picks, st = binned_select(mid, n, mrad, captot=H * W)
        assert status == "done" and st.get("plateaus") == 1 and picks == want, (n, status, st, len(picks))
    top = sc.copy(); top[sc > np.quantile(sc, 0.6)] = 7.0      # the plateau IS the top of the map: every pick a tie-break by position
    want = _oracle(top, 150, mrad)
    status, picks, st = binned_select(top, 150, mrad, captot=H * W)
    assert status == "done" and st.get("plateaus") == 1 and picks == want
    # near-ties: distinct keys packed into one sub-slice of the value range -> a full bin of mixed keys
    near = sc.copy()
    m = (near > floor) & (near < cap)
    near[m] = 0.5 * (floor + cap) + 1e-13 * rng.standard_normal(int(m.sum()))
    want = _oracle(near, 120, mrad)
    status, picks, st = binned_select(near, 120, mrad, captot=H * W)
    assert status == "bail" and st["reason"] == "overflow" and 0 < len(picks) < 120 and picks == want[:len(picks)], (status, st, len(picks))
