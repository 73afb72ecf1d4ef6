import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# (unc_type, pur_type) behind each fixture tag written by tests/golden/make_fixtures.py
COMBOS = {
    "halo": ("entropy", "radius"),
    "ripu": ("entropy", "ripu"),
    "hyper": ("entropy", "hyper"),
    "hyperK10": ("entropy", "hyper"),
    "none_radius": ("none", "radius"),
    "pixent_euc": ("pixel_entropy", "euc_norm"),
    "oracle": ("oracle_acc", "oracle_ripu"),
    "ent_none": ("entropy", "none"),
    "vestigial": ("hyperbolic", "ripu"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # the GPU boxes show 256 CPUs under a 16-core cgroup quota: thread pools as wide as the machine get throttled
    # (a 1-minute suite was seen taking 7); size torch's intra-op pool and the oracle's OpenMP team to what is usable
    try:
        import torch
        from oracle import halo_oracle
        torch.set_num_threads(max(1, min(torch.get_num_threads(), halo_oracle.usable_cpus())))
    except Exception:
        pass


def case_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "case_*.npz")))


def case_tags(d):
    return sorted({k.split("__")[0] for k in d.files if k.endswith("__score")})


def all_case_combos():
    out = []
    for f in case_files():
        d = np.load(f)
        for t in case_tags(d):
            out.append((os.path.basename(f)[:-4], t))
    return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


def assert_same_nan(a, b):
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"


def max_abs_diff(a, b):
    """max |a-b| where NaNs must coincide and infinities must be equal."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert_same_nan(a, b)
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), "infinities differ"
    m = np.isnan(a) | inf
    if m.all():
        return 0.0
    return float(np.abs(a[~m] - b[~m]).max())


def opposing_neighbours_embedding(C=16, h=12, w=20, seed=0):
    """Adversarial low-res embedding for the Gram form of the radius: a checkerboard of v and -v (1 - eps) with eps
    from 1e-1 to 1e-12 per cell -- the interpolated vector between two neighbours nearly vanishes and the 10 Gram terms cancel --
    plus a block of random vectors, a block of random vectors projected onto the ball's boundary (norm 1 - 1e-5) and a block of
    OPPOSING boundary vectors.  (1, C, h, w) float64."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal(C)
    u /= np.linalg.norm(u)
    eps = 10.0 ** (-rng.uniform(1, 12, (h, w)))
    sign = (-1.0) ** np.add.outer(np.arange(h), np.arange(w))
    emb = np.zeros((1, C, h, w))
    emb[0] = u[:, None, None] * (0.5 * (1 - eps) * sign)[None]
    bh, bw = max(1, h // 3), max(1, w // 3)
    emb[0, :, :bh, :bw] = rng.standard_normal((C, bh, bw)) * 0.1
    b = rng.standard_normal((C, bh, bw))
    emb[0, :, -bh:, -bw:] = b / np.linalg.norm(b, axis=0, keepdims=True) * (1 - 1e-5)
    emb[0, :, -bh:, :bw] = u[:, None, None] * ((1 - 1e-5) * sign[-bh:, :bw])[None]
    return emb
