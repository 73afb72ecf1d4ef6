"""The strongest parity fact of this repository -- "the reference's own Python, run here, produces these arrays" -- as a test:
re-run tests/golden/make_fixtures.py (which imports /root/reference) into a temporary directory and compare every array of
every committed .npz bit for bit.  Skipped where the reference tree is absent (the GPU box): fixtures are data, the reference
never travels."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF = os.environ.get("HALO_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "core", "active")), reason="reference tree not present (fixtures are data; the reference does not travel)")
def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    env = dict(os.environ, HALO_FIXTURE_OUT=str(tmp_path), HALO_REFERENCE=REF)
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_fixtures.py")], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    committed = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    fresh = sorted(glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert [os.path.basename(f) for f in committed] == [os.path.basename(f) for f in fresh], "the generator writes a different set of files"
    n_arrays = 0
    for fc, ff in zip(committed, fresh):
        a, b = np.load(fc), np.load(ff)
        assert sorted(a.files) == sorted(b.files), os.path.basename(fc)
        for k in a.files:
            x, y = a[k], b[k]
            assert x.dtype == y.dtype and x.shape == y.shape, (os.path.basename(fc), k)
            same = x.tobytes() == y.tobytes() if x.dtype.kind in "fc" else np.array_equal(x, y)
            assert same, "fixture drift: %s[%s] is not what the reference produces today" % (os.path.basename(fc), k)
            n_arrays += 1
    assert n_arrays > 300
