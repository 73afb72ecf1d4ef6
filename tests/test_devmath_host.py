"""The device math recipes (halo_amd/csrc/halo_devmath.hpp: straight-line exp/log with the special cases patched in by
selects, and the *_core forms without them) evaluated on the host against the oracle's branchy statement
(oracle/halo_oracle_math.h).  Both are fixed sequences of IEEE-754 operations, so the host computes what gfx950 computes;
the check walks float32 bit patterns with a stride (every pattern near the cut-offs, infinities and subnormals) and a
few million float64 patterns.  Stride 1 -- all 2^32 patterns -- takes ten minutes and was run once per change of the header."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_device_math_recipes_match_the_oracle_on_the_host(tmp_path):
    exe = tmp_path / "devmath_check"
    src = os.path.join(ROOT, "tests", "native", "devmath_host_check.cpp")
    r = subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", src, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "509"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
