"""SURVEY 5's sanitizer job as committed tests: the CPU-side C of this repository -- the persistence helpers
of the product (halo_amd/csrc/halo_host.c: a hand-written deflate encoder, PCLMULQDQ CRC folding, per-thread scratch, up to 16
concurrent writer threads) and the oracle (oracle/halo_oracle.c) -- built with -fsanitize=address,undefined and driven over
every entry point, and the writer functions under -fsanitize=thread with concurrent callers.  Never on the GPU: device code has
no sanitizer on this pool.  The sanitized processes import numpy only (tests/native/sanitize_*_driver.py); the files they write are
decoded here, in an ordinary process."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "halo_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
GCC = shutil.which("gcc")
pytestmark = pytest.mark.skipif(GCC is None, reason="needs gcc")


def _san_lib(name):
    p = subprocess.run([GCC, "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_sanitized(cmd, preload, extra_env=None):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="4")
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0 and "driver ok" in r.stdout, report
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, report
    return r


@pytest.mark.skipif(_san_lib("libasan.so") is None, reason="gcc has no libasan")
def test_host_library_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    import torch
    from PIL import Image
    from halo_amd.core.active.build import _IndicatorTemplate, compose_indicators, compose_mask
    so = str(tmp_path / "libhalo_host_asan.so")
    r = subprocess.run([GCC, "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-fno-sanitize-recover=undefined", os.path.join(CSRC, "halo_host.c"), "-o", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    shape = (45, 77)
    tpl = _IndicatorTemplate.get(shape)
    assert tpl.ok
    tpath = str(tmp_path / "tpl.npz")
    np.savez(tpath, raw=tpl.raw, shape=np.array(shape), off_active=tpl.off["active"], off_selected=tpl.off["selected"],
             crc_active=tpl.crc["active"], crc_selected=tpl.crc["selected"])
    out = tmp_path / "out"
    out.mkdir()
    _run_sanitized([sys.executable, os.path.join(NATIVE, "sanitize_host_driver.py"), so, str(out), tpath], _san_lib("libasan.so"))
    # what the sanitized process wrote, read back by the libraries the training side uses
    got = torch.load(str(out / "ind_plain.pth"))
    assert np.array_equal(got["active"].numpy(), np.load(out / "ind_plain_active.npy")) and np.array_equal(got["selected"].numpy(), np.load(out / "ind_plain_selected.npy"))
    inp = np.load(out / "retire_inputs.npz")
    om = np.full(shape, 255, np.int64)
    want_mask = compose_mask(om, inp["gt"], inp["picks"], 1)
    for mode in ("results", "compose"):
        assert np.array_equal(np.array(Image.open(out / ("ret_%s.png" % mode))), want_mask), mode
        ind = torch.load(str(out / ("ret_%s.pth" % mode)))
        wa, ws = (inp["act"], inp["sel"]) if mode == "results" else compose_indicators(inp["act"], inp["sel"], inp["picks"], 1, 5)
        assert np.array_equal(ind["active"].numpy(), wa) and np.array_equal(ind["selected"].numpy(), ws), mode
    assert np.array_equal(np.array(Image.open(out / "ret_noind.png")), np.full(shape, 255, np.uint8))
    assert Image.open(out / "png_9.png").size == (70000, 2)


@pytest.mark.skipif(_san_lib("libasan.so") is None, reason="gcc has no libasan")
def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    so = str(tmp_path / "libhalo_oracle_asan.so")
    r = subprocess.run([GCC, "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp",
                        "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
                        os.path.join(ROOT, "oracle", "halo_oracle.c"), "-o", so, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    _run_sanitized([sys.executable, os.path.join(NATIVE, "sanitize_oracle_driver.py"), ROOT], _san_lib("libasan.so"), {"HALO_ORACLE_LIB": so})


@pytest.mark.skipif(_san_lib("libtsan.so") is None, reason="gcc has no libtsan")
def test_concurrent_writers_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "host_tsan")
    r = subprocess.run([GCC, "-O1", "-g", "-std=c11", "-pthread", "-fsanitize=thread", os.path.join(CSRC, "halo_host.c"),
                        os.path.join(NATIVE, "host_tsan.c"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "files"
    out.mkdir()
    # (setarch -R: ThreadSanitizer of this gcc cannot map its shadow under the kernel's high-entropy ASLR)
    base = [exe, str(out), "16", "4"]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = None
    for cmd in ([["setarch", os.uname().machine, "-R"] + base] if shutil.which("setarch") else []) + [base]:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        if "0 failed calls" in r.stdout or "ThreadSanitizer: data race" in r.stderr:
            break                                     # the program ran (clean or not); anything else: setarch refused, try without it
    if "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot run under this kernel's address-space layout")
    assert r.returncode == 0 and "0 failed calls" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
