"""The float32 chain of FloatingRegionScore.forward (core/active/floating_region.py:152, 72-76, 90, 112-121, 204-210) against
what torch's CPU kernels return, BIT FOR BIT, op by op -- the facts that make the selected-pixel masks the reference's by
construction instead of by margin:

  torch.softmax(dim=0)          ATen vec_softmax: max, Sleef expf_u10 (x - max), running sum from +0, one division     exact
  torch.sum(dim=0 / dim=1)      ATen multi_row_sum: cascade of accumulators flushed every 16 / 256 / 4096 rows           exact
  x / math.log(19)              true division by the float32 constant                                                    exact
  entropy_conv (3 x 3 ones)     oneDNN above 20480 pixels: taps in row-major order from +0                               exact
  torch.log                     MKL VML vsLn (closed source, differs between its AVX2 and AVX-512 paths): the oracle's
                                correctly rounded logf is the ISA-independent target; the distance is measured below

Everything except the logarithm is independent of the host's instruction set (checked under ATEN_CPU_CAPABILITY=avx2 too).
These tests need torch on the CPU only; none reads /root/reference.
"""
import ctypes
import math
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle.halo_oracle as ho
from conftest import ROOT


def bits_differ(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return int((~((a == b) | ((a != a) & (b != b)))).sum())


def smooth_logits(O, H, W, seed, scale=2.0):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(1, O, max(H // 4, 2), max(W // 4, 2), generator=g) * scale
    return F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=True)[0].contiguous()


@pytest.mark.parametrize("O,H,W", [(19, 64, 128), (16, 48, 96), (19, 128, 2048), (3, 32, 64), (40, 40, 64), (19, 160, 320)])
def test_oracle_softmax_is_torch_softmax_bit_for_bit(O, H, W):
    assert (H * W) % (16 * max(1, torch.get_num_threads())) == 0 or H * W < 32768
    lg = smooth_logits(O, H, W, 3)
    lg[:, 0, 0] = 0.0                      # uniform pixel
    lg[:, 1, 1] *= 60.0                    # saturated pixel: exact zeros among the probabilities
    want = torch.softmax(lg, dim=0).numpy()
    got = ho.softmax(lg.numpy())
    assert bits_differ(got, want) == 0


def test_softmax_ragged_sizes_differ_only_in_atens_scalar_tails():
    """ATen's vec_softmax walks the flattened H x W axis 16 (AVX-512) or 8 (AVX2) pixels at a time inside each thread's chunk and
    finishes a chunk's remainder with std::exp (glibc) instead of Sleef: fewer than one vector per thread of pixels, whose position
    depends on the thread count.  Every image size the reference runs (1024 x 2048, 640 x 1280, 160 x 320) divides evenly; the oracle
    uses the vector function everywhere."""
    O, H, W = 19, 101, 203
    lg = smooth_logits(O, H, W, 3)
    want = torch.softmax(lg, dim=0).numpy()
    got = ho.softmax(lg.numpy())
    px = (got != want).any(axis=0)
    assert int(px.sum()) <= 16 * (torch.get_num_threads() + 1)
    assert np.abs(got - want).max() <= 6e-8


@pytest.mark.parametrize("n", [1, 3, 15, 16, 17, 19, 31, 32, 33, 100, 255, 256, 257, 300, 4097, 5000])
def test_oracle_class_sum_is_torch_sum_bit_for_bit(n):
    """ATen reduces 64 columns at a time with the cascade the oracle states; the last (columns mod 64) of a thread's chunk go through
    row_sum, a four-way interleaved variant -- so the shapes here, like every image size the reference runs, are multiples of 64
    columns per thread (a ragged width moves fewer than 64 pixels per thread to that other order)."""
    g = torch.Generator().manual_seed(n)
    t = (torch.rand(n, 32, 64, generator=g) - 0.3) * torch.exp(torch.randn(n, 32, 64, generator=g) * 3)
    assert bits_differ(ho.sum_dim0(t.numpy()), torch.sum(t, dim=0).numpy()) == 0
    # the same reduction over dim 1 of a (1, n, H, W) tensor (compute_region_impurity, floating_region.py:116-119)
    assert bits_differ(ho.sum_dim0(t.numpy()), torch.sum(t[None], dim=1, keepdim=True).numpy()[0, 0]) == 0
    if n == 19:
        big = (torch.rand(n, 512, 1024, generator=g) - 0.3)
        assert bits_differ(ho.sum_dim0(big.numpy()), torch.sum(big, dim=0).numpy()) == 0


def test_division_by_log19_is_true_division():
    g = torch.Generator().manual_seed(1)
    s = torch.rand(257, 129, generator=g) * 3
    want = (s / math.log(19)).numpy()
    assert bits_differ(s.numpy() / np.float32(math.log(19)), want) == 0
    assert bits_differ(s.numpy() * (np.float32(1) / np.float32(math.log(19))), want) > 0        # not the reciprocal form


@pytest.mark.parametrize("H,W", [(101, 203), (128, 161), (160, 320), (255, 511), (640, 1280)])
def test_oracle_box_sum_is_atens_conv_above_20480_pixels(H, W):
    assert H * W > 20480
    g = torch.Generator().manual_seed(H)
    x = torch.rand(H, W, generator=g)
    conv = torch.nn.Conv2d(1, 1, 3, 1, 1, bias=False)
    conv.weight.data.fill_(1.0)
    with torch.no_grad():
        want = conv(x[None, None])[0, 0].numpy()
    assert bits_differ(ho.box_sum(x.numpy(), 3), want) == 0


def test_oracle_logf_is_correctly_rounded_on_the_paths_domain():
    """p + 1e-6 for p in [0, 1], d + 1e-6 for window fractions, 1 +- z of the float32 artanh: every float32 of [9e-7, 2) with a
    stride, against the binary64 logarithm rounded once (they agree on every positive normal float32 but 3, all above 9; the
    exhaustive run is tools/gen_logf_table.py's companion check, recorded in DESIGN.md)."""
    lo, hi = np.float32(9e-7).view(np.uint32), np.float32(2.0).view(np.uint32)
    u = np.arange(int(lo), int(hi), 5, dtype=np.uint32)
    x = u.view(np.float32)
    want = np.log(x.astype(np.float64)).astype(np.float32)
    assert bits_differ(ho.logf(x), want) == 0
    # the window fractions of compute_region_impurity: every d = n / c + 1e-6, c in {4, 6, 9} (3 x 3) and {9, ..., 25} (5 x 5)
    d = np.array([np.float32(n) / np.float32(c) + np.float32(1e-6) for c in range(1, 26) for n in range(0, c + 1)], np.float32)
    assert bits_differ(ho.logf(d), np.log(d.astype(np.float64)).astype(np.float32)) == 0


def test_distance_to_torch_log_on_this_host():
    """Not a parity claim, a measurement with a loose ceiling: on an AVX-512 host MKL's vsLn differs from the correctly rounded value
    in ~5e-5 of the softmax probabilities (one ulp each); on an AVX2 host in ~7e-2.  The window fractions agree exactly on AVX-512."""
    lg = smooth_logits(19, 256, 512, 5)
    p = torch.softmax(lg, dim=0)
    q = (p + 1e-6)
    frac = bits_differ(ho.logf(q.numpy()), torch.log(q).numpy()) / q.numel()
    cap = torch.backends.cpu.get_cpu_capability()
    print("torch.log vs correctly rounded logf: %.3e of the values differ (%s)" % (frac, cap))
    if cap == "AVX512":
        assert frac < 1e-3
        d = torch.tensor([n / c + 1e-6 for c in (4, 6, 9) for n in range(1, c + 1)], dtype=torch.float32)
        d = (torch.tensor([float(n) for c in (4, 6, 9) for n in range(1, c + 1)]) /
             torch.tensor([float(c) for c in (4, 6, 9) for n in range(1, c + 1)]) + 1e-6)
        assert bits_differ(ho.logf(d.numpy()), torch.log(d.repeat(64)).numpy()[:d.numel()]) == 0
    else:
        assert frac < 0.2


def test_pixel_entropy_chain_against_torch():
    """softmax -> -p log(p + 1e-6) -> sum -> / log 19 (floating_region.py:123-127) with torch's own log values handed in, so that the
    closed-source logarithm is out of the picture: the rest of the chain is bit for bit."""
    lg = smooth_logits(19, 96, 160, 9)
    p = torch.softmax(lg, dim=0)
    terms = (-p * torch.log(p + 1e-6))
    want = (torch.sum(terms, dim=0) / math.log(19)).numpy()
    got = ho.sum_dim0(terms.numpy()) / np.float32(math.log(19))
    assert bits_differ(got, want) == 0
    # and end to end, the only difference is the logarithm's: a handful of pixels, one ulp
    ent = ho.uncertainty_from_probs(ho.softmax(lg.numpy()), "pixel_entropy", do_box=False)[0, 0]
    nd = bits_differ(ent, want)
    print("pixel entropy: %d of %d pixels differ from torch's" % (nd, ent.size))
    if torch.backends.cpu.get_cpu_capability() == "AVX512":
        assert nd <= ent.size // 500
        assert np.abs(ent - want).max() <= 2.5e-7


_SHIM = None


def _sleef():
    global _SHIM
    if _SHIM is None:
        src = os.path.join(ROOT, "tests", "native", "sleef_shim.c")
        out = os.path.join(ROOT, "oracle", "_build", "libsleef_shim.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        r = subprocess.run(["gcc", "-O2", "-mavx2", "-shared", "-fPIC", src, "-o", out, "-ldl"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lib = ctypes.CDLL(out)
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so")
        if not lib.shim_open(tl.encode()):
            pytest.skip("libtorch_cpu.so not loadable through dlopen")
        _SHIM = lib
    return _SHIM


@pytest.mark.skipif(shutil.which("gcc") is None or "avx2" not in open("/proc/cpuinfo").read(), reason="needs gcc and an AVX2 host")
def test_oracle_expf_is_sleefs_expf_u10_as_linked_into_libtorch():
    lib = _sleef()
    # every float32 of [-105, -2^-20] with a stride, the zeros, the cut-offs, the positive range up to the overflow
    u = np.concatenate([np.arange(np.float32(-2.0 ** -20).view(np.uint32), np.float32(-105.0).view(np.uint32), 37, dtype=np.uint32),
                        np.arange(0, np.float32(101.0).view(np.uint32), 97, dtype=np.uint32)])
    u = u[: u.size // 8 * 8]
    x = np.ascontiguousarray(u.view(np.float32))
    y = np.empty_like(x)
    rc = lib.shim_call_f8(b"Sleef_expf8_u10", x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    if rc != 0:
        pytest.skip("this libtorch does not export Sleef_expf8_u10")
    assert bits_differ(ho.expf(x), y) == 0


def test_both_headers_hold_the_generated_logf_table():
    """oracle/halo_oracle_math.h and halo_amd/csrc/halo_devmath.hpp carry their own copies of the (r_j, -log r_j) table;
    tools/gen_logf_table.py recomputes it with mpmath and compares both."""
    pytest.importorskip("mpmath")
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_logf_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
