"""CPU-side checks of the C-ABI boundary and host logic (no GPU, no compute calls)."""
import ctypes
import os
import re

import numpy as np

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "halo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(halo_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for must in ("halo_expmap0_project", "halo_hypermlr_logits", "halo_dist0", "halo_pdist", "halo_score_maps",
                 "halo_greedy_select", "halo_version", "halo_last_error", "halo_bilinear_upsample"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from halo_amd import _build, _lib
    so = _build.build()
    assert os.path.exists(so)
    h = ctypes.CDLL(so)
    for s in _declared_symbols():
        assert hasattr(h, s), "libhalo_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == _declared_symbols(), "ctypes table out of sync with include/halo_hip.h"
    assert _lib.lib().halo_version() == _lib.ABI_VERSION
    text = open(os.path.join(ROOT, "include", "halo_hip.h")).read()
    assert "#define HALO_ABI_VERSION %d" % _lib.ABI_VERSION in text


def test_host_library_exports_every_symbol_its_header_declares():
    """include/halo_host.h: the plain-C helpers of the persistence step (libhalo_host.so, no HIP) -- every declared entry point is
    exported, the header's version is the binding's, and the compiler checked the definitions against the declarations (the .c
    file includes the header)."""
    from halo_amd import _build, _hostlib
    text = open(os.path.join(ROOT, "include", "halo_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(halo_[a-z0-9_]+)\s*\(", text)))
    assert {"halo_retire_image", "halo_compose_mask", "halo_write_indicator", "halo_png_gray8_write", "halo_crc32", "halo_host_version"} <= set(syms)
    h = ctypes.CDLL(_build.build_host())
    for s in syms:
        assert hasattr(h, s), "libhalo_host.so does not export %s" % s
    assert "#define HALO_HOST_ABI_VERSION %d" % _hostlib.ABI_VERSION in open(os.path.join(ROOT, "include", "halo_host.h")).read()
    assert _hostlib.lib().halo_host_version() == _hostlib.ABI_VERSION
    assert '#include "../../include/halo_host.h"' in open(os.path.join(ROOT, "halo_amd", "csrc", "halo_host.c")).read()


def test_workspace_queries_are_pure_host_functions():
    from halo_amd import _lib
    L = _lib.lib()
    assert L.halo_score_workspace_bytes(1, 1024, 2048) >= 1024 * 2048 * 18
    assert L.halo_score_workspace_bytes(0, 4, 4) == 0
    assert L.halo_hypermlr_workspace_bytes(19, 256) >= (3 * 19 + 2 * 19 * 256) * 8
    # the fused HyperMLR backward: a size for the shapes it serves (<= 20 classes, 64 | C <= 256), 0 = "use halo_hypermlr_bwd_terms" otherwise
    n = L.halo_hypermlr_backward_workspace_bytes(2, 64, 19, 160 * 320)
    assert n >= (2 * (2 * 19 + 1) * 160 * 320 + 64 * 40 + 256 * 2 * 19 * 64) * 8
    for bad in ((2, 8, 19, 100), (2, 96, 19, 100), (2, 320, 19, 100), (2, 64, 21, 100), (0, 64, 19, 100), (2, 64, 19, 0)):
        assert L.halo_hypermlr_backward_workspace_bytes(*bad) == 0, bad
    assert L.halo_loss_workspace_bytes(129) >= (3 + 1) * 16            # a 1 x 129 image: three strips of the row-walking forward
    # the binned selector stages up to 2 x 121 x n candidates per image (28 bytes each); the serial kernel needs none
    assert L.halo_select_workspace_bytes(4, 1024, 2048, 2331, 5) > 4 * 2 * 121 * 2331 * 28
    assert L.halo_select_workspace_bytes(4, 1024, 2048, 2331, 40) == 256          # mask radius above 14: serial kernel only
    assert L.halo_select_workspace_bytes(0, 8, 8, 1, 1) == 0


def test_argument_errors_are_reported_not_crashed():
    from halo_amd import _lib
    L = _lib.lib()
    rc = L.halo_score_maps(None, 0, None, 0, 0, None, None, 1, 19, 0, 8, 8, 0, 4, 0, 3, 3, 100, 1.0, None, None, None,
                           None, 0, None)
    assert rc == -1 and b"null" in L.halo_last_error()
    rc = L.halo_greedy_select(None, 1, 1, 8, 8, 1, 1, 5, None, None, None, None, None, None, None, 0, 0, None)
    assert rc == -1


def test_no_cpu_fallback():
    """CPU tensors must raise: a silent CPU path would void every parity claim."""
    from halo_amd._lib import HaloHipError
    from halo_amd.core.active.build import select_pixels_to_label
    from halo_amd.core.active.floating_region import FloatingRegionScore
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR
    frs = FloatingRegionScore(in_channels=19, size=3, purity_type="radius")
    with pytest.raises(HaloHipError):
        frs(torch.zeros(1, 19, 8, 8), torch.zeros(1, 4, 8, 8, dtype=torch.float64), unc_type="entropy", pur_type="radius")
    with pytest.raises(HaloHipError):
        HyperMapper().expmap(torch.zeros(2, 3))
    with pytest.raises(HaloHipError):
        with torch.no_grad():
            HyperMLR(4, 3)(torch.zeros(1, 4, 2, 2, dtype=torch.float64))
    with pytest.raises(HaloHipError):
        HyperMLR(4, 3)(torch.zeros(1, 4, 2, 2, dtype=torch.float64))          # training path: no CPU either
    with pytest.raises(HaloHipError):
        z = torch.zeros(8, 8)
        select_pixels_to_label(z, 1, 1, 5, z.bool(), z.bool(), z.long(), z.long())


def test_reference_error_behaviour():
    from halo_amd.core.active.floating_region import FloatingRegionScore
    with pytest.raises(AssertionError, match="error size"):          # floating_region.py:36
        FloatingRegionScore(size=4)
    frs = FloatingRegionScore(size=3, purity_type="radius")
    with pytest.raises(NotImplementedError, match="purity type 'bogus' not implemented"):   # :199-202
        frs(torch.zeros(1, 19, 4, 4), unc_type="entropy", pur_type="bogus")
    assert FloatingRegionScore(size=5, purity_type="hyper", K=7).purity_size == 3           # :54-55
    # nn.Conv2d's own argument check (the reference forwards padding_mode to it, floating_region.py:49,63)
    assert FloatingRegionScore(size=3, padding_mode="reflect").padding_mode == "reflect"
    with pytest.raises(ValueError, match="padding_mode must be one of"):
        FloatingRegionScore(size=3, padding_mode="mirror")


def test_hypermlr_parameters_match_reference_names_and_dtypes():
    from halo_amd.core.utils.hyperbolic import HyperMLR
    m = HyperMLR(64, 19, c=1.0)
    sd = m.state_dict()
    assert set(sd) == {"P_MLR", "A_MLR"}
    assert sd["P_MLR"].shape == (19, 64) and sd["P_MLR"].dtype == torch.float64
    bound = 1.0 / (64 ** 0.5)      # kaiming_uniform_(a=sqrt(5)) on fan_in 64
    assert float(sd["P_MLR"].abs().max()) <= bound


def test_cfg_standin_has_reference_defaults():
    from halo_amd.core.configs import cfg
    assert cfg.MODEL.CURVATURE == 1.0 and cfg.MODEL.NUM_CLASSES == 19 and cfg.MODEL.HYPER is True
    assert cfg.ACTIVE.RADIUS_K == 1 and cfg.ACTIVE.MASK_RADIUS_K == 5 and cfg.ACTIVE.K == 100
    assert cfg.ACTIVE.BUDGET == 0.05 and len(cfg.ACTIVE.SELECT_ITER) == 5


def test_direct_png_writer_decodes_to_the_same_mode_L_image(tmp_path):
    """RegionSelection writes its masks with a direct PNG writer instead of PIL's encoder (build.py:163-164 in the
    reference): what cityscapes.py:231 reads back must be the identical uint8 image, mode 'L'."""
    import numpy as np
    from PIL import Image
    from halo_amd.core.active.build import _persist, write_png_gray8
    rng = np.random.default_rng(0)
    for shape in ((64, 96), (1, 1), (7, 3), (33, 1), (1, 50)):
        a = rng.integers(0, 256, shape).astype(np.uint8)
        if shape == (64, 96):
            a[:] = 255; a[10:13, 20:23] = 7
        p = str(tmp_path / "x.png")
        write_png_gray8(p, a)
        im = Image.open(p)
        assert im.mode == "L" and im.size == (shape[1], shape[0]) and np.array_equal(np.array(im, dtype=np.uint8), a)
        ref = str(tmp_path / "ref.png")
        Image.fromarray(a).save(ref)                                   # the reference's call
        assert np.array_equal(np.array(Image.open(ref)), np.array(Image.open(p)))
    # the driver's persistence step: PNG + indicator dict of CPU bool tensors
    act = torch.from_numpy(rng.random((64, 96)) < 0.1)
    _persist(a if a.shape == (64, 96) else np.full((64, 96), 255, np.uint8), act, act.clone(), str(tmp_path / "m.png"), str(tmp_path / "i.pth"))
    ind = torch.load(str(tmp_path / "i.pth"))
    assert set(ind) == {"active", "selected"} and ind["active"].dtype == torch.bool and torch.equal(ind["active"], act)


def test_native_png_encoder_decodes_to_the_same_image(tmp_path, monkeypatch):
    """halo_amd/csrc/halo_host.c (libhalo_host.so, plain C): one fixed-Huffman deflate block of distance-1 run matches.  Every
    shape / content class decodes (PIL = libpng + zlib) to the identical mode-L image -- runs shorter and longer than the 258-byte
    match limit incl. the 259 / 260 / 261 tails, rows that start with zeros (they join the filter byte's run), noise without any
    run (9 bits per pixel), single pixels, strided row views -- and so does the zlib twin behind HALO_PNG_ZLIB=1."""
    import numpy as np
    from PIL import Image
    from halo_amd import _hostlib
    from halo_amd.core.active.build import write_png_gray8
    assert _hostlib.lib().halo_host_version() == _hostlib.ABI_VERSION
    rng = np.random.default_rng(0)
    p = str(tmp_path / "m.png")

    def check(a, writer=_hostlib.png_gray8_write):
        writer(p, a)
        im = Image.open(p)
        im.load()
        assert im.mode == "L" and im.size == (a.shape[1], a.shape[0]) and np.array_equal(np.array(im), a)

    for shape in ((64, 96), (1, 1), (7, 3), (33, 1), (1, 50), (3, 259), (3, 260), (3, 261), (2, 262), (5, 600), (2, 517), (1, 70000)):
        check(rng.integers(0, 256, shape).astype(np.uint8))
        for v in (0, 255, 7):
            check(np.full(shape, v, np.uint8))
        a = np.full(shape, 255, np.uint8)
        a[rng.random(shape) < 0.05] = 3
        check(a)
        a = np.zeros(shape, np.uint8)
        a[rng.random(shape) < 0.5] = 1
        check(a)
        check(a, write_png_gray8)
    mask = np.full((256, 512), 255, np.uint8)
    for _ in range(146):
        y, x = rng.integers(1, 255), rng.integers(1, 511)
        mask[y - 1:y + 2, x - 1:x + 2] = rng.integers(0, 19, (3, 3))
    check(mask)
    assert os.path.getsize(p) < mask.size // 20                          # runs compress: < 5 % of the raw size
    big = rng.integers(0, 256, (10, 300)).astype(np.uint8)
    check(big[:, 10:200])                                               # rows 300 bytes apart
    check(big[:, 10:200], write_png_gray8)
    monkeypatch.setenv("HALO_PNG_ZLIB", "1")
    check(mask, write_png_gray8)
    buf = np.empty(16, np.uint8)                                          # too small a buffer is refused, not overrun
    assert _hostlib.lib().halo_png_gray8_encode(mask.ctypes.data, 256, 512, 512, buf.ctypes.data, 16) == 0


def test_native_retire_crc_compose_and_indicator_template(tmp_path):
    """libhalo_host.so's one-call writer (RegionSelection, mask_staging="table"): CRC-32 by carry-less multiplication and by
    tables == zlib.crc32 on every length 0..600 and some long ones; halo_compose_mask == the numpy statement compose_mask (low
    bytes, windows clipped at the borders, every integer width); halo_retire_image writes a PNG that decodes to that mask and an
    indicator file that torch.load reads back as the reference's dict of two bool tensors (the per-shape template of torch.save's
    own bytes, self-checked when it is built)."""
    import zlib
    import numpy as np
    from PIL import Image
    from halo_amd import _hostlib
    from halo_amd.core.active.build import _IndicatorTemplate, compose_mask
    L = _hostlib.lib()
    rng = np.random.default_rng(1)
    for mode in (0, 1):
        L.halo_crc32_mode(mode)
        try:
            for n in list(range(0, 600)) + [1023, 4097, 65536, 2097153]:
                a = rng.integers(0, 256, n).astype(np.uint8)
                for start in (0, 0x12345678):
                    assert L.halo_crc32(start, a.ctypes.data, n) == (zlib.crc32(a.tobytes(), start) & 0xffffffff), (mode, n, start)
        finally:
            L.halo_crc32_mode(0)
    for (H, W), dt in (((40, 64), np.int64), ((7, 3), np.int32), ((33, 70), np.uint8), ((1, 1), np.int16)):
        om = rng.integers(0, 256 if dt == np.uint8 else 600, (H, W)).astype(dt)
        gt = rng.integers(0, 256 if dt == np.uint8 else 300, (H, W)).astype(dt)
        k = min(9, H * W)
        picks = np.zeros((k + 2, 3))
        picks[:k, 0], picks[:k, 1] = rng.integers(0, H, k), rng.integers(0, W, k)
        picks[0, :2] = (0, 0)
        picks[k - 1, :2] = (H - 1, W - 1)
        act, sel = np.ascontiguousarray(rng.random((H, W)) < 0.2), np.ascontiguousarray(rng.random((H, W)) < 0.05)
        tpl = _IndicatorTemplate.get((H, W))
        assert tpl.ok
        for radius in (0, 1, 2):
            want = compose_mask(om, gt, picks[:k], radius)
            ref = om.astype(np.int64).copy()
            for h, w, _ in picks[:k]:
                h, w = int(h), int(w)
                ref[max(h - radius, 0):h + radius + 1, max(w - radius, 0):w + radius + 1] = gt[max(h - radius, 0):h + radius + 1, max(w - radius, 0):w + radius + 1]
            assert np.array_equal(want, ref.astype(np.uint8))                       # the numpy statement == the reference's slices + cast
            got = np.empty((H, W), np.uint8)
            assert L.halo_compose_mask(got.ctypes.data, om.ctypes.data, om.dtype.itemsize, gt.ctypes.data, gt.dtype.itemsize, H, W,
                                       picks.ctypes.data, k, radius) == 0
            assert np.array_equal(got, want)
            p1, p2 = str(tmp_path / "m.png"), str(tmp_path / "i.pth")
            _hostlib.retire_image(p1, p2, om, gt, picks, k, radius, act, sel, tpl)
            im = Image.open(p1)
            assert im.mode == "L" and np.array_equal(np.array(im), want)
            d = torch.load(p2)
            assert set(d) == {"active", "selected"} and d["active"].dtype == torch.bool and d["selected"].dtype == torch.bool
            assert np.array_equal(d["active"].numpy(), act) and np.array_equal(d["selected"].numpy(), sel)
    with pytest.raises(OSError):
        _hostlib.retire_image(str(tmp_path / "no_such_dir" / "m.png"), str(tmp_path / "i.pth"), om, gt, picks, 0, 1, act, sel, None)


def test_native_writer_rewrites_files_in_place_with_the_same_bytes(tmp_path):
    """Round after round the acquisition rewrites the same mask / indicator paths; the native writer does not truncate first (the
    file keeps its page-cache pages: halo_host.c:write_pieces) and cuts the length afterwards.  Over a longer old file, over a
    shorter one and into a fresh path the bytes on disk are identical; a directory in the way is an OSError."""
    from halo_amd import _hostlib
    from halo_amd.core.active.build import _IndicatorTemplate
    rng = np.random.default_rng(77)
    H, W = 24, 40
    om = np.full((H, W), 255, np.int64)
    gt = rng.integers(0, 19, (H, W)).astype(np.int64)
    act, sel = np.ascontiguousarray(rng.random((H, W)) < 0.2), np.ascontiguousarray(rng.random((H, W)) < 0.05)
    picks = np.array([[3.0, 4.0, 0.5], [20.0, 33.0, 0.25]])
    tpl = _IndicatorTemplate.get((H, W))
    fresh = (str(tmp_path / "fresh.png"), str(tmp_path / "fresh.pth"))
    _hostlib.retire_image(fresh[0], fresh[1], om, gt, picks, 2, 1, act, sel, tpl)
    want = tuple(open(f, "rb").read() for f in fresh)
    for old_len in (1 << 20, 7, 0):
        paths = (str(tmp_path / f"m{old_len}.png"), str(tmp_path / f"i{old_len}.pth"))
        for f in paths:
            open(f, "wb").write(b"\xa5" * old_len)
        _hostlib.retire_image(paths[0], paths[1], om, gt, picks, 2, 1, act, sel, tpl)
        assert tuple(open(f, "rb").read() for f in paths) == want, old_len
        _hostlib.png_gray8_write(paths[1], rng.integers(0, 256, (64, 64)).astype(np.uint8))      # a longer file over it, then the short one again
        _hostlib.png_gray8_write(paths[1], np.full((H, W), 255, np.uint8))
        assert np.array_equal(np.array(__import__("PIL.Image").Image.open(paths[1])), np.full((H, W), 255, np.uint8))
        assert os.path.getsize(paths[1]) < 200
    os.mkdir(tmp_path / "dir.png")
    with pytest.raises(OSError):
        _hostlib.png_gray8_write(str(tmp_path / "dir.png"), np.zeros((4, 4), np.uint8))


def test_runtime_calls_go_through_the_hip_runtime_torch_bundles():
    """The few plain HIP runtime calls of the host code (the recording stream of RegionSelection) must reach the runtime instance
    torch runs on: where the wheel bundles libamdhip64.so, _lib.hip_runtime() is that file, not what a bare soname finds in /opt/rocm
    (two runtime instances in one process do not share streams)."""
    import torch
    from halo_amd import _lib
    h = _lib.hip_runtime()
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        assert os.path.samefile(h._name, bundled)
    for sym in ("hipStreamCreateWithFlags", "hipStreamEndCapture", "hipEventSynchronize"):
        assert hasattr(h, sym), sym


def test_import_leaves_the_environment_alone_and_configure_is_explicit(monkeypatch):
    """VERDICT r3 #5 / ADVICE r3: `import halo_amd` must not set GPU_MAX_HW_QUEUES (a process-wide runtime setting that also
    governs the training iterations' streams).  halo_amd.configure(hw_queues=2) is the explicit opt-in (bench.py and tools/ call
    it before the first HIP call).  (Rounds 3-5: RegionSelection also hinted at two queues; with round 6's selector its driver measures
    FASTER on ROCm's default of four, profiles/r06_hw_queues.txt, and the hint is gone.)"""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = ("import sys, os; sys.path.insert(0, %r); import halo_amd; import halo_amd.core.active.build; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES', 'unset')); print(halo_amd.configure(hw_queues=2)); "
            "print(os.environ['GPU_MAX_HW_QUEUES']); print(halo_amd.configure())") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.split()
    assert out == ["unset", "2", "2", "2"], out
    env["GPU_MAX_HW_QUEUES"] = "3"
    code = "import sys, os; sys.path.insert(0, %r); import halo_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % ROOT
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip() == "3"


def test_product_abi_has_no_measurement_aids():
    """include/halo_hip.h keeps only entry points that replace a reference call (+ halo_event_*); the HBM
    probes and the contiguous-range allocator live in tools/libhalo_probe.so, which nothing under halo_amd/ loads."""
    syms = _declared_symbols()
    for gone in ("halo_pool_alloc", "halo_pool_free", "halo_pool_alloc_stats", "halo_hbm_read_probe", "halo_hbm_walk_probe"):
        assert gone not in syms
    import halo_amd.pool as pool
    for gone in ("alloc_contiguous", "probe_streaming", "contiguous_memory_stats"):
        assert not hasattr(pool, gone)
    import subprocess
    r = subprocess.run(["grep", "-rl", "halo_probe", os.path.join(ROOT, "halo_amd"), "--include=*.py", "--include=*.hip", "--include=*.hpp"],
                       capture_output=True, text=True)
    assert r.stdout.strip() in ("", os.path.join(ROOT, "halo_amd", "csrc", "halo_pool.hip")), r.stdout   # one comment pointing at tools/
    from tools import halo_probe
    h = ctypes.CDLL(halo_probe.build())
    for s_ in ("halo_hbm_read_probe", "halo_hbm_walk_probe", "halo_pool_alloc", "halo_pool_free", "halo_pool_alloc_stats"):
        assert hasattr(h, s_)


def test_writer_threads_follow_the_ranks_share_of_the_host(monkeypatch):
    from halo_amd import _host
    monkeypatch.setattr(_host, "usable_cpus", lambda: 16)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _host.host_threads_per_rank(cap=8) == 2                   # 8 ranks on a 16-core quota: 2 writers each
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert _host.host_threads_per_rank(cap=8) == 8 and _host.host_threads_per_rank() == 16
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "64")
    assert _host.host_threads_per_rank(cap=8) == 1


def test_native_png_encoder_around_the_match_length_limits(tmp_path):
    """The run-length encoder splits a run into deflate matches of at most 258 bytes and must leave no 1- or 2-byte tail a match
    cannot take: rows built from runs of 1-5, 257-262, 516-520 bytes and whole rows, at widths on both sides of 258 / 516 / 774
    / 1032, the filter byte joining a leading run of zeros -- decoded by PIL, pixel for pixel."""
    from PIL import Image
    from halo_amd import _hostlib
    rng = np.random.default_rng(0)
    p = str(tmp_path / "a.png")
    for trial in range(150):
        H = int(rng.integers(1, 7))
        W = int(rng.choice([1, 2, 3, 7, 255, 256, 257, 258, 259, 260, 261, 262, 515, 516, 517, 518, 519, 520, 521, 774, 775, 776, 777, 1031, 1033, 2048]))
        a = np.empty((H, W), np.uint8)
        for y in range(H):
            x = 0
            while x < W:
                L = int(rng.choice([1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 516, 517, 518, 519, 520, W]))
                a[y, x:x + L] = int(rng.choice([0, 0, 255, 255, int(rng.integers(0, 256))]))
                x += L
        _hostlib.png_gray8_write(p, a)
        b = np.array(Image.open(p))
        assert b.shape == a.shape and np.array_equal(a, b), (trial, H, W)
