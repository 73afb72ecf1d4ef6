"""Child process of tests/test_sanitizers.py: drives every entry point of a SANITIZER build of libhalo_host.so (path in argv[1])
with numpy + ctypes only (no torch: the process runs under LD_PRELOAD=libasan), writing its files into argv[2]; the parent
-- an ordinary process -- then decodes them with PIL / torch.load.  argv[3]: an .npz with the indicator template of the
shape (raw bytes, payload offsets, CRC field offsets) the parent built with torch.save.  Exit code 0 = every call returned
what it should; the sanitizer's own report goes to stderr and makes the exit code non-zero."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

so, out, tpl_path = sys.argv[1], sys.argv[2], sys.argv[3]
h = C.CDLL(so)
h.halo_png_gray8_bound.restype = C.c_size_t
h.halo_png_gray8_bound.argtypes = [C.c_int64, C.c_int64]
h.halo_png_gray8_encode.restype = C.c_size_t
h.halo_png_gray8_encode.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t]
h.halo_png_gray8_write.restype = C.c_int
h.halo_png_gray8_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
h.halo_crc32.restype = C.c_uint32
h.halo_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
h.halo_compose_mask.restype = C.c_int
h.halo_compose_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]
h.halo_write_indicator.restype = C.c_int
h.halo_write_indicator.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
h.halo_retire_image.restype = C.c_int
h.halo_retire_image.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
h.halo_compose_indicators.restype = C.c_int
h.halo_compose_indicators.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
h.halo_host_thread_release.restype = None

rng = np.random.default_rng(5)

# --- CRC-32 against zlib: every length around the folding widths, unaligned starts, a running value
buf = rng.integers(0, 256, 70000, dtype=np.uint8)
for n in list(range(0, 200)) + [255, 256, 257, 511, 512, 513, 4095, 4096, 4097, 65535, 65536, 69999]:
    for off in (0, 1, 3, 7):
        if off + n > buf.size:
            continue
        got = h.halo_crc32(0, buf[off:].ctypes.data, n)
        assert got == zlib.crc32(buf[off:off + n].tobytes()), (n, off)
assert h.halo_crc32(zlib.crc32(b"abc"), buf.ctypes.data, 1000) == zlib.crc32(buf[:1000].tobytes(), zlib.crc32(b"abc"))


def decode_png(data):
    """8-bit greyscale PNG -> rows (filter type 0 or whatever the encoder chose is undone here for types 0-1 only)"""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, W = 8, b"", None
    while pos < len(data):
        ln = int.from_bytes(data[pos:pos + 4], "big")
        typ = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + ln]
        assert int.from_bytes(data[pos + 8 + ln:pos + 12 + ln], "big") == zlib.crc32(typ + body)
        if typ == b"IHDR":
            W, H = int.from_bytes(body[:4], "big"), int.from_bytes(body[4:8], "big")
            assert body[8:] == bytes([8, 0, 0, 0, 0])
        if typ == b"IDAT":
            idat += body
        pos += 12 + ln
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, W + 1)
    assert not raw[:, 0].any(), "filter type 0 expected"
    return raw[:, 1:]


# --- PNG encoder: shapes around the run-length limits, constant / random / striped images, padded rows, 1-pixel edges
shapes = [(1, 1), (1, 300), (300, 1), (7, 258), (7, 259), (5, 515), (33, 67), (64, 128), (48, 96), (2, 70000)]
for k, (H, W) in enumerate(shapes):
    for kind in range(4):
        if kind == 0:
            img = np.full((H, W), 255, np.uint8)
        elif kind == 1:
            img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        elif kind == 2:
            img = np.repeat(rng.integers(0, 19, (H, (W + 8) // 9), dtype=np.uint8), 9, axis=1)[:, :W].copy()
        else:
            img = np.full((H, W), 255, np.uint8)
            img[rng.random((H, W)) < 0.01] = 3
        cap = h.halo_png_gray8_bound(H, W)
        outb = np.empty(cap, np.uint8)
        n = h.halo_png_gray8_encode(img.ctypes.data, H, W, W, outb.ctypes.data, cap)
        assert 0 < n <= cap
        assert np.array_equal(decode_png(outb[:n].tobytes()), img), (H, W, kind)
        # (the output buffer holds exactly `bound` bytes: an overrun is the sanitizer's to see); a padded row stride; a short buffer
        pad = np.zeros((H, W + 13), np.uint8)
        pad[:, :W] = img
        assert h.halo_png_gray8_encode(pad.ctypes.data, H, W, W + 13, outb.ctypes.data, cap) == n
        assert h.halo_png_gray8_encode(img.ctypes.data, H, W, W, outb.ctypes.data, cap - 1) == 0        # too small: refused, not overrun
        assert h.halo_png_gray8_encode(img.ctypes.data, H, W, W - 1, outb.ctypes.data, cap) == 0        # stride below the width
    assert h.halo_png_gray8_write(os.path.join(out, "png_%d.png" % k).encode(), img.ctypes.data, H, W, W) == 0
assert h.halo_png_gray8_write(os.path.join(out, "no_such_dir", "x.png").encode(), img.ctypes.data, 2, 2, 2) == -2

# --- compose_mask / compose_indicators: every integer width, picks on every border and corner, k = 0
H, W = 37, 53
picks = np.array([[0, 0, 1.0], [0, W - 1, .9], [H - 1, 0, .8], [H - 1, W - 1, .7], [17, 20, .6], [1, W - 2, .5], [H // 2, 0, .4]], np.float64)
gt = rng.integers(0, 19, (H, W)).astype(np.int64)
gt[rng.random((H, W)) < 0.05] = 255
for dt in (np.uint8, np.int16, np.int32, np.int64):
    om = np.full((H, W), 255, dt)
    gtd = gt.astype(dt)                                        # (kept alive across the call)
    for radius in (0, 1, 2, 60):
        mask = np.empty((H, W), np.uint8)
        assert h.halo_compose_mask(mask.ctypes.data, om.ctypes.data, om.itemsize, gtd.ctypes.data, om.itemsize, H, W, picks.ctypes.data, len(picks), radius) == 0
        want = np.full((H, W), 255, np.uint8)
        for ph, pw, _ in picks:
            ph, pw = int(ph), int(pw)
            want[max(ph - radius, 0):ph + radius + 1, max(pw - radius, 0):pw + radius + 1] = gt[max(ph - radius, 0):ph + radius + 1, max(pw - radius, 0):pw + radius + 1].astype(np.uint8)
        assert np.array_equal(mask, want), (dt, radius)
    assert h.halo_compose_mask(mask.ctypes.data, om.ctypes.data, om.itemsize, None, 0, H, W, None, 0, 1) == 0
assert h.halo_compose_mask(mask.ctypes.data, om.ctypes.data, 3, gt.ctypes.data, 8, H, W, picks.ctypes.data, 1, 1) == -1
# the three statements of the 64-bit narrowing (AVX-512, AVX2, scalar): ragged lengths, high bytes set, unaligned starts
h.halo_low_bytes_mode.argtypes = [C.c_int]
for hh, ww in ((1, 1), (1, 31), (3, 33), (5, 67), (7, 129), (64, 130)):
    wide = rng.integers(-2 ** 62, 2 ** 62, hh * ww + 1, dtype=np.int64)
    src = wide[1:]                                              # 8-byte aligned but not 64-byte aligned
    want = (src & 0xff).astype(np.uint8).reshape(hh, ww)
    for mode in (0, 1, 2):
        h.halo_low_bytes_mode(mode)
        got = np.zeros((hh, ww), np.uint8)
        assert h.halo_compose_mask(got.ctypes.data, src.ctypes.data, 8, None, 0, hh, ww, None, 0, 1) == 0
        assert np.array_equal(got, want), (hh, ww, mode)
h.halo_low_bytes_mode(0)
pa, ps = (rng.random((H, W)) < 0.02).astype(np.uint8), np.zeros((H, W), np.uint8)
a, s = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
assert h.halo_compose_indicators(a.ctypes.data, s.ctypes.data, pa.ctypes.data, ps.ctypes.data, H, W, picks.ctypes.data, len(picks), 1, 5) == 0
wa, ws = pa.copy(), ps.copy()
for ph, pw, _ in picks:
    ph, pw = int(ph), int(pw)
    wa[max(ph - 5, 0):ph + 6, max(pw - 5, 0):pw + 6] = 1
    ws[max(ph - 1, 0):ph + 2, max(pw - 1, 0):pw + 2] = 1
assert np.array_equal(a, wa) and np.array_equal(s, ws)
assert h.halo_compose_indicators(pa.ctypes.data, ps.ctypes.data, pa.ctypes.data, ps.ctypes.data, H, W, picks.ctypes.data, 3, 1, 5) == 0     # in place

# --- write_indicator / retire_image through the parent's template (both compose modes), files left for the parent to load
t = np.load(tpl_path)
raw, shape = t["raw"], tuple(int(v) for v in t["shape"])
H, W = shape
fa, fs = t["crc_active"].astype(np.uint64), t["crc_selected"].astype(np.uint64)
act, sel = np.ascontiguousarray(rng.random(shape) < 0.3), np.ascontiguousarray(rng.random(shape) < 0.6)
assert h.halo_write_indicator(os.path.join(out, "ind_plain.pth").encode(), raw.ctypes.data, raw.size, act.ctypes.data, sel.ctypes.data, act.size,
                              int(t["off_active"]), int(t["off_selected"]), fa.ctypes.data, fs.ctypes.data) == 0
np.save(os.path.join(out, "ind_plain_active.npy"), act)
np.save(os.path.join(out, "ind_plain_selected.npy"), sel)
gt = rng.integers(0, 19, shape).astype(np.int64)
om = np.full(shape, 255, np.int64)
picks = np.array([[0, 0, 1.0], [H - 1, W - 1, .9], [H // 2, W // 3, .8], [3, W - 1, .7]], np.float64)
for mode, cmr in (("results", -1), ("compose", 5)):
    rc = h.halo_retire_image(os.path.join(out, "ret_%s.png" % mode).encode(), os.path.join(out, "ret_%s.pth" % mode).encode(), om.ctypes.data, 8,
                             gt.ctypes.data, 8, H, W, picks.ctypes.data, len(picks), 1, act.ctypes.data, sel.ctypes.data, cmr,
                             raw.ctypes.data, raw.size, int(t["off_active"]), int(t["off_selected"]), fa.ctypes.data, fs.ctypes.data)
    assert rc == 0, (mode, rc)
# no indicator (tpl NULL), zero picks
assert h.halo_retire_image(os.path.join(out, "ret_noind.png").encode(), None, om.ctypes.data, 8, gt.ctypes.data, 8, H, W, picks.ctypes.data, 0, 1,
                           act.ctypes.data, sel.ctypes.data, -1, None, 0, 0, 0, None, None) == 0
np.savez(os.path.join(out, "retire_inputs.npz"), gt=gt, picks=picks, act=act, sel=sel)
h.halo_host_thread_release()
h.halo_host_thread_release()                                   # twice: idempotent
print("driver ok")
