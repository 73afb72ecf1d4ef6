"""Generates the 128-entry (r_j, L_j) table of the numeric contract's float32 logarithm (oracle/halo_oracle_math.h `ho_logf`,
halo_amd/csrc/halo_devmath.hpp `det_logf_core`) with mpmath, as C hexadecimal floating constants.

    bin j   mantissas m in [sqrt(1/2), sqrt(2)) whose bits lie in [B + j 2^16, B + (j + 1) 2^16),  B = bits(sqrt(1/2)) = 0x3f3504f3
    r_j     1 / (centre of the bin) rounded to 16 significant bits (so m r_j is exact in binary64); exactly 1 for the bin holding m = 1
    L_j     -log(r_j) rounded to binary64 (exactly 0 for that bin)

The binary64 logarithm of the contract (`ho_log_cr` / `det_log_cr_core`: geoopt's artanh, core/utils/hyperbolic.py:83) has its own
table, 256 bins over the same mantissa range indexed by the high word:

    r_j     1 / (centre of the bin) on a 9-bit grid (multiples of 2^-8 above 1, of 2^-9 below), so that z = m r_j - 1 is EXACT in
            binary64 and |z| < 2^-8 over the whole bin (both asserted here in rational arithmetic); exactly 1 for the bin holding m = 1
    L_j     -log(r_j) as a double-double (hi, lo)

    python tools/gen_logf_table.py            prints the tables; the two headers hold these lines verbatim
    python tools/gen_logf_table.py --check    compares with what the two headers hold (tests/test_aten_exact.py runs this)
"""
import re
import struct
import sys

import mpmath as mp

B = 0x3F3504F3
mp.mp.prec = 200


def f32(bits):
    return mp.mpf(struct.unpack("<f", struct.pack("<I", bits))[0])


def round_sig(x, bits):
    m, e = mp.frexp(x)                       # x = m 2^e, m in [0.5, 1)
    return mp.ldexp(mp.nint(mp.ldexp(m, bits)), e - bits)


def table():
    rows = []
    one = 0x3F800000
    for j in range(128):
        lo = B + (j << 16)
        if lo <= one < lo + (1 << 16):
            r, L = mp.mpf(1), mp.mpf(0)
        else:
            r = round_sig(1 / f32(lo + (1 << 15)), 16)
            L = -mp.log(r)
        rows.append((float(r), float(L)))    # float(): correctly rounded to binary64
    return rows


def lines():
    return ["    {%s, %s}," % (float.hex(r), float.hex(L)) for r, L in table()]


B64 = 0x3FE6A09E


def f64_hw(hw):
    return struct.unpack("<d", struct.pack("<II", 0, hw))[0]


def table64():
    from fractions import Fraction
    rows = []
    for j in range(256):
        lo, hi = Fraction(f64_hw(B64 + j * 4096)), Fraction(f64_hw(B64 + (j + 1) * 4096))
        if lo <= 1 < hi:
            r = Fraction(1)
        else:
            inv = 1 / ((lo + hi) / 2)
            step = Fraction(1, 256) if inv > 1 else Fraction(1, 512)
            r = round(inv / step) * step
        for m in (lo, hi - Fraction(1, 2 ** 60)):
            z = abs(m * r - 1)
            assert z < Fraction(1, 256), (j, float(z))                       # => m r - 1 fits 53 bits: exact under one fma
        L = -mp.log(mp.mpf(r.numerator) / mp.mpf(r.denominator))
        Lh = float(L)
        rows.append((float(r), Lh, float(L - mp.mpf(Lh))))
    return rows


def lines64():
    return ["    {%s, %s, %s}," % (float.hex(r), float.hex(h), float.hex(l)) for r, h, l in table64()]


def held(path, tag="LOGF_TABLE"):
    txt = open(path).read()
    m = re.search(tag + r"_BEGIN.*?\n(.*?)\n[^\n]*" + tag + "_END", txt, re.S)
    return [l.rstrip() for l in m.group(1).splitlines()]


if __name__ == "__main__":
    if "--check" in sys.argv:
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        want, want64 = lines(), lines64()
        for p in ("oracle/halo_oracle_math.h", "halo_amd/csrc/halo_devmath.hpp"):
            assert held(os.path.join(root, p)) == want, p + ": float32 table differs from the generator's"
            assert held(os.path.join(root, p), "LOG64_TABLE") == want64, p + ": binary64 table differs from the generator's"
        print("both headers hold the generated tables (128 + 256 rows)")
    elif "--f64" in sys.argv:
        print("\n".join(lines64()))
    else:
        print("\n".join(lines()))
