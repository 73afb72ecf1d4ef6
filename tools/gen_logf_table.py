"""Generates the 128-entry (r_j, L_j) table of the numeric contract's float32 logarithm (oracle/halo_oracle_math.h `ho_logf`,
halo_amd/csrc/halo_devmath.hpp `det_logf_core`) with mpmath, as C hexadecimal floating constants.

    bin j   mantissas m in [sqrt(1/2), sqrt(2)) whose bits lie in [B + j 2^16, B + (j + 1) 2^16),  B = bits(sqrt(1/2)) = 0x3f3504f3
    r_j     1 / (centre of the bin) rounded to 16 significant bits (so m r_j is exact in binary64); exactly 1 for the bin holding m = 1
    L_j     -log(r_j) rounded to binary64 (exactly 0 for that bin)

    python tools/gen_logf_table.py            prints both tables; the two headers hold these lines verbatim
    python tools/gen_logf_table.py --check    compares with what the two headers hold (tests/test_oracle_kat.py runs this)
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


def held(path):
    txt = open(path).read()
    m = re.search(r"LOGF_TABLE_BEGIN.*?\n(.*?)\n[^\n]*LOGF_TABLE_END", txt, re.S)
    return [l.rstrip() for l in m.group(1).splitlines()]


if __name__ == "__main__":
    if "--check" in sys.argv:
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        want = lines()
        for p in ("oracle/halo_oracle_math.h", "halo_amd/csrc/halo_devmath.hpp"):
            got = held(os.path.join(root, p))
            assert got == want, p + ": table differs from the generator's"
        print("both headers hold the generated table (128 rows)")
    else:
        print("\n".join(lines()))
