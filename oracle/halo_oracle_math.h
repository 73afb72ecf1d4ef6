/* TEST INFRASTRUCTURE -- not part of the product.  Only tests/, the smoke check
 * and bench.py's cpu_baseline leg may build, load or call anything under oracle/.
 *
 * Elementary functions of the HALO-AMD numeric contract (DESIGN.md "Numeric
 * contract"), written out as explicit IEEE-754 operation sequences so that this
 * CPU restatement and the HIP kernels (which carry their OWN copy of the same
 * published recipe, halo_amd/csrc/halo_devmath.hpp) produce bit-identical maps and
 * therefore bit-identical selected-pixel indices at any image size.
 *
 * Build with -ffp-contract=off: every fused multiply-add below is an explicit
 * fma()/fmaf(); nothing else may be contracted.
 *
 * expf is Sleef's expf_u10 (sleefsimdsp.c, xexpf), the function ATen's Vectorized<float>::exp() calls inside
 * torch.softmax on AVX2 and AVX-512 hosts (floating_region.py:152): range reduction with round-to-nearest-even,
 * a degree-6 polynomial in fma form, scaling by 2^q in two halves.  Bit for bit Sleef_expf8_u10 / Sleef_expf16_u10
 * of the libtorch in the build container (tests/test_aten_exact.py calls them), hence bit for bit torch.softmax.
 *
 * logf is the correctly rounded natural logarithm, computed in binary64: 128-entry table of (r_j, L_j = -log r_j)
 * over the mantissa range [sqrt(1/2), sqrt(2)) (tools/gen_logf_table.py), z = m r_j - 1 exactly, degree-6 Taylor
 * series of log1p, e ln 2 + L_j added, one rounding to float32.  Checked against every positive normal float32:
 * correctly rounded everywhere except four inputs above 5e7 (0x4c5d65a5, 0x4d604ebe, 0x65d890d3, 0x6f31a8ec; one
 * ulp), none below.  It stands in for torch.log (floating_region.py:72,119; geoopt's float32 artanh), which is
 * MKL's VML vsLn in "high accuracy" mode on the reference's CPU path: closed source, and NOT one function -- its
 * AVX2 and AVX-512 code paths differ from each other in 7 % of the values (measured in the build container with
 * MKL_ENABLE_INSTRUCTIONS).  The AVX-512 path, which this container runs, differs from the correctly rounded value
 * in 0.005 % of the values on softmax probabilities and in none of the 17 window fractions k/9, k/6, k/4 of
 * compute_region_impurity; the correctly rounded logarithm is the one target every such library approximates.
 * log (binary64) follows the classic fdlibm recipe (k*ln2 + log1p-style series in s = f/(2+f), < 1 ulp).
 */
#ifndef HALO_ORACLE_MATH_H
#define HALO_ORACLE_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float ho_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ho_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double ho_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline uint64_t ho_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

/* 2^k for -126 <= k <= 127 */
static inline float ho_pow2f(int k) { return ho_u2f((uint32_t)(k + 127) << 23); }

static inline float ho_expf(float x)
{
    if (x != x) return x;
    if (x < -104.0f) return 0.0f;                                      /* Sleef's two patches (it applies them after the scaling: same values) */
    if (x > 100.0f) return INFINITY;
    const int q = (int)rintf(x * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = fmaf((float)q, -0.693145751953125f, x);                 /* -L2Uf */
    s = fmaf((float)q, -1.428606765330187045e-06f, s);                /* -L2Lf */
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    const int q1 = q >> 1;                                             /* vldexp2: both factors stay normal */
    return (u * ho_pow2f(q1)) * ho_pow2f(q - q1);
}

/* (r_j, L_j) of ho_logf, tools/gen_logf_table.py */
static const double ho_logf_tab[128][2] = {
    /* LOGF_TABLE_BEGIN */
    {0x1.690a000000000p+0, -0x1.600f644134de3p-2},
    {0x1.6710000000000p+0, -0x1.5a704d57479e1p-2},
    {0x1.651c000000000p+0, -0x1.54da79650e302p-2},
    {0x1.632c000000000p+0, -0x1.4f48565f7917cp-2},
    {0x1.6142000000000p+0, -0x1.49bfcae2a8e33p-2},
    {0x1.5f5c000000000p+0, -0x1.443b35956b7f4p-2},
    {0x1.5d7e000000000p+0, -0x1.3ec669eed5a1dp-2},
    {0x1.5ba2000000000p+0, -0x1.395006f19e5e7p-2},
    {0x1.59ce000000000p+0, -0x1.33e9d4e2a3866p-2},
    {0x1.57fc000000000p+0, -0x1.2e82436cb81b9p-2},
    {0x1.5630000000000p+0, -0x1.29254f4ce05bcp-2},
    {0x1.546a000000000p+0, -0x1.23d32d42b6664p-2},
    {0x1.52a8000000000p+0, -0x1.1e860630285d0p-2},
    {0x1.50ea000000000p+0, -0x1.193df76c7b4d9p-2},
    {0x1.4f30000000000p+0, -0x1.13fb1e95b94ccp-2},
    {0x1.4d7c000000000p+0, -0x1.0ec3bdbb922f6p-2},
    {0x1.4bcc000000000p+0, -0x1.0991dee394341p-2},
    {0x1.4a20000000000p+0, -0x1.0465a08154ffap-2},
    {0x1.4878000000000p+0, -0x1.fe7e42966d65bp-3},
    {0x1.46d4000000000p+0, -0x1.f43d00730a0f1p-3},
    {0x1.4536000000000p+0, -0x1.ea145160786c6p-3},
    {0x1.439a000000000p+0, -0x1.dfeb53af840b2p-3},
    {0x1.4204000000000p+0, -0x1.d5db877180337p-3},
    {0x1.4070000000000p+0, -0x1.cbcbcbf30fde2p-3},
    {0x1.3ee2000000000p+0, -0x1.c1d5e234dae77p-3},
    {0x1.3d56000000000p+0, -0x1.b7e06a753ed33p-3},
    {0x1.3bce000000000p+0, -0x1.adf86e4c0313ap-3},
    {0x1.3a4a000000000p+0, -0x1.a41e2f79351f4p-3},
    {0x1.38ca000000000p+0, -0x1.9a51f02b9e008p-3},
    {0x1.374e000000000p+0, -0x1.9093f2fdd1fa7p-3},
    {0x1.35d6000000000p+0, -0x1.86e47af32007ap-3},
    {0x1.3460000000000p+0, -0x1.7d36832b8f0e3p-3},
    {0x1.32ee000000000p+0, -0x1.739777cb5e107p-3},
    {0x1.3180000000000p+0, -0x1.6a079d0f7aad2p-3},
    {0x1.3016000000000p+0, -0x1.60873792e32c6p-3},
    {0x1.2eae000000000p+0, -0x1.570904074ef49p-3},
    {0x1.2d4a000000000p+0, -0x1.4d9ab018fd3cep-3},
    {0x1.2be8000000000p+0, -0x1.442ed9346826ap-3},
    {0x1.2a8c000000000p+0, -0x1.3ae106130c54fp-3},
    {0x1.2930000000000p+0, -0x1.3188543c098a1p-3},
    {0x1.27da000000000p+0, -0x1.284e3361e2809p-3},
    {0x1.2684000000000p+0, -0x1.1f0961c1b6b1ap-3},
    {0x1.2534000000000p+0, -0x1.15e3afbc9688fp-3},
    {0x1.23e6000000000p+0, -0x1.0cc184809a5dbp-3},
    {0x1.229a000000000p+0, -0x1.03a2f832b9650p-3},
    {0x1.2152000000000p+0, -0x1.f52c9715088f1p-4},
    {0x1.200c000000000p+0, -0x1.e31b1dff3a3d6p-4},
    {0x1.1eca000000000p+0, -0x1.d12e47d16dc4dp-4},
    {0x1.1d8a000000000p+0, -0x1.bf49f6b2cbd0ap-4},
    {0x1.1c4c000000000p+0, -0x1.ad6e5ded70eefp-4},
    {0x1.1b12000000000p+0, -0x1.9bb8a1fa99d4bp-4},
    {0x1.19da000000000p+0, -0x1.8a0c46b611fd8p-4},
    {0x1.18a6000000000p+0, -0x1.7886b1bb4da18p-4},
    {0x1.1772000000000p+0, -0x1.66edd76c35b44p-4},
    {0x1.1642000000000p+0, -0x1.557c6f14d483fp-4},
    {0x1.1516000000000p+0, -0x1.44330f676bcf5p-4},
    {0x1.13ec000000000p+0, -0x1.32f49edb8bdccp-4},
    {0x1.12c4000000000p+0, -0x1.21c1552cbe640p-4},
    {0x1.119e000000000p+0, -0x1.10996a8d2f571p-4},
    {0x1.107a000000000p+0, -0x1.fefa2f4a6e1cbp-5},
    {0x1.0f5a000000000p+0, -0x1.dd158c7443c83p-5},
    {0x1.0e3a000000000p+0, -0x1.bb0cdd7b37edbp-5},
    {0x1.0d1e000000000p+0, -0x1.9959991defffdp-5},
    {0x1.0c06000000000p+0, -0x1.77fcf47faaad7p-5},
    {0x1.0aee000000000p+0, -0x1.567d63556fdf4p-5},
    {0x1.09d8000000000p+0, -0x1.35183dc34b08cp-5},
    {0x1.08c6000000000p+0, -0x1.140bdcf13b1bep-5},
    {0x1.07b4000000000p+0, -0x1.e5ba6e56885b6p-6},
    {0x1.06a6000000000p+0, -0x1.a411912616526p-6},
    {0x1.059a000000000p+0, -0x1.62a254a29b594p-6},
    {0x1.0490000000000p+0, -0x1.216daf6d9321ap-6},
    {0x1.0388000000000p+0, -0x1.c0e9338c24217p-7},
    {0x1.0282000000000p+0, -0x1.3f701b07cff62p-7},
    {0x1.017e000000000p+0, -0x1.7ce4184a28d45p-8},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.fdee000000000p-1, 0x1.0989877101c47p-8},
    {0x1.f9fe000000000p-1, 0x1.82c60f07ba2d3p-7},
    {0x1.f61e000000000p-1, 0x1.3f578ffbf5b23p-6},
    {0x1.f24c000000000p-1, 0x1.bc798ee257f83p-6},
    {0x1.ee8a000000000p-1, 0x1.1c3ffe4d08ba6p-5},
    {0x1.ead6000000000p-1, 0x1.59d2a08328007p-5},
    {0x1.e730000000000p-1, 0x1.96f1364ef38fap-5},
    {0x1.e398000000000p-1, 0x1.d3983dcb58901p-5},
    {0x1.e00c000000000p-1, 0x1.07f3263b25d88p-4},
    {0x1.dc8e000000000p-1, 0x1.25db15e6293b9p-4},
    {0x1.d91e000000000p-1, 0x1.43821e99d9cf5p-4},
    {0x1.d5ba000000000p-1, 0x1.60f7dd307fc30p-4},
    {0x1.d262000000000p-1, 0x1.7e3ad97f10026p-4},
    {0x1.cf16000000000p-1, 0x1.9b49971bf0bc9p-4},
    {0x1.cbd6000000000p-1, 0x1.b822957ad6129p-4},
    {0x1.c8a2000000000p-1, 0x1.d4c4500ab66bap-4},
    {0x1.c57a000000000p-1, 0x1.f12d3e55e1debp-4},
    {0x1.c25c000000000p-1, 0x1.06b7025c3209bp-3},
    {0x1.bf4a000000000p-1, 0x1.14b991505193cp-3},
    {0x1.bc40000000000p-1, 0x1.22aff2ddbd971p-3},
    {0x1.b944000000000p-1, 0x1.307de291d07edp-3},
    {0x1.b650000000000p-1, 0x1.3e3e6c21234d3p-3},
    {0x1.b366000000000p-1, 0x1.4be7b85d111c9p-3},
    {0x1.b086000000000p-1, 0x1.597926c83d881p-3},
    {0x1.adb0000000000p-1, 0x1.66f21552ea96ep-3},
    {0x1.aae4000000000p-1, 0x1.7451e066def93p-3},
    {0x1.a820000000000p-1, 0x1.81a18b4220535p-3},
    {0x1.a566000000000p-1, 0x1.8ed6e70c7b36dp-3},
    {0x1.a2b6000000000p-1, 0x1.9bf14bd76ab00p-3},
    {0x1.a00c000000000p-1, 0x1.a903c0f18fac1p-3},
    {0x1.9d6e000000000p-1, 0x1.b5f042c3b6f49p-3},
    {0x1.9ad6000000000p-1, 0x1.c2d3de43f7227p-3},
    {0x1.9846000000000p-1, 0x1.cfa43de7ef121p-3},
    {0x1.95c0000000000p-1, 0x1.dc56cae452f5ap-3},
    {0x1.9340000000000p-1, 0x1.e8ff2622babc7p-3},
    {0x1.90c8000000000p-1, 0x1.f592c67605d58p-3},
    {0x1.8e5a000000000p-1, 0x1.010370c1995eep-2},
    {0x1.8bf2000000000p-1, 0x1.0737ba6044b63p-2},
    {0x1.8990000000000p-1, 0x1.0d6615f4ba783p-2},
    {0x1.8736000000000p-1, 0x1.13891caeabd3bp-2},
    {0x1.84e4000000000p-1, 0x1.19a08b5b0757ep-2},
    {0x1.829a000000000p-1, 0x1.1fac1e4788a17p-2},
    {0x1.8054000000000p-1, 0x1.25b6398fbba47p-2},
    {0x1.7e18000000000p-1, 0x1.2baeb40b5eac8p-2},
    {0x1.7be0000000000p-1, 0x1.31a55d07a8591p-2},
    {0x1.79b0000000000p-1, 0x1.378f469437fb5p-2},
    {0x1.7786000000000p-1, 0x1.3d719ec2aa7c7p-2},
    {0x1.7562000000000p-1, 0x1.434c370b5fcd8p-2},
    {0x1.7344000000000p-1, 0x1.491ee0780df26p-2},
    {0x1.712e000000000p-1, 0x1.4ee3df7d4558fp-2},
    {0x1.6f1c000000000p-1, 0x1.54a6149c3732fp-2},
    {0x1.6d10000000000p-1, 0x1.5a5fcb795780ep-2},
    {0x1.6b0a000000000p-1, 0x1.6010d37976b67p-2},
    /* LOGF_TABLE_END */
};

/* natural log, correctly rounded (see the header comment) */
static inline float ho_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = ho_f2u(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = ho_f2u(x); e = -23; }
    const uint32_t v = u - 0x3f3504f3u;                          /* bits(sqrt(1/2)) */
    e += (int32_t)v >> 23;                                        /* x = m 2^e, m in [sqrt(1/2), sqrt(2)) */
    const float m = ho_u2f(u - (v & 0xff800000u));
    const int j = (int)((v >> 16) & 0x7fu);
    const double r = ho_logf_tab[j][0], L = ho_logf_tab[j][1];
    const double z = fma((double)m, r, -1.0);                     /* exact: 24 x 16 bits */
    const double z2 = z * z;
    double q = -0x1.5555555555555p-3;                             /* -1/6 */
    q = fma(q, z, 0.2);
    q = fma(q, z, -0.25);
    q = fma(q, z, 0x1.5555555555555p-2);                          /* 1/3 */
    q = fma(q, z, -0.5);
    const double p = fma(z2, q, z);
    const double y = fma((double)e, 0x1.62e42fefa39efp-1, L);     /* ln 2 */
    return (float)(y + p);
}

/* natural log, binary64; contract domain x > 0 finite normal */
static inline double ho_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    uint64_t u = ho_d2u(x);
    if (u < 0x0010000000000000ull) { x = x * 18014398509481984.0; u = ho_d2u(x); k = -54; }
    /* normalise to [sqrt(2)/2, sqrt(2)) */
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    double f = ho_u2d(u) - 1.0;
    double hfsq = (0.5 * f) * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return fma(dk, ln2_hi, (f - (hfsq - fma(s, hfsq + R, dk * ln2_lo))));
}

#endif
