// Elementary functions of the HALO-AMD numeric contract for gfx950 device code.
//
// The scoring maps feed an argmax-driven greedy selector whose output must be bit-identical
// between runs, ranks and the CPU checker -- and, as far as the reference's CPU libraries allow,
// to the reference's own maps -- so exp/log are not taken from the device libm.  Each is a fixed
// sequence of IEEE-754 operations with every fused multiply-add written out; the translation unit is built
// with -ffp-contract=off so nothing else is contracted.  sqrt and '/' are the correctly
// rounded device instruction sequences hipcc emits by default.
//
//   expf   Sleef's expf_u10 (xexpf, sleefsimdsp.c): what ATen's Vectorized<float>::exp() evaluates inside
//          torch.softmax on AVX2 / AVX-512 hosts (core/active/floating_region.py:152) -- same constants, same
//          fma chain, same two-step scaling, hence torch.softmax's bits.
//   logf   the correctly rounded natural logarithm, through binary64: 128-entry (r_j, -log r_j) table over the
//          mantissa range [sqrt(1/2), sqrt(2)), z = m r_j - 1 exactly, degree-6 series of log1p, e ln 2 added, one
//          rounding to float32 (correctly rounded for every positive normal float32 but four above 5e7).  Stands in
//          for torch.log (floating_region.py:72,119), which is MKL's closed-source vsLn: its AVX-512 path differs from
//          the correctly rounded value in 0.005 % of softmax probabilities, its AVX2 path in 7 % (oracle/halo_oracle_math.h).
//   log    fdlibm-style binary64 log for geoopt's artanh (dist0, core/utils/hyperbolic.py:83).
#pragma once
#ifndef HALO_DEVMATH_HOST_CHECK      // tests/native/devmath_host_check.cpp evaluates these recipes on the host
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace halo {

__device__ __forceinline__ float pow2f_(int k) { return __uint_as_float((uint32_t)(k + 127) << 23); }

// All of these are straight-line code: the special cases (NaN, out of range, zero, infinity) are patched in with selects
// at the end instead of returning early, so that a pixel's classes compile into one basic block the scheduler can
// interleave and pack (early returns cost three scalar exec-mask instructions per test and a pipeline bubble per block:
// ~600 branches per fused-entropy pass).  In-range inputs take exactly the operations of the oracle's branchy statement
// (oracle/halo_oracle_math.h); tests/native/devmath_host_check.cpp compares the two on the host, bit for bit.
// The *_core functions are the main path alone, for callers that have already excluded the special cases.

// expf for finite x in [-104, 100]
__device__ __forceinline__ float det_expf_core(float x)
{
    float k = __builtin_rintf(x * 1.44269502162933349609375f);     // R_LN2f
    float r = __builtin_fmaf(k, -0.693145751953125f, x);           // -L2Uf
    r = __builtin_fmaf(k, -1.428606765330187045e-06f, r);          // -L2Lf
    float p = 0.000198527617612853646278381f;
    p = __builtin_fmaf(p, r, 0.00139304355252534151077271f);
    p = __builtin_fmaf(p, r, 0.00833336077630519866943359f);
    p = __builtin_fmaf(p, r, 0.0416664853692054748535156f);
    p = __builtin_fmaf(p, r, 0.166666671633720397949219f);
    p = __builtin_fmaf(p, r, 0.5f);
    float y = 1.0f + __builtin_fmaf(r * r, p, r);
    int ki = (int)k;
    int k1 = ki >> 1;
    int k2 = ki - k1;
    return (y * pow2f_(k1)) * pow2f_(k2);
}

// expf for x in [-87, 0.35] (the lean softmax hands in x - max, in [-64, 0]): det_expf_core with its last step -- the scaling
// by 2^k in two exact halves, needed where y 2^k leaves the normal range -- as ONE exact scaling.  Here y is in [0.5, 2) and k in
// [-126, 0], so y 2^k is a normal float either way: the same bits.
// Device form (round 4, after tools/micro/op_rate.hip: on gfx950 v_fma / v_mul / v_add_f32 and v_add_u32 issue in 2 cycles per
// wave, every other VALU instruction -- v_rndne, v_cvt, v_ldexp, v_frexp, v_cmp, v_cndmask, v_max -- in 4, v_rcp / v_exp in 8):
// the rounding to an integer is two additions with 1.5 * 2^23 (t + M rounds t to the nearest integer, ties to even, exactly like
// rintf for |t| < 2^22; t lies in [-127, 1]) and the scaling adds k to y's exponent field -- k sits in the low mantissa bits of
// t + M in two's complement, so (bits(t + M) << 23) is k << 23 modulo 2^32 (the constant's own bits shift out): ONE
// v_lshl_add_u32 where v_rndne + v_cvt_i32 + v_ldexp were three 4-cycle instructions.  Same values at every step.
__device__ __forceinline__ float det_expf_core_small(float x)
{
    const float t = x * 1.44269502162933349609375f;
    const float tm = t + 12582912.0f;
    float k = tm - 12582912.0f;                      // == rintf(t)
    float r = __builtin_fmaf(k, -0.693145751953125f, x);
    r = __builtin_fmaf(k, -1.428606765330187045e-06f, r);
    float p = 0.000198527617612853646278381f;
    p = __builtin_fmaf(p, r, 0.00139304355252534151077271f);
    p = __builtin_fmaf(p, r, 0.00833336077630519866943359f);
    p = __builtin_fmaf(p, r, 0.0416664853692054748535156f);
    p = __builtin_fmaf(p, r, 0.166666671633720397949219f);
    p = __builtin_fmaf(p, r, 0.5f);
    float y = 1.0f + __builtin_fmaf(r * r, p, r);
    return __uint_as_float(__float_as_uint(y) + (__float_as_uint(tm) << 23));      // == ldexpf(y, (int)k)
}

__device__ __forceinline__ float det_expf(float x)
{
    // The reduction runs on a clamped copy (in-range x is unchanged; NaN becomes a bound), so every intermediate is finite.
    // Sleef patches x < -104 to 0 and x > 100 to +inf after the scaling; the clamped value lands there by itself:
    // y * 2^72 * 2^72 overflows at xc = 100, y * 2^-75 * 2^-75 with y < 1 rounds to zero at xc = -104 -- the host check
    // walks every float32 input beyond the cut-offs.
    const float res = det_expf_core(__builtin_fminf(__builtin_fmaxf(x, -104.0f), 100.0f));
    return x != x ? x : res;
}

// (r_j, L_j = -log r_j) of the float32 logarithm, tools/gen_logf_table.py: 2 KB, read per lane (it stays in the L1 / L2)
static __device__ const double logf_tab_[128][2] = {
    // LOGF_TABLE_BEGIN
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
    // LOGF_TABLE_END
};

// logf of the positive normal float whose bits are u, plus e0 * ln 2: v = bits - bits(sqrt(1/2)) holds the exponent of the
// mantissa range [sqrt(1/2), sqrt(2)) in its top 9 bits (arithmetic shift) and the table index in the next 7.
__device__ __forceinline__ float logf_core_(uint32_t u, int e0)
{
    const uint32_t v = u - 0x3f3504f3u;
    const int e = e0 + ((int)v >> 23);
    const float m = __uint_as_float(u - (v & 0xff800000u));
    const int j = (int)((v >> 16) & 0x7fu);
    const double r = logf_tab_[j][0], L = logf_tab_[j][1];
    const double z = __builtin_fma((double)m, r, -1.0);              // exact: 24 x 16 bits
    const double z2 = z * z;
    double q = -0x1.5555555555555p-3;                                // -1/6
    q = __builtin_fma(q, z, 0.2);
    q = __builtin_fma(q, z, -0.25);
    q = __builtin_fma(q, z, 0x1.5555555555555p-2);                   // 1/3
    q = __builtin_fma(q, z, -0.5);
    const double p = __builtin_fma(z2, q, z);
    const double y = __builtin_fma((double)e, 0x1.62e42fefa39efp-1, L);   // ln 2
    return (float)(y + p);
}

// logf for positive normal finite x
__device__ __forceinline__ float det_logf_core(float x) { return logf_core_(__float_as_uint(x), 0); }

__device__ __forceinline__ float det_logf(float x)
{
    const uint32_t u0 = __float_as_uint(x);
    const bool sub = u0 < 0x00800000u;                       // positive subnormal (or +0, patched below)
    const float xs = sub ? x * 8388608.0f : x;
    float res = logf_core_(__float_as_uint(xs), sub ? -23 : 0);
    const float ninf = __uint_as_float(0xff800000u), qnan = __uint_as_float(0x7fc00000u);   // named, so that clang emits selects
    res = u0 == 0x7f800000u ? x : res;
    res = x == 0.0f ? ninf : res;
    res = x < 0.0f ? qnan : res;
    return x != x ? x : res;
}

// log of the positive normal double whose bits are u, plus k0 * ln 2
__device__ __forceinline__ double log_core_(uint64_t u, int k0)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = k0;
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    double f = __longlong_as_double((long long)u) - 1.0;
    double hfsq = (0.5 * f) * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return __builtin_fma(dk, ln2_hi, (f - (hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo))));
}

// log for positive normal finite x
__device__ __forceinline__ double det_log_core(double x) { return log_core_((uint64_t)__double_as_longlong(x), 0); }

__device__ __forceinline__ double det_log(double x)
{
    const uint64_t u0 = (uint64_t)__double_as_longlong(x);
    const bool sub = u0 < 0x0010000000000000ull;             // positive subnormal (or +0, patched below)
    const double xs = sub ? x * 18014398509481984.0 : x;
    double res = log_core_((uint64_t)__double_as_longlong(xs), sub ? -54 : 0);
    const double ninf = __longlong_as_double(0xfff0000000000000ll), qnan = __longlong_as_double(0x7ff8000000000000ll);
    res = u0 == 0x7ff0000000000000ull ? x : res;
    res = x == 0.0 ? ninf : res;
    res = x < 0.0 ? qnan : res;
    return x != x ? x : res;
}

// asinh for the HyperMLR epilogue: sign(x) * log1p(t),  t = |x| + x^2 / (1 + sqrt(1 + x^2))  (= |x| + sqrt(1+x^2) - 1
// without cancellation), log1p(t) = log(u) + (t - (u - 1)) / u with u = 1 + t.  One formula for every magnitude the
// logits can reach (|x| < 1e150), about half the instructions of the library's asinh, error <= 2 ulp.
__device__ __forceinline__ double asinh_det(double x)
{
    const double a = __builtin_fabs(x), a2 = a * a;
    const double t = a + a2 / (1.0 + __builtin_sqrt(1.0 + a2));
    const double u = 1.0 + t;
    const double r = det_log_core(u) + (t - (u - 1.0)) / u;    // u >= 1; for infinite or NaN u the second term is NaN anyway
    return x != x ? x : __builtin_copysign(r, x);
}

// geoopt artanh: clamp to +-(1-1e-7), 0.5*(log(1+z) - log(1-z)) in float64
__device__ __forceinline__ double artanh_clamped(double z)
{
    const double lim = 1.0 - 1e-7;
    if (z > lim) z = lim;
    if (z < -lim) z = -lim;
    const double r = (det_log_core(1.0 + z) - det_log_core(1.0 - z)) * 0.5;    // both arguments in [1e-7, 2)
    return z != z ? z : r;
}

// geoopt dist0 = 2 * artan_k(||x||), k = -c:  ks = sqrt(|k| + 1e-15), rks = 1/ks (host doubles)
__device__ __forceinline__ double dist0_from_ssq(double ssq, double ks, double rks)
{
    return 2.0 * (rks * artanh_clamped(__builtin_sqrt(ssq) * ks));
}
// float32 tensor: all of it in float32 -- geoopt's stereographic artanh takes its two logs in the input dtype
// (x.clamp(-1+1e-7, 1-1e-7); 0.5 * (log(1 + x) - log(1 - x))); both arguments are positive normal floats in [2^-23, 2)
__device__ __forceinline__ float dist0_from_ssq(float ssq, double ks, double rks)
{
    float n = __builtin_sqrtf(ssq);
    float z = n * (float)ks;
    const float lim = (float)(1.0 - 1e-7);
    if (z > lim) z = lim;
    if (z < -lim) z = -lim;
    float a = (det_logf_core(1.0f + z) - det_logf_core(1.0f - z)) * 0.5f;
    a = z != z ? z : a;
    return 2.0f * ((float)rks * a);
}

// Bilinear combination of the four taps of one output element, F.interpolate(mode='bilinear', align_corners=True)
// (build.py:123-135; classifier.py:375-377,556-557) in ATen's order -- columns first, rows second, each p*q + r*s as
// fma(p, q, r*s): bit for bit torch's CPU kernel at the shapes the path runs (oracle/halo_oracle.c, halo_o_bilinear_*).
__device__ __forceinline__ double col_lerp(double lx0, double lx1, double v0, double v1) { return __builtin_fma(lx0, v0, lx1 * v1); }
__device__ __forceinline__ float col_lerp(float lx0, float lx1, float v0, float v1) { return __builtin_fmaf(lx0, v0, lx1 * v1); }
template <typename T>
__device__ __forceinline__ T bilerp(T v00, T v01, T v10, T v11, T lx0, T lx1, T ly0, T ly1)
{
    return col_lerp(ly0, ly1, col_lerp(lx0, lx1, v00, v01), col_lerp(lx0, lx1, v10, v11));
}

}  // namespace halo
