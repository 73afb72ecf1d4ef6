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
//   log_cr binary64 log to ~2^-64 for geoopt's artanh (dist0, core/utils/hyperbolic.py:83): 256-entry table, exact reduction, the
//          leading terms carried as a double-double (round 6: torch.log of a float64 tensor is MKL's vdLn, 0.3 % from the correctly
//          rounded value; the fdlibm recipe of rounds 1-5 was 7.7 % from it, i.e. 9 % of the radius pixels differed in their last bit).
//   log    fdlibm-style binary64 log (< 1 ulp), kept for asinh.
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
// `tab`: the table to read -- logf_tab_ itself (device memory: a 16-byte gather through the vector cache per logarithm) or a copy of it
// a kernel staged in LDS (stage_logf_table): kernels that stream HBM through the same texture path read it from LDS instead.
__device__ __forceinline__ float logf_core_t(uint32_t u, int e0, const double (*tab)[2])
{
    const uint32_t v = u - 0x3f3504f3u;
    const int e = e0 + ((int)v >> 23);
    const float m = __uint_as_float(u - (v & 0xff800000u));
    const int j = (int)((v >> 16) & 0x7fu);
    const double r = tab[j][0], L = tab[j][1];
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

__device__ __forceinline__ float logf_core_(uint32_t u, int e0) { return logf_core_t(u, e0, logf_tab_); }

// logf for positive normal finite x
__device__ __forceinline__ float det_logf_core(float x) { return logf_core_(__float_as_uint(x), 0); }
__device__ __forceinline__ float det_logf_core(float x, const double (*tab)[2]) { return logf_core_t(__float_as_uint(x), 0, tab); }

#ifndef HALO_DEVMATH_HOST_CHECK
// copy of the logarithm's table in LDS: every thread of the block calls this before its first logarithm (a barrier inside)
template <int NTHREADS>
__device__ __forceinline__ void stage_logf_table(double (*lds_tab)[2])
{
    for (int j = threadIdx.x; j < 128; j += NTHREADS) { lds_tab[j][0] = logf_tab_[j][0]; lds_tab[j][1] = logf_tab_[j][1]; }
    __syncthreads();
}
#endif

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

// (r_j, L_j hi, L_j lo) of det_log_cr_core, tools/gen_logf_table.py --f64: 6 KB, read once or twice per pixel
static __device__ const double log64_tab_[256][3] = {
    // LOG64_TABLE_BEGIN
    {0x1.6a00000000000p+0, -0x1.62c82f2b9c795p-2, -0x1.7b7af915300e5p-57},
    {0x1.6900000000000p+0, -0x1.5ff3070a793d4p-2, 0x1.bc60efafc6f6ep-57},
    {0x1.6800000000000p+0, -0x1.5d1bdbf5809cap-2, -0x1.4236383dc7fe1p-56},
    {0x1.6700000000000p+0, -0x1.5a42ab0f4cfe2p-2, 0x1.8ebcb7dee9a3dp-56},
    {0x1.6600000000000p+0, -0x1.5767717455a6cp-2, -0x1.526adb283660cp-56},
    {0x1.6500000000000p+0, -0x1.548a2c3add263p-2, 0x1.819cf7e308ddbp-57},
    {0x1.6400000000000p+0, -0x1.51aad872df82dp-2, -0x1.3927ac19f55e3p-59},
    {0x1.6300000000000p+0, -0x1.4ec973260026ap-2, 0x1.42a87d977dc5ep-56},
    {0x1.6200000000000p+0, -0x1.4be5f957778a1p-2, 0x1.259b35b04813dp-57},
    {0x1.6100000000000p+0, -0x1.49006804009d1p-2, 0x1.9ffc341f177dcp-57},
    {0x1.6000000000000p+0, -0x1.4618bc21c5ec2p-2, -0x1.f42decdeccf1dp-56},
    {0x1.5f00000000000p+0, -0x1.432ef2a04e814p-2, 0x1.29931715ac903p-56},
    {0x1.5e00000000000p+0, -0x1.404308686a7e4p-2, 0x1.0bcfb6082ce6dp-56},
    {0x1.5d00000000000p+0, -0x1.3d54fa5c1f710p-2, 0x1.e3265c6a1c98dp-56},
    {0x1.5c00000000000p+0, -0x1.3a64c556945eap-2, 0x1.c68651945f97cp-57},
    {0x1.5b00000000000p+0, -0x1.3772662bfd85bp-2, 0x1.b5629d8117de7p-59},
    {0x1.5a00000000000p+0, -0x1.347dd9a987d55p-2, 0x1.4dd4c580919f8p-57},
    {0x1.5900000000000p+0, -0x1.31871c9544185p-2, 0x1.51acc4c09b379p-60},
    {0x1.5800000000000p+0, -0x1.2e8e2bae11d31p-2, 0x1.8f4cdb95ebdf9p-56},
    {0x1.5800000000000p+0, -0x1.2e8e2bae11d31p-2, 0x1.8f4cdb95ebdf9p-56},
    {0x1.5700000000000p+0, -0x1.2b9303ab89d25p-2, 0x1.896b5fd852ad4p-56},
    {0x1.5600000000000p+0, -0x1.2895a13de86a3p-2, -0x1.7ad24c13f040ep-56},
    {0x1.5500000000000p+0, -0x1.2596010df763ap-2, 0x1.0f76c57075e9ep-58},
    {0x1.5400000000000p+0, -0x1.22941fbcf7966p-2, 0x1.76f5eb09628afp-56},
    {0x1.5300000000000p+0, -0x1.1f8ff9e48a2f3p-2, 0x1.c9fdf9a0c4b07p-56},
    {0x1.5200000000000p+0, -0x1.1c898c16999fbp-2, 0x1.0e5c62aff1c44p-60},
    {0x1.5100000000000p+0, -0x1.1980d2dd4236fp-2, -0x1.9d3d1b0e4d147p-56},
    {0x1.5000000000000p+0, -0x1.1675cababa60ep-2, -0x1.ce63eab883717p-61},
    {0x1.5000000000000p+0, -0x1.1675cababa60ep-2, -0x1.ce63eab883717p-61},
    {0x1.4f00000000000p+0, -0x1.136870293a8b0p-2, -0x1.7b66298edd24ap-56},
    {0x1.4e00000000000p+0, -0x1.1058bf9ae4ad5p-2, -0x1.89fa0ab4cb31dp-58},
    {0x1.4d00000000000p+0, -0x1.0d46b579ab74bp-2, -0x1.03ec81c3cbd92p-57},
    {0x1.4c00000000000p+0, -0x1.0a324e27390e3p-2, -0x1.7dcfde8061c03p-56},
    {0x1.4b00000000000p+0, -0x1.071b85fcd590dp-2, -0x1.d1707f97bde80p-58},
    {0x1.4b00000000000p+0, -0x1.071b85fcd590dp-2, -0x1.d1707f97bde80p-58},
    {0x1.4a00000000000p+0, -0x1.0402594b4d041p-2, 0x1.28ec217a5022dp-57},
    {0x1.4900000000000p+0, -0x1.00e6c45ad501dp-2, 0x1.cb9568ff6feadp-57},
    {0x1.4800000000000p+0, -0x1.fb9186d5e3e2bp-3, 0x1.caaae64f21acbp-57},
    {0x1.4700000000000p+0, -0x1.f550a564b7b37p-3, -0x1.c5f6dfd018c37p-61},
    {0x1.4600000000000p+0, -0x1.ef0adcbdc5936p-3, -0x1.48637950dc20dp-57},
    {0x1.4600000000000p+0, -0x1.ef0adcbdc5936p-3, -0x1.48637950dc20dp-57},
    {0x1.4500000000000p+0, -0x1.e8c0252aa5a60p-3, 0x1.6e03a39bfc89bp-59},
    {0x1.4400000000000p+0, -0x1.e27076e2af2e6p-3, 0x1.61578001e0162p-59},
    {0x1.4300000000000p+0, -0x1.dc1bca0abec7dp-3, -0x1.834c51998b6fcp-57},
    {0x1.4200000000000p+0, -0x1.d5c216b4fbb91p-3, -0x1.6e443597e4d40p-57},
    {0x1.4200000000000p+0, -0x1.d5c216b4fbb91p-3, -0x1.6e443597e4d40p-57},
    {0x1.4100000000000p+0, -0x1.cf6354e09c5dcp-3, -0x1.239a07d55b695p-57},
    {0x1.4000000000000p+0, -0x1.c8ff7c79a9a22p-3, 0x1.4f689f8434012p-57},
    {0x1.3f00000000000p+0, -0x1.c2968558c18c1p-3, 0x1.73dee38a3fb6bp-57},
    {0x1.3e00000000000p+0, -0x1.bc286742d8cd6p-3, -0x1.4fce744870f55p-58},
    {0x1.3e00000000000p+0, -0x1.bc286742d8cd6p-3, -0x1.4fce744870f55p-58},
    {0x1.3d00000000000p+0, -0x1.b5b519e8fb5a4p-3, -0x1.ba27fdc19e1a0p-57},
    {0x1.3c00000000000p+0, -0x1.af3c94e80bff3p-3, 0x1.398cff3641985p-58},
    {0x1.3b00000000000p+0, -0x1.a8becfc882f19p-3, 0x1.e8c37918c39ebp-58},
    {0x1.3b00000000000p+0, -0x1.a8becfc882f19p-3, 0x1.e8c37918c39ebp-58},
    {0x1.3a00000000000p+0, -0x1.a23bc1fe2b563p-3, -0x1.93711b07a998cp-59},
    {0x1.3900000000000p+0, -0x1.9bb362e7dfb83p-3, -0x1.575e31f003e0cp-57},
    {0x1.3800000000000p+0, -0x1.9525a9cf456b4p-3, -0x1.d904c1d4e2e26p-57},
    {0x1.3800000000000p+0, -0x1.9525a9cf456b4p-3, -0x1.d904c1d4e2e26p-57},
    {0x1.3700000000000p+0, -0x1.8e928de886d41p-3, 0x1.569d851a56770p-57},
    {0x1.3600000000000p+0, -0x1.87fa06520c911p-3, 0x1.bf7fdbfa08d9ap-57},
    {0x1.3500000000000p+0, -0x1.815c0a14357ebp-3, 0x1.4be48073a0564p-58},
    {0x1.3500000000000p+0, -0x1.815c0a14357ebp-3, 0x1.4be48073a0564p-58},
    {0x1.3400000000000p+0, -0x1.7ab890210d909p-3, -0x1.be36b2d6a0608p-59},
    {0x1.3300000000000p+0, -0x1.740f8f54037a5p-3, 0x1.b264062a84cdbp-58},
    {0x1.3300000000000p+0, -0x1.740f8f54037a5p-3, 0x1.b264062a84cdbp-58},
    {0x1.3200000000000p+0, -0x1.6d60fe719d21dp-3, 0x1.caae268ecd179p-57},
    {0x1.3100000000000p+0, -0x1.66acd4272ad51p-3, 0x1.0900e4e1ea8b2p-58},
    {0x1.3000000000000p+0, -0x1.5ff3070a793d4p-3, 0x1.bc60efafc6f6ep-58},
    {0x1.3000000000000p+0, -0x1.5ff3070a793d4p-3, 0x1.bc60efafc6f6ep-58},
    {0x1.2f00000000000p+0, -0x1.59338d9982086p-3, 0x1.65d22aa8ad7cfp-58},
    {0x1.2e00000000000p+0, -0x1.526e5e3a1b438p-3, 0x1.746ff8a470d3ap-57},
    {0x1.2e00000000000p+0, -0x1.526e5e3a1b438p-3, 0x1.746ff8a470d3ap-57},
    {0x1.2d00000000000p+0, -0x1.4ba36f39a55e5p-3, -0x1.68981bcc36756p-57},
    {0x1.2c00000000000p+0, -0x1.44d2b6ccb7d1ep-3, -0x1.9f4f6543e1f88p-57},
    {0x1.2c00000000000p+0, -0x1.44d2b6ccb7d1ep-3, -0x1.9f4f6543e1f88p-57},
    {0x1.2b00000000000p+0, -0x1.3dfc2b0ecc62ap-3, 0x1.ab3a8e7d81017p-58},
    {0x1.2a00000000000p+0, -0x1.371fc201e8f74p-3, -0x1.de6cb62af18a0p-58},
    {0x1.2a00000000000p+0, -0x1.371fc201e8f74p-3, -0x1.de6cb62af18a0p-58},
    {0x1.2900000000000p+0, -0x1.303d718e47fd3p-3, 0x1.6b9c7d96091fap-63},
    {0x1.2800000000000p+0, -0x1.29552f81ff523p-3, -0x1.301771c407dbfp-57},
    {0x1.2800000000000p+0, -0x1.29552f81ff523p-3, -0x1.301771c407dbfp-57},
    {0x1.2700000000000p+0, -0x1.2266f190a5acbp-3, -0x1.f547bf1809e88p-57},
    {0x1.2600000000000p+0, -0x1.1b72ad52f67a0p-3, -0x1.483023472cd74p-58},
    {0x1.2600000000000p+0, -0x1.1b72ad52f67a0p-3, -0x1.483023472cd74p-58},
    {0x1.2500000000000p+0, -0x1.14785846742acp-3, -0x1.a28813e3a7f07p-57},
    {0x1.2400000000000p+0, -0x1.0d77e7cd08e59p-3, -0x1.9a5dc5e9030acp-57},
    {0x1.2400000000000p+0, -0x1.0d77e7cd08e59p-3, -0x1.9a5dc5e9030acp-57},
    {0x1.2300000000000p+0, -0x1.0671512ca596ep-3, -0x1.50c647eb86499p-58},
    {0x1.2200000000000p+0, -0x1.fec9131dbeabbp-4, 0x1.5746b9981b36cp-58},
    {0x1.2200000000000p+0, -0x1.fec9131dbeabbp-4, 0x1.5746b9981b36cp-58},
    {0x1.2100000000000p+0, -0x1.f0a30c01162a6p-4, -0x1.85f325c5bbacdp-58},
    {0x1.2000000000000p+0, -0x1.e27076e2af2e6p-4, 0x1.61578001e0162p-60},
    {0x1.2000000000000p+0, -0x1.e27076e2af2e6p-4, 0x1.61578001e0162p-60},
    {0x1.1f00000000000p+0, -0x1.d4313d66cb35dp-4, -0x1.790dd951d90fap-58},
    {0x1.1e00000000000p+0, -0x1.c5e548f5bc743p-4, -0x1.5d617ef8161b1p-60},
    {0x1.1e00000000000p+0, -0x1.c5e548f5bc743p-4, -0x1.5d617ef8161b1p-60},
    {0x1.1d00000000000p+0, -0x1.b78c82bb0eda1p-4, -0x1.0878cf0327e21p-61},
    {0x1.1d00000000000p+0, -0x1.b78c82bb0eda1p-4, -0x1.0878cf0327e21p-61},
    {0x1.1c00000000000p+0, -0x1.a926d3a4ad563p-4, -0x1.942f48aa70ea9p-58},
    {0x1.1b00000000000p+0, -0x1.9ab42462033adp-4, 0x1.2099e1c184e8ep-59},
    {0x1.1b00000000000p+0, -0x1.9ab42462033adp-4, 0x1.2099e1c184e8ep-59},
    {0x1.1a00000000000p+0, -0x1.8c345d6319b21p-4, 0x1.4a697ab3424a9p-61},
    {0x1.1a00000000000p+0, -0x1.8c345d6319b21p-4, 0x1.4a697ab3424a9p-61},
    {0x1.1900000000000p+0, -0x1.7da766d7b12cdp-4, 0x1.eeedfcdd94131p-58},
    {0x1.1800000000000p+0, -0x1.6f0d28ae56b4cp-4, 0x1.906d99184b992p-58},
    {0x1.1800000000000p+0, -0x1.6f0d28ae56b4cp-4, 0x1.906d99184b992p-58},
    {0x1.1700000000000p+0, -0x1.60658a93750c4p-4, 0x1.388458ec21b6ap-58},
    {0x1.1700000000000p+0, -0x1.60658a93750c4p-4, 0x1.388458ec21b6ap-58},
    {0x1.1600000000000p+0, -0x1.51b073f06183fp-4, -0x1.a49e39a1a8be4p-58},
    {0x1.1500000000000p+0, -0x1.42edcbea646f0p-4, -0x1.ddd4f935996c9p-59},
    {0x1.1500000000000p+0, -0x1.42edcbea646f0p-4, -0x1.ddd4f935996c9p-59},
    {0x1.1400000000000p+0, -0x1.341d7961bd1d1p-4, 0x1.b599f227becbbp-58},
    {0x1.1400000000000p+0, -0x1.341d7961bd1d1p-4, 0x1.b599f227becbbp-58},
    {0x1.1300000000000p+0, -0x1.253f62f0a1417p-4, 0x1.c125963fc4cfdp-62},
    {0x1.1200000000000p+0, -0x1.16536eea37ae1p-4, 0x1.79da3e8c22cdap-60},
    {0x1.1200000000000p+0, -0x1.16536eea37ae1p-4, 0x1.79da3e8c22cdap-60},
    {0x1.1100000000000p+0, -0x1.075983598e471p-4, -0x1.80da5333c45b8p-59},
    {0x1.1100000000000p+0, -0x1.075983598e471p-4, -0x1.80da5333c45b8p-59},
    {0x1.1000000000000p+0, -0x1.f0a30c01162a6p-5, -0x1.85f325c5bbacdp-59},
    {0x1.1000000000000p+0, -0x1.f0a30c01162a6p-5, -0x1.85f325c5bbacdp-59},
    {0x1.0f00000000000p+0, -0x1.d276b8adb0b52p-5, -0x1.1e3c53257fd47p-61},
    {0x1.0f00000000000p+0, -0x1.d276b8adb0b52p-5, -0x1.1e3c53257fd47p-61},
    {0x1.0e00000000000p+0, -0x1.b42dd711971bfp-5, 0x1.eb9759c130499p-60},
    {0x1.0d00000000000p+0, -0x1.95c830ec8e3ebp-5, -0x1.f5a0e80520bf2p-59},
    {0x1.0d00000000000p+0, -0x1.95c830ec8e3ebp-5, -0x1.f5a0e80520bf2p-59},
    {0x1.0c00000000000p+0, -0x1.77458f632dcfcp-5, -0x1.18d3ca87b9296p-59},
    {0x1.0c00000000000p+0, -0x1.77458f632dcfcp-5, -0x1.18d3ca87b9296p-59},
    {0x1.0b00000000000p+0, -0x1.58a5bafc8e4d5p-5, 0x1.ce55c2b4e2b72p-59},
    {0x1.0b00000000000p+0, -0x1.58a5bafc8e4d5p-5, 0x1.ce55c2b4e2b72p-59},
    {0x1.0a00000000000p+0, -0x1.39e87b9febd60p-5, 0x1.5bfa937f551bbp-59},
    {0x1.0a00000000000p+0, -0x1.39e87b9febd60p-5, 0x1.5bfa937f551bbp-59},
    {0x1.0900000000000p+0, -0x1.1b0d98923d980p-5, 0x1.e9ae889bac481p-60},
    {0x1.0900000000000p+0, -0x1.1b0d98923d980p-5, 0x1.e9ae889bac481p-60},
    {0x1.0800000000000p+0, -0x1.f829b0e783300p-6, -0x1.33e3f04f1ef23p-60},
    {0x1.0700000000000p+0, -0x1.b9fc027af9198p-6, 0x1.0ae69229dc868p-64},
    {0x1.0700000000000p+0, -0x1.b9fc027af9198p-6, 0x1.0ae69229dc868p-64},
    {0x1.0600000000000p+0, -0x1.7b91b07d5b11bp-6, 0x1.5b602ace3a510p-60},
    {0x1.0600000000000p+0, -0x1.7b91b07d5b11bp-6, 0x1.5b602ace3a510p-60},
    {0x1.0500000000000p+0, -0x1.3cea44346a575p-6, 0x1.0cb5a902b3a1cp-62},
    {0x1.0500000000000p+0, -0x1.3cea44346a575p-6, 0x1.0cb5a902b3a1cp-62},
    {0x1.0400000000000p+0, -0x1.fc0a8b0fc03e4p-7, 0x1.83092c59642a1p-62},
    {0x1.0400000000000p+0, -0x1.fc0a8b0fc03e4p-7, 0x1.83092c59642a1p-62},
    {0x1.0300000000000p+0, -0x1.7dc475f810a77p-7, 0x1.16d7687d3df21p-62},
    {0x1.0300000000000p+0, -0x1.7dc475f810a77p-7, 0x1.16d7687d3df21p-62},
    {0x1.0200000000000p+0, -0x1.fe02a6b106789p-8, 0x1.e44b7e3711ebfp-67},
    {0x1.0200000000000p+0, -0x1.fe02a6b106789p-8, 0x1.e44b7e3711ebfp-67},
    {0x1.0100000000000p+0, -0x1.ff00aa2b10bc0p-9, -0x1.2821ad5a6d353p-63},
    {0x1.0100000000000p+0, -0x1.ff00aa2b10bc0p-9, -0x1.2821ad5a6d353p-63},
    {0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0},
    {0x1.ff00000000000p-1, 0x1.0040155d5889ep-9, -0x1.8f98e1113f403p-65},
    {0x1.fd00000000000p-1, 0x1.8121214586b54p-8, 0x1.c14b9f9377a1dp-65},
    {0x1.fb00000000000p-1, 0x1.41929f96832f0p-7, -0x1.c5517f64bc223p-61},
    {0x1.f900000000000p-1, 0x1.c317384c75f06p-7, 0x1.806208c04c220p-61},
    {0x1.f700000000000p-1, 0x1.228fb1fea2e28p-6, -0x1.cd7b66e01c26dp-61},
    {0x1.f500000000000p-1, 0x1.63d6178690bd6p-6, -0x1.8ed4d357c9c97p-64},
    {0x1.f300000000000p-1, 0x1.a55f548c5c43fp-6, 0x1.ec1a5f86d41f9p-62},
    {0x1.f100000000000p-1, 0x1.e72bf2813ce51p-6, 0x1.75b44595cab18p-60},
    {0x1.ef00000000000p-1, 0x1.149e3e4005a8dp-5, -0x1.53482d1f9d7d7p-61},
    {0x1.ee00000000000p-1, 0x1.252f32f8d183fp-5, -0x1.947f792615916p-59},
    {0x1.ec00000000000p-1, 0x1.466aed42de3eap-5, -0x1.cdd6f7f4a137ep-59},
    {0x1.ea00000000000p-1, 0x1.67c94f2d4bb58p-5, 0x1.0413e6505e603p-59},
    {0x1.e800000000000p-1, 0x1.894aa149fb343p-5, 0x1.a8be97660a23dp-60},
    {0x1.e600000000000p-1, 0x1.aaef2d0fb10fcp-5, 0x1.a353bb42e0addp-61},
    {0x1.e400000000000p-1, 0x1.ccb73cdddb2ccp-5, -0x1.e48fb0500efd4p-59},
    {0x1.e300000000000p-1, 0x1.dda8adc67ee4ep-5, 0x1.4e6c986f44c55p-59},
    {0x1.e100000000000p-1, 0x1.ffa6911ab9301p-5, -0x1.cd9f1f95c2eedp-59},
    {0x1.df00000000000p-1, 0x1.10e45b3cae831p-4, -0x1.a4a128d192686p-58},
    {0x1.dd00000000000p-1, 0x1.2207b5c78549ep-4, -0x1.cc0fbce104eaap-58},
    {0x1.dc00000000000p-1, 0x1.2aa04a44717a5p-4, -0x1.d15d38d2fa3f7p-58},
    {0x1.da00000000000p-1, 0x1.3bdf5a7d1ee64p-4, 0x1.7a976d3b5b45fp-59},
    {0x1.d800000000000p-1, 0x1.4d3115d207eacp-4, 0x1.769f42c7842ccp-58},
    {0x1.d700000000000p-1, 0x1.55e10050e0384p-4, -0x1.45f9d61c68c1bp-58},
    {0x1.d500000000000p-1, 0x1.674f089365a7ap-4, -0x1.9acd8b33f8fdcp-58},
    {0x1.d300000000000p-1, 0x1.78d02263d82d3p-4, 0x1.abca5b4fdb880p-58},
    {0x1.d200000000000p-1, 0x1.8197e2f40e3f0p-4, 0x1.b9f2dffbeed43p-60},
    {0x1.d000000000000p-1, 0x1.9335e5d594989p-4, -0x1.478a85704ccb7p-58},
    {0x1.ce00000000000p-1, 0x1.a4e7640b1bc38p-4, -0x1.5b5ca203e4259p-58},
    {0x1.cd00000000000p-1, 0x1.adc77ee5aea8cp-4, 0x1.37d8f39bee659p-58},
    {0x1.cb00000000000p-1, 0x1.bf968769fca11p-4, -0x1.cdc9f6f5f38c7p-59},
    {0x1.c900000000000p-1, 0x1.d179788219364p-4, 0x1.9daf7df76ad2ap-59},
    {0x1.c800000000000p-1, 0x1.da727638446a2p-4, 0x1.401fa71733019p-58},
    {0x1.c600000000000p-1, 0x1.ec739830a1120p-4, -0x1.a2bf991780d3fp-59},
    {0x1.c500000000000p-1, 0x1.f57bc7d9005dbp-4, -0x1.9361574fb24e2p-58},
    {0x1.c300000000000p-1, 0x1.03cdc0a51ec0dp-3, 0x1.39e2d3f8b7d10p-57},
    {0x1.c200000000000p-1, 0x1.08598b59e3a07p-3, -0x1.dd7009902bf32p-57},
    {0x1.c000000000000p-1, 0x1.1178e8227e47cp-3, -0x1.0e63a5f01c691p-58},
    {0x1.bf00000000000p-1, 0x1.160c8024b27b1p-3, -0x1.2d56ff61c2bfbp-57},
    {0x1.bd00000000000p-1, 0x1.1f3b925f25d41p-3, 0x1.62c9ef939ac5dp-59},
    {0x1.bc00000000000p-1, 0x1.23d712a49c202p-3, -0x1.6e38161051d69p-57},
    {0x1.ba00000000000p-1, 0x1.2d1610c86813ap-3, -0x1.499a3f25af95fp-58},
    {0x1.b900000000000p-1, 0x1.31b994d3a4f85p-3, -0x1.c4716bdfc0cc9p-58},
    {0x1.b700000000000p-1, 0x1.3b08b6757f2a9p-3, 0x1.70d6cdf05266cp-60},
    {0x1.b600000000000p-1, 0x1.3fb45a59928ccp-3, -0x1.d87e6a354d056p-57},
    {0x1.b400000000000p-1, 0x1.4913d8333b561p-3, -0x1.0d5604930f135p-58},
    {0x1.b300000000000p-1, 0x1.4dc7b897bc1c8p-3, -0x1.927d47803c5f4p-57},
    {0x1.b100000000000p-1, 0x1.5737cc9018cddp-3, 0x1.4f4d710fec38ep-57},
    {0x1.b000000000000p-1, 0x1.5bf406b543db2p-3, -0x1.1f5b44c0df7e7p-61},
    {0x1.ae00000000000p-1, 0x1.6574ebe8c133ap-3, -0x1.d34f0f4621bedp-60},
    {0x1.ad00000000000p-1, 0x1.6a399dabbd383p-3, 0x1.96332bd4b341fp-57},
    {0x1.ac00000000000p-1, 0x1.6f0128b756abcp-3, -0x1.8de59c21e166cp-57},
    {0x1.aa00000000000p-1, 0x1.7898d85444c73p-3, 0x1.ef8f6ebcfb201p-58},
    {0x1.a900000000000p-1, 0x1.7d6903caf5ad0p-3, -0x1.ac5f0c075b847p-59},
    {0x1.a700000000000p-1, 0x1.871213750e994p-3, 0x1.d685f35eea2a0p-57},
    {0x1.a600000000000p-1, 0x1.8beafeb38fe8cp-3, 0x1.55aa8b6997a40p-58},
    {0x1.a500000000000p-1, 0x1.90c6db9fcbcd9p-3, 0x1.054473941ad99p-57},
    {0x1.a300000000000p-1, 0x1.9a8778debaa38p-3, 0x1.f47dfd871f87fp-57},
    {0x1.a200000000000p-1, 0x1.9f6c407089664p-3, 0x1.35a19605e67efp-59},
    {0x1.a100000000000p-1, 0x1.a454082e6ab05p-3, 0x1.df207dc5c34c6p-58},
    {0x1.9f00000000000p-1, 0x1.ae2ca6f672bd4p-3, 0x1.ab5ca9eaa088ap-57},
    {0x1.9e00000000000p-1, 0x1.b31d8575bce3dp-3, -0x1.6353ab386a94dp-57},
    {0x1.9d00000000000p-1, 0x1.b811730b823d2p-3, 0x1.a0ee735d9f0ecp-60},
    {0x1.9b00000000000p-1, 0x1.c2028ab17f9b4p-3, 0x1.f11aa3853a5f1p-57},
    {0x1.9a00000000000p-1, 0x1.c6ffbc6f00f71p-3, -0x1.8e58b2c57a4a5p-57},
    {0x1.9900000000000p-1, 0x1.cc000c9db3c52p-3, 0x1.53d154280394fp-57},
    {0x1.9800000000000p-1, 0x1.d1037f2655e7bp-3, 0x1.60629242471a2p-57},
    {0x1.9600000000000p-1, 0x1.db13db0d48940p-3, 0x1.aa11d49f96cb9p-58},
    {0x1.9500000000000p-1, 0x1.e020cc6235ab5p-3, 0x1.fea48dd7b81d1p-58},
    {0x1.9400000000000p-1, 0x1.e530effe71012p-3, 0x1.2276041f43042p-59},
    {0x1.9300000000000p-1, 0x1.ea4449f04aaf5p-3, -0x1.d33919ab94074p-57},
    {0x1.9100000000000p-1, 0x1.f474b134df229p-3, -0x1.27c77ded76aadp-58},
    {0x1.9000000000000p-1, 0x1.f991c6cb3b379p-3, 0x1.f665066f980a2p-57},
    {0x1.8f00000000000p-1, 0x1.feb2233ea07cdp-3, 0x1.8de00938b4c40p-61},
    {0x1.8e00000000000p-1, 0x1.01eae5626c691p-2, -0x1.18290bd2932e2p-59},
    {0x1.8d00000000000p-1, 0x1.047e60cde83b8p-2, -0x1.0779634061cbcp-56},
    {0x1.8b00000000000p-1, 0x1.09aa572e6c6d4p-2, 0x1.43c2e68684d53p-57},
    {0x1.8a00000000000p-1, 0x1.0c42d676162e3p-2, 0x1.162c79d5d11eep-58},
    {0x1.8900000000000p-1, 0x1.0edd060b78081p-2, -0x1.92b49ef282b09p-57},
    {0x1.8800000000000p-1, 0x1.1178e8227e47cp-2, -0x1.0e63a5f01c691p-57},
    {0x1.8700000000000p-1, 0x1.14167ef367783p-2, 0x1.e0936abd4fa6ep-62},
    {0x1.8500000000000p-1, 0x1.1956d3b9bc2fap-2, 0x1.7b9d68d50a15dp-56},
    {0x1.8400000000000p-1, 0x1.1bf99635a6b95p-2, -0x1.12aeb84249223p-57},
    {0x1.8300000000000p-1, 0x1.1e9e1678899f4p-2, 0x1.512c3749a1e4ep-56},
    {0x1.8200000000000p-1, 0x1.214456d0eb8d4p-2, 0x1.f7ae91aeba60ap-57},
    {0x1.8100000000000p-1, 0x1.23ec5991eba49p-2, 0x1.bb75d1addf870p-60},
    {0x1.8000000000000p-1, 0x1.269621134db92p-2, 0x1.e0efadd9db02bp-56},
    {0x1.7f00000000000p-1, 0x1.2941afb186b7cp-2, -0x1.856e61c515740p-57},
    {0x1.7e00000000000p-1, 0x1.2bef07cdc9354p-2, -0x1.82dad7fd86088p-56},
    {0x1.7c00000000000p-1, 0x1.314f1e1d35ce4p-2, -0x1.3d69909e5c3dcp-56},
    {0x1.7b00000000000p-1, 0x1.3401e12aecba1p-2, -0x1.cd55b8a4746c0p-58},
    {0x1.7a00000000000p-1, 0x1.36b6776be1117p-2, -0x1.324f0e883858ep-58},
    {0x1.7900000000000p-1, 0x1.396ce359bbf54p-2, -0x1.ce2b31b31e8b0p-58},
    {0x1.7800000000000p-1, 0x1.3c25277333184p-2, -0x1.2ad27e50a8ec6p-56},
    {0x1.7700000000000p-1, 0x1.3edf463c1683ep-2, 0x1.83d680d3c1084p-56},
    {0x1.7600000000000p-1, 0x1.419b423d5e8c7p-2, 0x1.0dbb243827392p-57},
    {0x1.7500000000000p-1, 0x1.44591e0539f49p-2, -0x1.2b125247b0fa5p-56},
    {0x1.7400000000000p-1, 0x1.4718dc271c41bp-2, 0x1.8fb4c14c56eefp-60},
    {0x1.7300000000000p-1, 0x1.49da7f3bcc41fp-2, -0x1.9964a168ccacap-57},
    {0x1.7200000000000p-1, 0x1.4c9e09e172c3cp-2, -0x1.123615b147a5dp-58},
    {0x1.7100000000000p-1, 0x1.4f637ebba9810p-2, -0x1.58cb3124b9245p-56},
    {0x1.7000000000000p-1, 0x1.522ae0738a3d8p-2, -0x1.8f7e9b38a6979p-57},
    {0x1.6f00000000000p-1, 0x1.54f431b7be1a9p-2, -0x1.aacfdbbdab914p-56},
    {0x1.6e00000000000p-1, 0x1.57bf753c8d1fbp-2, -0x1.0908d15f88b63p-57},
    {0x1.6d00000000000p-1, 0x1.5a8cadbbedfa1p-2, -0x1.e6c2bdfb3e037p-58},
    {0x1.6c00000000000p-1, 0x1.5d5bddf595f30p-2, -0x1.6541148cbb8a2p-56},
    {0x1.6b00000000000p-1, 0x1.602d08af091ecp-2, -0x1.6e8920c09b73fp-58},
    // LOG64_TABLE_END
};

// s + e = a + b exactly (Knuth)
__device__ __forceinline__ void two_sum_(double a, double b, double &s, double &e)
{
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

// log to ~2^-64 for positive NORMAL finite x (oracle/halo_oracle_math.h: ho_log_cr, the same operations)
__device__ __forceinline__ double det_log_cr_core(double x)
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    const uint32_t off = hx & 0x000fffffu;
    const int j = (int)(off >> 12);
    hx = off + 0x3fe6a09eu;
    const double m = __longlong_as_double((long long)(((uint64_t)hx << 32) | (u & 0xffffffffull)));
    const double r = log64_tab_[j][0], Lh = log64_tab_[j][1], Ll = log64_tab_[j][2];
    const double z = __builtin_fma(m, r, -1.0);                     // exact, |z| < 2^-8
    const double zh = z * z, zl = __builtin_fma(z, z, -zh);         // z^2 = zh + zl exactly
    double P = -0.1;
    P = __builtin_fma(P, z, 0x1.c71c71c71c71cp-4);
    P = __builtin_fma(P, z, -0.125);
    P = __builtin_fma(P, z, 0x1.2492492492492p-3);
    P = __builtin_fma(P, z, -0x1.5555555555555p-3);
    P = __builtin_fma(P, z, 0.2);
    P = __builtin_fma(P, z, -0.25);
    P = __builtin_fma(P, z, 0x1.5555555555555p-2);
    const double tail = (zh * z) * P;
    const double dk = (double)k;
    const double t1 = dk * 0x1.62e42fefa3800p-1;                    // ln2_hi: 42 bits, exact product
    double hi, e1, e2, e3;
    two_sum_(t1, Lh, hi, e1);
    two_sum_(hi, z, hi, e2);
    two_sum_(hi, -0.5 * zh, hi, e3);
    const double lo = ((e1 + e2) + e3) + ((__builtin_fma(dk, 0x1.ef35793c76730p-45, Ll) - 0.5 * zl) + tail);
    return hi + lo;
}

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
    const double r = (det_log_cr_core(1.0 + z) - det_log_cr_core(1.0 - z)) * 0.5;    // both arguments in [1e-7, 2)
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
