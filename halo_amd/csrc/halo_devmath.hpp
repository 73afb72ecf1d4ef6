// Elementary functions of the HALO-AMD numeric contract for gfx950 device code.
//
// The scoring maps feed an argmax-driven greedy selector whose output must be bit-identical
// between runs, ranks and the CPU checker, so exp/log are not taken from the device libm
// (whose results differ from any host libm in the last ulp).  Instead each is a fixed
// sequence of IEEE-754 operations -- Cephes-style single-precision expf/logf, fdlibm-style
// double log -- with every fused multiply-add written out; the translation unit is built
// with -ffp-contract=off so nothing else is contracted.  sqrt and '/' are the correctly
// rounded device instructions sequences hipcc emits by default.
//
// Stands in for: torch.softmax / torch.log in FloatingRegionScore
// (core/active/floating_region.py:72,119,152) and the float64 torch.log inside geoopt's
// artanh (used by dist0, core/utils/hyperbolic.py:83).
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

// expf for finite x in [-104, 89]
__device__ __forceinline__ float det_expf_core(float x)
{
    float k = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = __builtin_fmaf(k, -0.693359375f, x);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r) + 1.0f;
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
    float r = __builtin_fmaf(k, -0.693359375f, x);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    return __uint_as_float(__float_as_uint(y) + (__float_as_uint(tm) << 23));      // == ldexpf(y, (int)k)
}

__device__ __forceinline__ float det_expf(float x)
{
    // The reduction runs on a clamped copy (in-range x is unchanged; NaN becomes a bound), so every intermediate is finite.
    // Beyond the oracle's cut-offs (x > 88.7228... -> +inf, x < -103.972... -> 0) the clamped value lands there by itself:
    // y * 2^64 * 2^64 overflows for every xc in (88.7228, 89], y * 2^-75 * 2^-75 with y < 1 rounds to zero for every xc in
    // [-104, -103.972) -- the host check walks every such float32 input.
    const float res = det_expf_core(__builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.0f));
    return x != x ? x : res;
}

// logf of the positive normal float whose bits are u, plus e0 * ln 2
__device__ __forceinline__ float logf_core_(uint32_t u, int e0)
{
    int e = e0 + ((int)(u >> 23) - 126);
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    const bool low = m < 0.707106769084930419921875f;
    e -= low ? 1 : 0;
    m = (low ? m + m : m) - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

// logf for positive normal finite x: logf_core_ with its normalisation -- mantissa m0 in [0.5, 1) and exponent, then "m0 <
// sqrt(1/2): double it and lower the exponent" (a compare, two selects, an integer subtract) -- written on the bits: with
// v = bits(x) - bits(sqrt(1/2)) (0x3f3504f3, the constant of the comparison), k = v >> 23 (arithmetic) is the adjusted exponent
// and bits(x) - (v & 0xff800000) the adjusted mantissa in [sqrt(1/2), sqrt(2)) -- the mantissa field is below the constant's
// exactly when m0 < sqrt(1/2), and then the subtraction borrows one from the exponent.  Two 2-cycle integer subtractions and two
// 4-cycle bit operations where the frexp pair, the compare and the selects were seven 4-cycle instructions; the same m, the same
// exponent, the same operations afterwards.
__device__ __forceinline__ float det_logf_core(float x)
{
    const uint32_t u = __float_as_uint(x);
    const uint32_t v = u - 0x3f3504f3u;
    const int e = (int)v >> 23;
    float m = __uint_as_float(u - (v & 0xff800000u)) - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

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
