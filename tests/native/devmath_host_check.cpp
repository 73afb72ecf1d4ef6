// Host-side equivalence check of the device math recipes (halo_amd/csrc/halo_devmath.hpp, straight-line code with the
// special cases patched in by selects) against the oracle's branchy statement (oracle/halo_oracle_math.h).
// Both are sequences of IEEE-754 operations, so the host evaluates them exactly as gfx950 does.
//   g++ -O2 -ffp-contract=off -fno-fast-math devmath_host_check.cpp -o check && ./check [stride]
// stride 1 = every float32 bit pattern (minutes); the test suite runs a coarser stride plus every pattern within
// the ranges where the device code relies on natural overflow / underflow instead of an explicit cut-off.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" {
#include "../../oracle/halo_oracle_math.h"
}

#define __device__
#define __forceinline__ inline
#define HALO_DEVMATH_HOST_CHECK 1
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double __longlong_as_double(long long u) { double f; memcpy(&f, &u, 8); return f; }
static inline long long __double_as_longlong(double f) { long long u; memcpy(&u, &f, 8); return u; }
#include "../../halo_amd/csrc/halo_devmath.hpp"

static long bad = 0;
static void cmpf(const char *what, uint32_t u, float a, float b)
{
    if (__float_as_uint(a) != __float_as_uint(b)) {
        if (bad < 20) printf("%s(%08x): device recipe %08x, oracle %08x\n", what, u, __float_as_uint(a), __float_as_uint(b));
        ++bad;
    }
}
static void cmpd(const char *what, uint64_t u, double a, double b)
{
    if (__double_as_longlong(a) != __double_as_longlong(b)) {
        if (bad < 20) printf("%s(%016llx): device recipe %016llx, oracle %016llx\n", what, (unsigned long long)u,
                             (unsigned long long)__double_as_longlong(a), (unsigned long long)__double_as_longlong(b));
        ++bad;
    }
}
static void one32(uint32_t u)
{
    const float x = __uint_as_float(u);
    cmpf("expf", u, halo::det_expf(x), ho_expf(x));
    cmpf("logf", u, halo::det_logf(x), ho_logf(x));
    // the *_core forms on the domains their callers guarantee
    if (x >= -104.0f && x <= 100.0f) cmpf("expf_core", u, halo::det_expf_core(x), ho_expf(x));
    if (x >= -87.0f && x <= 0.35f) cmpf("expf_core_small", u, halo::det_expf_core_small(x), ho_expf(x));
    if (u >= 0x00800000u && u < 0x7f800000u) cmpf("logf_core", u, halo::det_logf_core(x), ho_logf(x));
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 61;
    long n = 0;
    for (uint64_t i = 0; i < (1ull << 32); i += stride, ++n) one32((uint32_t)i);
    // every pattern from just inside the cut-offs outwards to +-1024, and every NaN / infinity neighbourhood
    for (uint32_t u = __float_as_uint(88.0f); u <= __float_as_uint(1024.0f); ++u, ++n) one32(u);
    // the lean softmax's whole argument range [-64, 0], every pattern (64 down to 2^-20, and the zeros), and the ends of
    // det_expf_core_small's domain
    for (uint32_t u = __float_as_uint(-9.5e-7f); u <= __float_as_uint(-64.0f); ++u, ++n) one32(u);
    for (uint32_t u = __float_as_uint(-64.0f); u <= __float_as_uint(-87.0f); u += 3, ++n) one32(u);
    for (uint32_t u = 0; u <= __float_as_uint(0.35f); u += 97, ++n) one32(u);
    for (uint32_t u = __float_as_uint(-103.0f); u <= __float_as_uint(-1024.0f); ++u, ++n) one32(u);
    for (uint32_t u = 0x7f7ffff0u; u < 0x7f800100u; ++u, ++n) { one32(u); one32(u | 0x80000000u); }
    for (uint32_t u = 0; u < 0x00800100u; u += 1 + (u >> 12), ++n) { one32(u); one32(u | 0x80000000u); }
    // float64 log: xorshift patterns (all classes), mantissa sweeps around 1 and sqrt(1/2), subnormals, specials
    uint64_t s = 88172645463325252ull;
    const long nd = (long)(400000000ull / (stride < 1 ? 1 : stride)) + 1000000;
    for (long i = 0; i < nd; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t u = s;
        if (i % 5 == 0) u &= 0x000fffffffffffffull;
        if (i % 7 == 0) u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
        if (i % 11 == 0) u = (u & 0x000fffffffffffffull) | 0x3fe0000000000000ull;
        const double x = __longlong_as_double((long long)u);
        cmpd("log", u, halo::det_log(x), ho_log(x));
        if (u >= 0x0010000000000000ull && u < 0x7ff0000000000000ull) {
            cmpd("log_core", u, halo::det_log_core(x), ho_log(x));
            cmpd("log_cr_core", u, halo::det_log_cr_core(x), ho_log_cr(x));
        }
    }
    const double sp[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 1.0, 5e-324, 2.2250738585072014e-308, -1.0, 1.7976931348623157e308};
    for (double x : sp) cmpd("log", (uint64_t)__double_as_longlong(x), halo::det_log(x), ho_log(x));
    printf("checked %ld float32 patterns, %ld float64 patterns: %ld mismatches\n", n, nd, bad);
    return bad != 0;
}
