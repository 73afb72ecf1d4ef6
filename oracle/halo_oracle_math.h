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
 * expf / logf follow the classic Cephes single-precision recipes (range reduction
 * + minimax polynomial, about 1 ulp); log follows the classic fdlibm recipe
 * (k*ln2 + log1p-style series in s = f/(2+f), < 1 ulp).  They stand in for the
 * reference's ATen/Sleef calls (torch.softmax, torch.log: floating_region.py:72,
 * 119,152; geoopt artanh's torch.log) and agree with them to a few ulp, which the
 * golden fixtures check (scores <= 1e-4, selected indices exact).
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
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541015625f) return 0.0f;
    float k = rintf(x * 1.44269502162933349609375f);          /* log2(e) */
    float r = fmaf(k, -0.693359375f, x);                       /* ln2 hi  */
    r = fmaf(k, 2.12194440e-4f, r);                            /* -ln2 lo */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int ki = (int)k;
    int k1 = ki >> 1;               /* floor(k/2): both factors stay normal */
    int k2 = ki - k1;
    return (y * ho_pow2f(k1)) * ho_pow2f(k2);
}

/* natural log, x > 0 finite and normal is the contract; the rest is for safety */
static inline float ho_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = ho_f2u(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = ho_f2u(x); e = -23; }
    e += (int)(u >> 23) - 126;                                 /* x = m * 2^e, m in [0.5,1) */
    float m = ho_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106769084930419921875f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = m + y;
    return fmaf(fe, 0.693359375f, r);
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
