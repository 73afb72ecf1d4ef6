/* TEST INFRASTRUCTURE -- not part of the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call this file.
 *
 * CPU restatement (plain C, optional OpenMP) of the reference's per-pixel
 * hyperbolic acquisition-scoring path.  Reference = paolomandica/HALO, pure
 * Python; every function cites the reference lines it follows.  The geoopt
 * formulas (expmap0/project/logmap0/dist0/dist) are restated from geoopt's
 * published stereographic/math.py -- geoopt is an un-pinned, un-vendored
 * dependency of the reference (requirements.txt:13), so that layer is
 * "parity unpinned"; everything in-tree is pinned by the .npz files under tests/golden, which
 * were produced by running the reference's own code (tests/golden/make_fixtures.py).
 *
 * Layout convention: tensors are dense row-major exactly as the reference holds
 * them: logit (O,H,W) f32, decoder_out (C,H,W) f64 or f32, maps (H,W).
 *
 * Numeric contract (shared with the HIP kernels): per-pixel sums run
 * sequentially over the channel / class index with explicit fma; exp/log come
 * from halo_oracle_math.h; sqrt and division are IEEE.  Compile with
 * -ffp-contract=off.
 */
#include "halo_oracle_math.h"
#include <stdlib.h>
#include <string.h>

typedef long long i64;
typedef unsigned char u8;

enum { HALO_UNC_ENTROPY = 0, HALO_UNC_PIXEL_ENTROPY = 1, HALO_UNC_ORACLE_ACC = 2, HALO_UNC_ZEROS = 3 };
enum { HALO_PUR_RIPU = 0, HALO_PUR_ORACLE_RIPU = 1, HALO_PUR_HYPER = 2, HALO_PUR_NONE = 3,
       HALO_PUR_RADIUS = 4, HALO_PUR_EUC_NORM = 5 };
enum { HALO_F32 = 0, HALO_F64 = 1 };

/* ---- exported math probes (so tests can pin the elementary functions) ---- */
float halo_o_expf(float x) { return ho_expf(x); }
float halo_o_logf(float x) { return ho_logf(x); }
double halo_o_log(double x) { return ho_log(x); }
double halo_o_log_cr(double x) { return ho_log_cr(x); }

void halo_o_expf_v(const float *x, float *y, i64 n)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) y[i] = ho_expf(x[i]);
}
void halo_o_logf_v(const float *x, float *y, i64 n)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) y[i] = ho_logf(x[i]);
}

/* sabs(k)**0.5 with k = -c  (geoopt sabs: |x| + 1e-15) */
static double k_sqrt(double c) { return sqrt(fabs(-c) + 1e-15); }

/* geoopt artanh: clamp to +-(1-1e-7), 0.5*(log(1+z) - log(1-z)) in float64 */
static double artanh_clamped(double z)
{
    const double lim = 1.0 - 1e-7;
    if (z > lim) z = lim;
    if (z < -lim) z = -lim;
    return (ho_log_cr(1.0 + z) - ho_log_cr(1.0 - z)) * 0.5;
}

/* sum of squares over a strided vector, sequential fma chain in float64 */
static double ssq_f64(const double *x, i64 n, i64 stride)
{
    double a = 0.0;
    for (i64 i = 0; i < n; ++i) { double v = x[i * stride]; a = fma(v, v, a); }
    return a;
}
/* x.double() first (hyperbolic.py:37), then the float64 chain */
static double ssq_f32_as_f64(const float *x, i64 n, i64 stride)
{
    double a = 0.0;
    for (i64 i = 0; i < n; ++i) { double v = (double)x[i * stride]; a = fma(v, v, a); }
    return a;
}
/* float32 tensor reduced in float32 (x.norm on a float32 decoder_out): fmaf chain.
 * Both chains reproduce ATen's CPU norm over dim=1 bit for bit on FMA hosts
 * (checked against torch 2.10 in the build container). */
static float ssq_f32(const float *x, i64 n, i64 stride)
{
    float a = 0.0f;
    for (i64 i = 0; i < n; ++i) { float v = x[i * stride]; a = fmaf(v, v, a); }
    return a;
}

/* ------------------------------------------------------------------------- *
 * HyperMapper.expmap (core/utils/hyperbolic.py:28-39):
 *   expmap0(x.double(), k=-c, dim) then project(.., k=-c, dim), eps=1e-5 (f64).
 * x viewed as (outer, C, inner), reduction over C.  in_dtype: HALO_F32/HALO_F64.
 * ------------------------------------------------------------------------- */
void halo_o_expmap0_project(const void *x, int in_dtype, double *y, i64 outer, i64 C, i64 inner, double c)
{
    const double ks = k_sqrt(c), rks = 1.0 / ks;
    const double maxnorm = (1.0 - 1e-5) / sqrt(fabs(-c) + 1e-15);
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 o = 0; o < outer; ++o)
        for (i64 i = 0; i < inner; ++i) {
            const i64 base = o * C * inner + i;
            double n = in_dtype == HALO_F64 ? sqrt(ssq_f64((const double *)x + base, C, inner))
                                            : sqrt(ssq_f32_as_f64((const float *)x + base, C, inner));
            if (n < 1e-15) n = 1e-15;
            double a = n * ks;
            if (a > 15.0) a = 15.0;
            if (a < -15.0) a = -15.0;
            const double g = rks * tanh(a);
            double s2 = 0.0;
            for (i64 ch = 0; ch < C; ++ch) {
                double u = in_dtype == HALO_F64 ? ((const double *)x)[base + ch * inner]
                                                : (double)((const float *)x)[base + ch * inner];
                double v = g * (u / n);
                y[base + ch * inner] = v;
                s2 = fma(v, v, s2);
            }
            double ny = sqrt(s2);
            if (ny < 1e-15) ny = 1e-15;
            if (ny > maxnorm)
                for (i64 ch = 0; ch < C; ++ch) y[base + ch * inner] = y[base + ch * inner] / ny * maxnorm;
        }
}

/* HyperMapper.logmap (hyperbolic.py:51-60): project(logmap0(x.double())), last dim => inner=1 */
void halo_o_logmap0_project(const double *x, double *y, i64 outer, i64 C, i64 inner, double c)
{
    const double ks = k_sqrt(c), rks = 1.0 / ks;
    const double maxnorm = (1.0 - 1e-5) / sqrt(fabs(-c) + 1e-15);
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 o = 0; o < outer; ++o)
        for (i64 i = 0; i < inner; ++i) {
            const i64 base = o * C * inner + i;
            double n = sqrt(ssq_f64(x + base, C, inner));
            if (n < 1e-15) n = 1e-15;
            const double g = rks * artanh_clamped(n * ks);
            double s2 = 0.0;
            for (i64 ch = 0; ch < C; ++ch) {
                double v = (x[base + ch * inner] / n) * g;
                y[base + ch * inner] = v;
                s2 = fma(v, v, s2);
            }
            double ny = sqrt(s2);
            if (ny < 1e-15) ny = 1e-15;
            if (ny > maxnorm)
                for (i64 ch = 0; ch < C; ++ch) y[base + ch * inner] = y[base + ch * inner] / ny * maxnorm;
        }
}

/* radius of one pixel: geoopt dist0 = 2 * artan_k(||x||) (hyperbolic.py:74-83) */
static double dist0_from_ssq_f64(double ssq, double ks, double rks)
{
    return 2.0 * (rks * artanh_clamped(sqrt(ssq) * ks));
}
/* float32 input: everything in float32 -- geoopt's stereographic artanh is x.clamp(-1+1e-7, 1-1e-7) followed by
 * 0.5 * (log(1 + x) - log(1 - x)) IN THE INPUT DTYPE (the float64 detour belongs to the older poincare/math.py
 * Artanh).  The clamp bound 1 - 1e-7 rounds to 1 - 2^-23 in float32. */
static float dist0_from_ssq_f32(float ssq, double ks, double rks)
{
    float n = sqrtf(ssq);
    float z = n * (float)ks;
    const float lim = (float)(1.0 - 1e-7);
    if (z > lim) z = lim;
    if (z < -lim) z = -lim;
    float a = (ho_logf(1.0f + z) - ho_logf(1.0f - z)) * 0.5f;
    if (z != z) a = z;
    return 2.0f * ((float)rks * a);
}

/* HyperMapper.poincare_distance_origin (hyperbolic.py:74-83); out dtype = in dtype */
void halo_o_dist0(const void *x, int dtype, void *out, i64 outer, i64 C, i64 inner, double c)
{
    const double ks = k_sqrt(c), rks = 1.0 / ks;
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 o = 0; o < outer; ++o)
        for (i64 i = 0; i < inner; ++i) {
            const i64 base = o * C * inner + i;
            if (dtype == HALO_F64)
                ((double *)out)[o * inner + i] = dist0_from_ssq_f64(ssq_f64((const double *)x + base, C, inner), ks, rks);
            else
                ((float *)out)[o * inner + i] = dist0_from_ssq_f32(ssq_f32((const float *)x + base, C, inner), ks, rks);
        }
}

/* HyperMapper.poincare_distance (hyperbolic.py:62-72): geoopt dist over the last dim, f64 */
void halo_o_dist(const double *x, const double *y, double *out, i64 n, i64 d, double c)
{
    const double k = -c, ks = k_sqrt(c), rks = 1.0 / ks;
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r < n; ++r) {
        const double *a = x + r * d, *b = y + r * d;
        double x2 = 0, y2 = 0, xy = 0;
        for (i64 j = 0; j < d; ++j) {
            double u = -a[j], v = b[j];
            x2 = fma(u, u, x2); y2 = fma(v, v, y2); xy = fma(u, v, xy);
        }
        const double ca = 1.0 - 2.0 * k * xy - k * y2, cb = 1.0 + k * x2;
        double den = 1.0 - 2.0 * k * xy + k * k * x2 * y2;
        if (den < 1e-15) den = 1e-15;
        double s2 = 0;
        for (i64 j = 0; j < d; ++j) {
            double m = (ca * (-a[j]) + cb * b[j]) / den;
            s2 = fma(m, m, s2);
        }
        out[r] = 2.0 * (rks * artanh_clamped(sqrt(s2) * ks));
    }
}

/* ------------------------------------------------------------------------- *
 * HyperMLR._hyper_logits (hyperbolic.py:120-184).  x (B,C,hw) f64, P/A (O,C) f64,
 * out (B,O,hw) f64.
 * ------------------------------------------------------------------------- */
void halo_o_hypermlr(const double *x, const double *P, const double *A, double *out,
                     i64 B, i64 C, i64 O, i64 hw, double c)
{
    const double K = c, sqK = sqrt(K);
    const double maxnorm = (1.0 - 1e-3) / sqK;
    double *pp = (double *)malloc(sizeof(double) * O), *anorm = (double *)malloc(sizeof(double) * O);
    double *pa = (double *)malloc(sizeof(double) * O), *An = (double *)malloc(sizeof(double) * O * C);
    for (i64 o = 0; o < O; ++o) {
        double sp = 0, sa = 0;
        for (i64 j = 0; j < C; ++j) { sp = fma(P[o * C + j], P[o * C + j], sp); sa = fma(A[o * C + j], A[o * C + j], sa); }
        double np_ = sqrt(sp); pp[o] = np_ * np_;                 /* torch.norm(-P)**2, :137 */
        anorm[o] = sqrt(sa);                                       /* :172 */
        double dn = anorm[o] < 1e-12 ? 1e-12 : anorm[o];           /* F.normalize eps, :173 */
        double s = 0;
        for (i64 j = 0; j < C; ++j) { An[o * C + j] = A[o * C + j] / dn; s = fma(-P[o * C + j], An[o * C + j], s); }
        pa[o] = s;                                                 /* sum(-P * normed_A), :176 */
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 b = 0; b < B; ++b)
        for (i64 i = 0; i < hw; ++i) {
            const double *xb = x + b * C * hw + i;
            double nx = sqrt(ssq_f64(xb, C, hw));
            const double xx = nx * nx;                             /* torch.norm(x,dim=1)**2, :136 */
            for (i64 o = 0; o < O; ++o) {
                double px = 0, xa = 0;
                for (i64 j = 0; j < C; ++j) { px = fma(xb[j * hw], -P[o * C + j], px); xa = fma(xb[j * hw], An[o * C + j], xa); }
                const double sqsq = ((K * xx) * K) * pp[o];        /* :146 */
                const double Aa = (1.0 + (2.0 * K) * px) + K * xx; /* :150 */
                const double Bb = 1.0 - K * pp[o];                 /* :151 */
                double D = (1.0 + (2.0 * K) * px) + sqsq;          /* :152 */
                if (!(D >= 1e-12)) D = D != D ? D : 1e-12;         /* torch.max(D, 1e-12), :153 */
                const double al = Aa / D, be = Bb / D;
                const double mob = ((al * al) * pp[o] + (be * be) * xx) + ((2.0 * al) * be) * px; /* :159 */
                const double sq = sqrt(mob);
                double sqc = sq;
                if (!(sqc >= 1e-12)) sqc = sqc != sqc ? sqc : 1e-12;
                const double pn = sq > maxnorm ? maxnorm / sqc : 1.0;          /* :163-166 */
                const double mp = sq < maxnorm ? mob : maxnorm * maxnorm;       /* :167-170 */
                double md = (be * xa + al * pa[o]) * pn;                        /* :175-178 */
                double lden = 1.0 - K * mp;
                if (!(lden >= 1e-12)) lden = lden != lden ? lden : 1e-12;
                const double lamb = 2.0 / lden;                                 /* :179 */
                const double sine = (sqK * md) * lamb;                          /* :180 */
                out[(b * O + o) * hw + i] = ((2.0 / sqK) * anorm[o]) * asinh(sine); /* :182-183 */
            }
        }
    free(pp); free(anorm); free(pa); free(An);
}

/* ------------------------------------------------------------------------- *
 * F.interpolate(mode="bilinear", align_corners=True)  (build.py:123-125,133-135;
 * classifier.py:375-377,556-557).  planes x (h,w) -> planes x (H,W).
 * Source index = dst * (in-1)/(out-1) evaluated in the tensor's own dtype; the four taps are
 * combined the way ATen writes it (UpSampleKernel.cpp / UpSampleBilinear2d.cu:
 * h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)), columns first, rows second, every
 * "p*q + r*s" contracted as fma(p, q, r*s):
 *     top = fma(lx0, v00, lx1*v01)   bot = fma(lx0, v10, lx1*v11)   out = fma(ly0, top, ly1*bot)
 * That is bit for bit what torch's CPU kernel returns on an FMA host at the shapes the path runs
 * (160x320 -> 1024x2048 float64, 640x1280 -> 1024x2048 float32, the fixture shapes; 1 or 8 threads:
 * tests/test_oracle_golden.py::test_bilinear_is_torchs_cpu_kernel_bit_for_bit).  ATen has other code
 * paths (tiny outputs with power-of-two scales were seen 1 ulp away), so 1 ulp stays the bar in general.
 * (Rounds 1-3 used product weights w_ij = ly_i*lx_j and one 4-term fma chain: 1 ulp from ATen nearly
 * everywhere.)
 * ------------------------------------------------------------------------- */
void halo_o_bilinear_f64(const double *src, double *dst, i64 planes, i64 h, i64 w, i64 H, i64 W)
{
    const double sh = H > 1 ? (double)(h - 1) / (double)(H - 1) : 0.0;
    const double sw = W > 1 ? (double)(w - 1) / (double)(W - 1) : 0.0;
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 p = 0; p < planes; ++p)
        for (i64 y = 0; y < H; ++y) {
            const double fy = sh * (double)y;
            i64 y0 = (i64)fy; if (y0 > h - 1) y0 = h - 1;
            const i64 y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const double ly1 = fy - (double)y0, ly0 = 1.0 - ly1;
            const double *r0 = src + (p * h + y0) * w, *r1 = src + (p * h + y1) * w;
            for (i64 x = 0; x < W; ++x) {
                const double fx = sw * (double)x;
                i64 x0 = (i64)fx; if (x0 > w - 1) x0 = w - 1;
                const i64 x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const double lx1 = fx - (double)x0, lx0 = 1.0 - lx1;
                const double top = fma(lx0, r0[x0], lx1 * r0[x1]), bot = fma(lx0, r1[x0], lx1 * r1[x1]);
                dst[(p * H + y) * W + x] = fma(ly0, top, ly1 * bot);
            }
        }
}
void halo_o_bilinear_f32(const float *src, float *dst, i64 planes, i64 h, i64 w, i64 H, i64 W)
{
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 p = 0; p < planes; ++p)
        for (i64 y = 0; y < H; ++y) {
            const float fy = sh * (float)y;
            i64 y0 = (i64)fy; if (y0 > h - 1) y0 = h - 1;
            const i64 y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            const float *r0 = src + (p * h + y0) * w, *r1 = src + (p * h + y1) * w;
            for (i64 x = 0; x < W; ++x) {
                const float fx = sw * (float)x;
                i64 x0 = (i64)fx; if (x0 > w - 1) x0 = w - 1;
                const i64 x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const float top = fmaf(lx0, r0[x0], lx1 * r0[x1]), bot = fmaf(lx0, r1[x0], lx1 * r1[x1]);
                dst[(p * H + y) * W + x] = fmaf(ly0, top, ly1 * bot);
            }
        }
}

/* ------------------------------------------------------------------------- *
 * Gram form of the radius / norm of a bilinearly upsampled float64 embedding (the
 * reference computes it by upsampling, core/active/build.py:133-135, then reducing,
 * core/utils/hyperbolic.py:74-83).  NOT the reference's evaluation order: the upsampled vector of an
 * output pixel is sum_i w_i v_i over the four corner vectors of its low-res cell, hence
 *     ||.||^2 = sum_{i<=j} (2 - [i==j]) w_i w_j <v_i, v_j> .
 * This function restates, operation for operation, what the HIP pair k_gram_lr + k_radius_gram
 * (halo_amd/csrc/halo_score.hip) computes, so that the product's 'gram' mode has a bit-exact CPU
 * twin like the 'exact' mode has: five maps over the low-res grid
 *     S = <v,v>  Hh = <v(y,x), v(y,n(x))>  Vv = <v(y,x), v(n(y),x)>  D1 = <v(y,x), v(n(y),n(x))>
 *     D2 = <v(y,n(x)), v(n(y),x)>          (n(.) = neighbour clamped to the grid = the tap i1)
 * as sequential fma chains over the channels from +0, then per output pixel the 10-term fma chain over
 * the corner pairs (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3) with coefficient w_a*w_b, doubled
 * by one addition when a != b.  Cancellation guard: with t = the same chain over |<v_a,v_b>|, a pixel with
 * s < 2^-10 t (every negative rounding residue included) is evaluated in the EXACT order instead -- the four
 * taps of every channel combined as in halo_o_bilinear_f64, then the fma chain over the squares -- so the Gram
 * value is only used where it keeps >= 43 bits: |s_gram - s_exact| <= 4 C u 2^10 s (1.2e-10 at C = 256).
 * feat (C,h,w) f64 -> out (H,W) f64; mode 0: dist0 (radius), 1: sqrt (norm).
 * ------------------------------------------------------------------------- */
void halo_o_gram_radius(const double *feat, i64 C, i64 h, i64 w, i64 H, i64 W, int mode, double c, double *out)
{
    const i64 hwl = h * w;
    double *gram = (double *)malloc(sizeof(double) * 5 * (size_t)hwl);
    double *S = gram, *Hh = S + hwl, *Vv = Hh + hwl, *D1 = Vv + hwl, *D2 = D1 + hwl;
#pragma omp parallel for collapse(2) schedule(static)
    for (i64 y = 0; y < h; ++y)
        for (i64 x = 0; x < w; ++x) {
            const i64 ny = y + 1 < h - 1 ? y + 1 : h - 1, nx = x + 1 < w - 1 ? x + 1 : w - 1;
            double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0, g4 = 0.0;
            for (i64 ch = 0; ch < C; ++ch) {
                const double *pl = feat + ch * hwl;
                const double v0 = pl[y * w + x], r0 = pl[y * w + nx], v1 = pl[ny * w + x], r1 = pl[ny * w + nx];
                g0 = fma(v0, v0, g0);
                g1 = fma(v0, r0, g1);
                g2 = fma(v0, v1, g2);
                g3 = fma(v0, r1, g3);
                g4 = fma(r0, v1, g4);
            }
            S[y * w + x] = g0; Hh[y * w + x] = g1; Vv[y * w + x] = g2; D1[y * w + x] = g3; D2[y * w + x] = g4;
        }
    const double ks = k_sqrt(c), rks = 1.0 / ks;
    const double sh = H > 1 ? (double)(h - 1) / (double)(H - 1) : 0.0;
    const double sw = W > 1 ? (double)(w - 1) / (double)(W - 1) : 0.0;
#pragma omp parallel for schedule(static)
    for (i64 y = 0; y < H; ++y) {
        const double fy = sh * (double)y;
        i64 y0 = (i64)fy; if (y0 > h - 1) y0 = h - 1;
        const i64 y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const double ly1 = fy - (double)y0, ly0 = 1.0 - ly1;
        for (i64 x = 0; x < W; ++x) {
            const double fx = sw * (double)x;
            i64 x0 = (i64)fx; if (x0 > w - 1) x0 = w - 1;
            const i64 x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const double lx1 = fx - (double)x0, lx0 = 1.0 - lx1;
            const double wt[4] = {ly0 * lx0, ly0 * lx1, ly1 * lx0, ly1 * lx1};
            const i64 c00 = y0 * w + x0, c01 = y0 * w + x1, c10 = y1 * w + x0, c11 = y1 * w + x1;
            const double G[10] = {S[c00], Hh[c00], Vv[c00], D1[c00], S[c01], D2[c00], Vv[c01], S[c10], Hh[c10], S[c11]};
            double s = 0.0, t = 0.0;
            int k = 0;
            for (int a = 0; a < 4; ++a)
                for (int b = a; b < 4; ++b, ++k) {
                    double coef = wt[a] * wt[b];
                    if (b != a) coef = coef + coef;
                    s = fma(coef, G[k], s);
                    t = fma(coef, fabs(G[k]), t);
                }
            if (s < t * 0x1p-10) {                      /* cancellation guard: this pixel in the EXACT order (NaN stays NaN) */
                double acc = 0.0;
                for (i64 ch = 0; ch < C; ++ch) {
                    const double *pl = feat + ch * hwl;
                    const double top = fma(lx0, pl[c00], lx1 * pl[c01]), bot = fma(lx0, pl[c10], lx1 * pl[c11]);
                    const double a = fma(ly0, top, ly1 * bot);
                    acc = fma(a, a, acc);
                }
                s = acc;
            }
            out[y * W + x] = mode == 0 ? dist0_from_ssq_f64(s, ks, rks) : sqrt(s);
        }
    }
    free(gram);
}

/* ------------------------------------------------------------------------- *
 * FloatingRegionScore.forward (core/active/floating_region.py:129-217)
 * ------------------------------------------------------------------------- */

/* softmax over O classes at pixel i (floating_region.py:152); p[] has O entries */
static void softmax_px(const float *logit, i64 O, i64 hw, i64 i, float *p)
{
    float m = logit[i];
    for (i64 c = 1; c < O; ++c) { float v = logit[c * hw + i]; if (v > m) m = v; }
    float s = 0.0f;
    for (i64 c = 0; c < O; ++c) { p[c] = ho_expf(logit[c * hw + i] - m); s = s + p[c]; }
    for (i64 c = 0; c < O; ++c) p[c] = p[c] / s;
}
static i64 argmax_px(const float *p, i64 O)
{
    i64 b = 0;
    for (i64 c = 1; c < O; ++c) if (p[c] > p[b]) b = c;
    return b;
}
/* torch.sum over a strided dimension of float32 terms, in ATen's order (aten/src/ATen/native/cpu/SumKernel.cpp,
 * multi_row_sum; floating_region.py:72,119 reduce dim 0 / dim 1 of a tensor that is contiguous over H x W): a cascade
 * of four accumulators -- rows are added one by one into acc[0]; after every 2^lp rows acc[0] is flushed into acc[1],
 * after every 2^(2 lp) rows acc[1] into acc[2], and so on; lp = max(4, ceil(log2 n) / 4); at the end the remainder in
 * acc[0] takes acc[1], acc[2], acc[3] in turn.  For the 19 classes: (t16 + t17 + t18) + (t0 + ... + t15), each
 * bracket from +0 left to right.  Bit for bit torch's CPU result, for the vectorised columns and the scalar tail
 * alike (tests/test_aten_exact.py). */
static float cascade_sum_f32(const float *t, i64 n)
{
    i64 cl = 0;
    while (((i64)1 << cl) < n) ++cl;                               /* utils::CeilLog2 */
    const i64 lp = cl / 4 > 4 ? cl / 4 : 4, step = (i64)1 << lp, mask = step - 1;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    i64 i = 0;
    while (i + step <= n) {
        for (i64 j = 0; j < step; ++j, ++i) acc[0] = acc[0] + t[i];
        for (int j = 1; j < 4; ++j) {
            acc[j] = acc[j] + acc[j - 1];
            acc[j - 1] = 0.0f;
            if ((i & (mask << (j * lp))) != 0) break;
        }
    }
    for (; i < n; ++i) acc[0] = acc[0] + t[i];
    for (int j = 1; j < 4; ++j) acc[0] = acc[0] + acc[j];
    return acc[0];
}

/* torch.sum(t, dim=0) of a dense (n, hw) float32 array (test probe of the order above) */
void halo_o_sum_dim0(const float *t, i64 n, i64 hw, float *out)
{
#pragma omp parallel
    {
        float *col = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(static)
        for (i64 i = 0; i < hw; ++i) {
            for (i64 c = 0; c < n; ++c) col[c] = t[c * hw + i];
            out[i] = cascade_sum_f32(col, n);
        }
        free(col);
    }
}

/* sum_c -p*log(p+1e-6) / log(19)   (floating_region.py:72-76,123-127: the 19 is hard-coded) */
static float entropy_px(const float *p, i64 O)
{
    float tbuf[64], *t = O <= 64 ? tbuf : (float *)malloc(sizeof(float) * (size_t)O);
    for (i64 c = 0; c < O; ++c) t[c] = (-p[c]) * ho_logf(p[c] + 1e-6f);
    const float a = cascade_sum_f32(t, O);
    if (t != tbuf) free(t);
    return a / (float)log(19.0);
}

/* nn.Conv2d(padding_mode=...) (floating_region.py:49,63): index a tap outside [0, n) reads, -1 = nothing ('zeros').
 * torch pads the input (F.pad: 'reflect' mirrors without repeating the edge, 'replicate' repeats the edge, 'circular'
 * wraps) and convolves without padding.  0 zeros, 1 reflect, 2 replicate, 3 circular (HALO_PAD_* of include/halo_hip.h). */
static i64 pad_index(i64 t, i64 n, int mode)
{
    if (t >= 0 && t < n) return t;
    if (mode == 0) return -1;
    if (mode == 2) return t < 0 ? 0 : n - 1;
    if (mode == 1) return t < 0 ? -t : 2 * (n - 1) - t;
    return t < 0 ? t + n : t - n;
}

/* k x k all-ones conv, padding k/2 (floating_region.py:42-51,90): taps added in row-major order from +0.  That is bit for bit
 * what F.conv2d returns on the CPU whenever ATen hands the convolution to oneDNN -- for this 1 -> 1 channel 3 x 3 filter: every
 * image of more than 20480 pixels (aten/src/ATen/native/Convolution.cpp, use_mkldnn), i.e. every size the reference's pipeline
 * runs (tests/test_aten_exact.py: 101 x 203 up to 1024 x 2048).  At or below 20480 pixels ATen unfolds the image and calls
 * MKL's sgemm, whose summation order is MKL's own: the small fixtures agree with this order to an ulp or two, not bitwise. */
static void box_sum_f32(const float *in, float *out, i64 H, i64 W, int k, int pad)
{
    const int r = k / 2;
#pragma omp parallel for schedule(static)
    for (i64 y = 0; y < H; ++y)
        for (i64 x = 0; x < W; ++x) {
            float a = 0.0f;
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    const i64 yy = pad_index(y + dy, H, pad), xx = pad_index(x + dx, W, pad);
                    float v = (yy >= 0 && xx >= 0) ? in[yy * W + xx] : 0.0f;
                    a = a + v;
                }
            out[y * W + x] = a;
        }
}

void halo_o_box_sum(const float *in, float *out, i64 H, i64 W, int k, int pad) { box_sum_f32(in, out, H, W, k, pad); }

/* compute_region_impurity (floating_region.py:112-121): window class histogram -> entropy/log(K) */
static void region_impurity(const i64 *pred, i64 K, int k, i64 H, i64 W, float *imp, float *count, int pad)
{
    const int r = k / 2;
    const float logK = (float)log((double)K);
#pragma omp parallel
    {
        float *hist = (float *)malloc(sizeof(float) * (size_t)K * 2), *term = hist + K;
#pragma omp for schedule(static)
        for (i64 y = 0; y < H; ++y)
            for (i64 x = 0; x < W; ++x) {
                for (i64 c = 0; c < K; ++c) hist[c] = 0.0f;
                float cnt = 0.0f;
                for (int dy = -r; dy <= r; ++dy)
                    for (int dx = -r; dx <= r; ++dx) {
                        const i64 yy = pad_index(y + dy, H, pad), xx = pad_index(x + dx, W, pad);
                        if (yy >= 0 && xx >= 0) { hist[pred[yy * W + xx]] += 1.0f; cnt += 1.0f; }
                    }
                for (i64 c = 0; c < K; ++c) {                          /* an empty bin contributes (-0) * log(1e-6) = +0 */
                    const float d = hist[c] / cnt;
                    term[c] = hist[c] > 0.0f ? (-d) * ho_logf(d + 1e-6f) : 0.0f;
                }
                imp[y * W + x] = cascade_sum_f32(term, K) / logK;  /* torch.sum(dim=1), :119 */
                count[y * W + x] = cnt;
            }
        free(hist);
    }
}

/* normalize_map (floating_region.py:22-23) */
static void normalize_f32(float *x, i64 n)
{
    float mn = x[0], mx = x[0];
    int has_nan = 0;
    for (i64 i = 0; i < n; ++i) { if (x[i] != x[i]) has_nan = 1; if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    if (has_nan) { mn = NAN; mx = NAN; }                    /* torch min/max propagate NaN */
    const float den = (float)((double)mx - (double)mn);
    for (i64 i = 0; i < n; ++i) x[i] = (x[i] - mn) / den;
}
static void normalize_f64(double *x, i64 n)
{
    double mn = x[0], mx = x[0];
    int has_nan = 0;
    for (i64 i = 0; i < n; ++i) { if (x[i] != x[i]) has_nan = 1; if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    if (has_nan) { mn = NAN; mx = NAN; }
    const double den = mx - mn;
    for (i64 i = 0; i < n; ++i) x[i] = (x[i] - mn) / den;
}

/* quantize_uncert_map (floating_region.py:94-110) on an (H,W) radius map */
static void quantize_f64(double *r, i64 n, i64 K, i64 *pred)
{
    normalize_f64(r, n);
    for (i64 i = 0; i < n; ++i) r[i] = 1.0 - r[i];
    normalize_f64(r, n);
    const double lo = -0.5 + 1e-5, hi = (double)K - 0.5 - 1e-5;
    for (i64 i = 0; i < n; ++i) {
        double p = r[i] * (double)K - 0.5;
        if (p < lo) p = lo;
        if (p > hi) p = hi;
        pred[i] = (i64)rint(p);
    }
}
static void quantize_f32(float *r, i64 n, i64 K, i64 *pred)
{
    normalize_f32(r, n);
    for (i64 i = 0; i < n; ++i) r[i] = 1.0f - r[i];
    normalize_f32(r, n);
    const float lo = (float)(-0.5 + 1e-5), hi = (float)((double)K - 0.5 - 1e-5);
    for (i64 i = 0; i < n; ++i) {
        float p = r[i] * (float)K - 0.5f;
        if (p < lo) p = lo;
        if (p > hi) p = hi;
        /* float32 and K > 128: (float)(K - 0.5 - 1e-5) is K - 0.5, which rounds half-even to K -- one past the last bin, where the
         * reference's F.one_hot raises ("Class values must be smaller than num_classes").  The last bin takes it. */
        const float q = rintf(p);
        pred[i] = (i64)(q > (float)(K - 1) ? (float)(K - 1) : q);
    }
}

/* The whole forward.  C == 0 with feat != NULL: `feat` is the (H,W) float64 map of per-pixel radii (pur RADIUS / HYPER) or
 * norms (EUC_NORM) itself, e.g. from halo_o_gram_radius, instead of the embedding it would be reduced from.
 * Outputs: score / impurity in f64 when (pur is RADIUS|EUC_NORM and
 * feat is f64), else f32 -- the caller passes buffers of the right width and reads
 * *score_dtype.  unc_out is always f32.  ksize = entropy conv size; pksize = purity conv
 * size (3 when the module was built for 'hyper', floating_region.py:54-55).
 * Returns 0, or -1 for an unknown purity type (NotImplementedError, :199-202). */
int halo_o_floating_region_score(const float *logit, const void *feat, int feat_dtype, const i64 *gt,
                                 i64 O, i64 C, i64 H, i64 W, int unc_type, int pur_type, int normalize,
                                 int ksize, int pksize, i64 K, double c,
                                 void *score, void *impurity, float *unc_out, int *score_dtype)
{
    const i64 hw = H * W;
    const int pad = (normalize >> 8) & 3;                       /* flags word as in include/halo_hip.h: bit 0 normalise, bits 8-9 padding */
    normalize &= 1;
    if (pur_type < 0 || pur_type > HALO_PUR_EUC_NORM) return -1;
    float *ent = (float *)malloc(sizeof(float) * hw);
    float *cnt = (float *)malloc(sizeof(float) * hw);
    i64 *pred = (i64 *)malloc(sizeof(i64) * hw);
    const int need_argmax = pur_type == HALO_PUR_RIPU || pur_type == HALO_PUR_ORACLE_RIPU || unc_type == HALO_UNC_ORACLE_ACC;

    /* --- uncertainty (floating_region.py:158-163, 70-92) --- */
#pragma omp parallel
    {
        float *p = (float *)malloc(sizeof(float) * (size_t)O);
#pragma omp for schedule(static)
        for (i64 i = 0; i < hw; ++i) {
            softmax_px(logit, O, hw, i, p);
            i64 am = need_argmax ? argmax_px(p, O) : 0;
            if (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_PIXEL_ENTROPY) ent[i] = entropy_px(p, O);
            else if (unc_type == HALO_UNC_ORACLE_ACC) {
                i64 g = gt[i] == 255 ? am : gt[i];
                ent[i] = 1.0f - p[g];
            } else ent[i] = 0.0f;
            if (pur_type == HALO_PUR_RIPU) pred[i] = am;
            else if (pur_type == HALO_PUR_ORACLE_RIPU) pred[i] = gt[i] == 255 ? am : gt[i];
        }
        free(p);
    }
    if (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_ORACLE_ACC) box_sum_f32(ent, unc_out, H, W, ksize, pad);
    else for (i64 i = 0; i < hw; ++i) unc_out[i] = ent[i];   /* pixel_entropy: no conv; zeros: conv(0)=0 */

    /* --- purity (floating_region.py:165-202) --- */
    const double ks = k_sqrt(c), rks = 1.0 / ks;
    const int f64out = (pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM) && feat_dtype == HALO_F64;
    *score_dtype = f64out ? HALO_F64 : HALO_F32;
    double *imp64 = (double *)impurity;
    float *imp32 = (float *)impurity;
    if (pur_type == HALO_PUR_RIPU || pur_type == HALO_PUR_ORACLE_RIPU) {
        region_impurity(pred, O, pksize, H, W, imp32, cnt, pad);
    } else if (pur_type == HALO_PUR_HYPER) {
        if (feat_dtype == HALO_F64) {
            double *r = (double *)malloc(sizeof(double) * hw);
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < hw; ++i) r[i] = C == 0 ? ((const double *)feat)[i] : dist0_from_ssq_f64(ssq_f64((const double *)feat + i, C, hw), ks, rks);
            quantize_f64(r, hw, K, pred);
            free(r);
        } else {
            float *r = (float *)malloc(sizeof(float) * hw);
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < hw; ++i) r[i] = dist0_from_ssq_f32(ssq_f32((const float *)feat + i, C, hw), ks, rks);
            quantize_f32(r, hw, K, pred);
            free(r);
        }
        region_impurity(pred, K, pksize, H, W, imp32, cnt, pad);
    } else {
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < hw; ++i) {
            cnt[i] = 1.0f;
            if (pur_type == HALO_PUR_NONE) imp32[i] = 0.0f;
            else if (feat_dtype == HALO_F64 && C == 0) imp64[i] = ((const double *)feat)[i];
            else if (feat_dtype == HALO_F64) {
                double s = ssq_f64((const double *)feat + i, C, hw);
                imp64[i] = pur_type == HALO_PUR_RADIUS ? dist0_from_ssq_f64(s, ks, rks) : sqrt(s);
            } else {
                float s = ssq_f32((const float *)feat + i, C, hw);
                imp32[i] = pur_type == HALO_PUR_RADIUS ? dist0_from_ssq_f32(s, ks, rks) : sqrtf(s);
            }
        }
    }
    /* prediction_uncertainty = region_uncertainty / count (:204) */
    for (i64 i = 0; i < hw; ++i) unc_out[i] = unc_out[i] / cnt[i];
    if (normalize) {                                                   /* :206-208 */
        normalize_f32(unc_out, hw);
        if (f64out) normalize_f64(imp64, hw); else normalize_f32(imp32, hw);
    }
    /* score = region_impurity * prediction_uncertainty (:210), f64*f32 -> f64 */
    if (f64out) for (i64 i = 0; i < hw; ++i) ((double *)score)[i] = imp64[i] * (double)unc_out[i];
    else for (i64 i = 0; i < hw; ++i) ((float *)score)[i] = imp32[i] * unc_out[i];
    free(ent); free(cnt); free(pred);
    return 0;
}

/* ---- helper methods the reference exposes by convention (floating_region.py:70-127) ---- */

/* compute_region_uncertainty / compute_pixel_entropy on softmax probabilities p (O,H,W) */
void halo_o_uncertainty_from_probs(const float *p, const i64 *gt, i64 O, i64 H, i64 W, int unc_type, int ksize,
                                   int do_box, float *out)
{
    const i64 hw = H * W;
    float *ent = (float *)malloc(sizeof(float) * hw);
#pragma omp parallel
    {
        float *q = (float *)malloc(sizeof(float) * (size_t)O);
#pragma omp for schedule(static)
        for (i64 i = 0; i < hw; ++i) {
            for (i64 c = 0; c < O; ++c) q[c] = p[c * hw + i];
            if (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_PIXEL_ENTROPY) ent[i] = entropy_px(q, O);
            else if (unc_type == HALO_UNC_ORACLE_ACC) { i64 g = gt[i] == 255 ? argmax_px(q, O) : gt[i]; ent[i] = 1.0f - q[g]; }
            else ent[i] = 0.0f;
        }
        free(q);
    }
    if (do_box & 1) box_sum_f32(ent, out, H, W, ksize, (do_box >> 8) & 3);        /* bits 8-9: padding mode */
    else for (i64 i = 0; i < hw; ++i) out[i] = ent[i];
    free(ent);
}

/* torch.softmax(logit, dim=0) as the contract computes it */
void halo_o_softmax(const float *logit, i64 O, i64 H, i64 W, float *p)
{
    const i64 hw = H * W;
#pragma omp parallel
    {
        float *q = (float *)malloc(sizeof(float) * (size_t)O);
#pragma omp for schedule(static)
        for (i64 i = 0; i < hw; ++i) { softmax_px(logit, O, hw, i, q); for (i64 c = 0; c < O; ++c) p[c * hw + i] = q[c]; }
        free(q);
    }
}

void halo_o_region_impurity(const i64 *pred, i64 K, int k, i64 H, i64 W, float *imp, float *count, int pad)
{
    region_impurity(pred, K, k, H, W, imp, count, pad);
}

void halo_o_quantize(const void *feat, int feat_dtype, i64 C, i64 H, i64 W, i64 K, double c, i64 *pred)
{
    const i64 hw = H * W;
    const double ks = k_sqrt(c), rks = 1.0 / ks;
    if (feat_dtype == HALO_F64) {
        double *r = (double *)malloc(sizeof(double) * hw);
        for (i64 i = 0; i < hw; ++i) r[i] = dist0_from_ssq_f64(ssq_f64((const double *)feat + i, C, hw), ks, rks);
        quantize_f64(r, hw, K, pred);
        free(r);
    } else {
        float *r = (float *)malloc(sizeof(float) * hw);
        for (i64 i = 0; i < hw; ++i) r[i] = dist0_from_ssq_f32(ssq_f32((const float *)feat + i, C, hw), ks, rks);
        quantize_f32(r, hw, K, pred);
        free(r);
    }
}

/* ------------------------------------------------------------------------- *
 * select_pixels_to_label (core/active/build.py:27-64)
 * torch.max ordering: NaN beats everything; ties keep the FIRST occurrence, so the
 * two-stage max over dim 0 then dim 0 picks the smallest w, then the smallest h.
 * picks (n,3) f64 rows (h, w, value) in selection order; returns the number picked.
 * ------------------------------------------------------------------------- */
#define BETTER(a, b) (((a) != (a)) ? !((b) != (b)) : (!((b) != (b)) && (a) > (b)))

i64 halo_o_select(void *score, int dtype, i64 H, i64 W, i64 n_regions, i64 active_radius, i64 mask_radius,
                  u8 *active, u8 *selected, i64 *active_mask, const i64 *gt, double *picks)
{
    double *colv = (double *)malloc(sizeof(double) * W);
    i64 *colh = (i64 *)malloc(sizeof(i64) * W);
    i64 np_ = 0;
    for (i64 it = 0; it < n_regions; ++it) {
        /* values, indices_h = torch.max(score, dim=0)   (:38) */
#pragma omp parallel for schedule(static)
        for (i64 x = 0; x < W; ++x) {
            double bv = dtype == HALO_F64 ? ((double *)score)[x] : (double)((float *)score)[x];
            i64 bh = 0;
            for (i64 y = 1; y < H; ++y) {
                double v = dtype == HALO_F64 ? ((double *)score)[y * W + x] : (double)((float *)score)[y * W + x];
                if (BETTER(v, bv)) { bv = v; bh = y; }
            }
            colv[x] = bv; colh[x] = bh;
        }
        /* max_value, indices_w = torch.max(values, dim=0)   (:39) */
        i64 w = 0;
        for (i64 x = 1; x < W; ++x) if (BETTER(colv[x], colv[w])) w = x;
        const double mv = colv[w];
        if (mv == -INFINITY) break;                                   /* :40-41 */
        const i64 h = colh[w];
        const i64 as_w = w - active_radius >= 0 ? w - active_radius : 0, as_h = h - active_radius >= 0 ? h - active_radius : 0;
        i64 ae_w = w + active_radius + 1, ae_h = h + active_radius + 1;
        const i64 ms_w = w - mask_radius >= 0 ? w - mask_radius : 0, ms_h = h - mask_radius >= 0 ? h - mask_radius : 0;
        i64 me_w = w + mask_radius + 1, me_h = h + mask_radius + 1;
        if (ae_w > W) ae_w = W;
        if (ae_h > H) ae_h = H;
        if (me_w > W) me_w = W;
        if (me_h > H) me_h = H;
        for (i64 y = ms_h; y < me_h; ++y)
            for (i64 x = ms_w; x < me_w; ++x) {
                if (dtype == HALO_F64) ((double *)score)[y * W + x] = -INFINITY; else ((float *)score)[y * W + x] = -INFINITY;
                active[y * W + x] = 1;
            }
        for (i64 y = as_h; y < ae_h; ++y)
            for (i64 x = as_w; x < ae_w; ++x) { selected[y * W + x] = 1; active_mask[y * W + x] = gt[y * W + x]; }
        /* the table's value column holds the score's class under the selection order (-0 ties with +0, every NaN is one value):
         * +0 for either zero, the canonical quiet NaN for any NaN -- what halo_greedy_select reports (the reference keeps no
         * table; found by tests/fuzz_parity.py on an all -0.0 map) */
        double rv = mv == 0.0 ? 0.0 : mv;
        if (mv != mv) { const unsigned long long qn = 0x7ff8000000000000ull; memcpy(&rv, &qn, 8); }
        picks[np_ * 3 + 0] = (double)h; picks[np_ * 3 + 1] = (double)w; picks[np_ * 3 + 2] = rv;
        ++np_;
    }
    free(colv); free(colh);
    return np_;
}
