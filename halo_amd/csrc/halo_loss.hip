// Training-side 3x3-window losses of the reference as fused HIP kernels:
//   NegativeLearningLoss   core/loss/negative_learning_loss.py:6-16
//   LocalConsistentLoss    core/loss/local_consistent_loss.py:5-17
//     = LocalDiscrepancy (softmax, 3x3 replicate-padded mean, l1 | kl)   core/loss/boundary.py:64-103
//     + DetectSPBoundary (8-neighbour Laplacian of the label map != 0)    core/loss/boundary.py:6-61
// All Euclidean, float32 tensors like the reference; sums are accumulated in float64 per block and
// finished in a fixed order, so a loss value is reproducible run to run.  Forward values and gradients
// are pinned to the reference's own autograd (tests/golden/losses.npz).
#include "halo_common.hpp"
#include "halo_devmath.hpp"

namespace halo {

constexpr int LTPB = 256;

__device__ __forceinline__ void block_sum2(double a, double b, double *out2)
{
    __shared__ double sa[LTPB / 64], sb[LTPB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < LTPB / 64; ++i) { a += sa[i]; b += sb[i]; }
        out2[0] = a;
        out2[1] = b;
    }
}

// partials (nblk, 2) -> sums[2], fixed order
__global__ void __launch_bounds__(LTPB) k_sum2_finalize(const double *__restrict__ partials, int nblk, double *__restrict__ sums)
{
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += LTPB) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    block_sum2(a, b, sums);
}

// ---------------------------------------------------------------- NegativeLearningLoss
// forward: sums = { sum -mask*log(1 - p + 1e-6), sum mask },  mask = p < threshold
// (n4 = n / 4 sixteen-byte groups when the tensor allows it -- round 5: the scalar grid-stride loop ran one dependent 4-byte load
//  per trip; the element order inside a thread's double sums is unchanged in the scalar tail, regrouped in the vector body)
typedef float f4_l __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(LTPB) k_negative_fwd(const float *__restrict__ p, long long n, long long n4, float thr, double *__restrict__ partials)
{
    double s = 0.0, c = 0.0;
    const f4_l *p4 = reinterpret_cast<const f4_l *>(p);
    for (long long i = (long long)blockIdx.x * LTPB + threadIdx.x; i < n4; i += (long long)gridDim.x * LTPB) {
        const f4_l v = __builtin_nontemporal_load(p4 + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[e] < thr) { s += (double)(-det_logf((1.0f - v[e]) + 1e-6f)); c += 1.0; }
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * LTPB + threadIdx.x; i < n; i += (long long)gridDim.x * LTPB) {
        const float v = p[i];
        if (v < thr) { s += (double)(-det_logf((1.0f - v) + 1e-6f)); c += 1.0; }
    }
    block_sum2(s, c, partials + 2 * blockIdx.x);
}
// backward: gp = g * mask / ((1 - p + 1e-6) * count)
__global__ void __launch_bounds__(LTPB) k_negative_bwd(const float *__restrict__ p, long long n, long long n4, float thr, const double *__restrict__ sums,
                                                       const float *__restrict__ gloss, float *__restrict__ gp)
{
    const long long i = (long long)blockIdx.x * LTPB + threadIdx.x;
    const float scale = (float)((double)gloss[0] / sums[1]);
    if (i < n4) {
        const f4_l v = __builtin_nontemporal_load(reinterpret_cast<const f4_l *>(p) + i);
        f4_l o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] < thr ? scale / ((1.0f - v[e]) + 1e-6f) : 0.0f;
        __builtin_nontemporal_store(o, reinterpret_cast<f4_l *>(gp) + i);
    }
    const long long q = 4 * n4 + i;                                       // scalar remainder: n % 4 elements, or all n when n4 = 0 (unaligned)
    if (q < n) {
        const float v = p[q];
        gp[q] = v < thr ? scale / ((1.0f - v) + 1e-6f) : 0.0f;
    }
}

// ---------------------------------------------------------------- softmax over the class planes (B,O,hw)
__global__ void __launch_bounds__(LTPB) k_softmax_nchw(const float *__restrict__ x, int O, long long hw, float *__restrict__ p)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * LTPB + threadIdx.x;
    if (i >= hw) return;
    const float *xb = x + (size_t)b * O * hw + i;
    float m = xb[0];
    for (int c = 1; c < O; ++c) { const float v = xb[(size_t)c * hw]; m = v > m ? v : m; }
    float s = 0.0f;
    for (int c = 0; c < O; ++c) s = s + det_expf(xb[(size_t)c * hw] - m);
    float *pb = p + (size_t)b * O * hw + i;
    for (int c = 0; c < O; ++c) pb[(size_t)c * hw] = det_expf(xb[(size_t)c * hw] - m) / s;
}

// The same values with every class of FOUR consecutive pixels in registers (O <= OMAX, hw % 4 == 0, 16-byte aligned planes): the
// planes are read once with O sixteen-byte loads in flight instead of three dependent 4-byte walks (round 5: 124 MB in ~0.2 ms).
template <int OMAX>
__global__ void __launch_bounds__(LTPB) k_softmax_nchw_v4(const float *__restrict__ x, int O, long long hw, float *__restrict__ p)
{
    const int b = blockIdx.y;
    const long long i4 = (long long)blockIdx.x * LTPB + threadIdx.x, hw4 = hw >> 2;
    if (i4 >= hw4) return;
    const f4_l *xb = reinterpret_cast<const f4_l *>(x + (size_t)b * O * hw) + i4;
    f4_l v[OMAX];
#pragma unroll
    for (int c = 0; c < OMAX; ++c)
        if (c < O) v[c] = __builtin_nontemporal_load(xb + (size_t)c * hw4);
    f4_l m = v[0];
#pragma unroll
    for (int c = 1; c < OMAX; ++c)
        if (c < O)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = v[c][e] > m[e] ? v[c][e] : m[e];
    f4_l sden = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < OMAX; ++c)
        if (c < O)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[c][e] = det_expf(v[c][e] - m[e]); sden[e] = sden[e] + v[c][e]; }
    f4_l *pb = reinterpret_cast<f4_l *>(p + (size_t)b * O * hw) + i4;
#pragma unroll
    for (int c = 0; c < OMAX; ++c)
        if (c < O) {
            f4_l o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[c][e] / sden[e];
            pb[(size_t)c * hw4] = o;
        }
}

// semantic boundary & valid label (boundary.py:48-61 with zero padding, local_consistent_loss.py:14-15)
__device__ __forceinline__ bool lcl_mask(const long long *__restrict__ lab, int h, int w, int y, int x)
{
    const long long c = lab[(size_t)y * w + x];
    if (c == 255) return false;
    float acc = 8.0f * (float)c;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (dy == 0 && dx == 0) continue;
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) acc -= (float)lab[(size_t)yy * w + xx];
        }
    return (long long)acc != 0;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x mod 8), and a 3x3 window kernel re-reads the rows above and below its
// own: with the plain blockIdx -> pixel-chunk map vertically adjacent chunks sit on different XCDs and each XCD's L2 fetches the same
// rows again.  Here XCD k walks one contiguous eighth of the chunks (whole groups of 8 only; the remainder keeps the plain map):
// k_lcl_bwd 178 -> 158 us at 2 x 19 x 640 x 1280.  (A strip-walking backward like k_lcl_fwd_strip -- separable weights, one
// carried sum per class and row -- measured 340 us: 202 registers, two waves per SIMD, eight rows in series per thread; not kept.)
__device__ __forceinline__ long long xcd_contiguous_chunk(unsigned bid, unsigned nblk)
{
    const unsigned per = nblk >> 3;
    return bid < (per << 3) ? (long long)(bid & 7u) * per + (bid >> 3) : (long long)bid;
}

// Per masked pixel: l = sum_c |p - mean| (l1) or sum_c p*log(p/(mean+1e-6)+1e-6) (kl), mean = 3x3 replicate-padded box
// mean; block partial sums of (l, 1).  When ca != nullptr also writes the mask as one byte per pixel and, AT MASKED PIXELS ONLY,
// per class a = dl/dp (direct) and b = dl/dmean for the backward pass: the boundary is a few per cent of the pixels, and round 4's
// dense coefficient maps were 2 x 124 MB of zeros written here and read nine times over in the backward.
__global__ void __launch_bounds__(LTPB) k_lcl_fwd(const float *__restrict__ p, const long long *__restrict__ label, int O, int h, int w,
                                                  int kl, double *__restrict__ partials, float *__restrict__ ca, float *__restrict__ cb,
                                                  unsigned char *__restrict__ mask)
{
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const long long i = (long long)blockIdx.x * LTPB + threadIdx.x;     // (the XCD-contiguous map of the backward kernel measured SLOWER here: 131 -> 199 us)
    double ls = 0.0, cnt = 0.0;
    if (i < hw) {
        const int y = (int)(i / w), x = (int)(i % w);
        const bool m = lcl_mask(label + (size_t)b * hw, h, w, y, x);
        if (mask) mask[(size_t)b * hw + i] = m ? 1 : 0;
        if (m) {
            const float *pb = p + (size_t)b * O * hw;
            const int ym = clampi(y - 1, 0, h - 1), yp = clampi(y + 1, 0, h - 1), xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
            float l = 0.0f;
#pragma unroll 4
            for (int c = 0; c < O; ++c) {
                const float *pc = pb + (size_t)c * hw;
                const float *r0 = pc + (size_t)ym * w, *r1 = pc + (size_t)y * w, *r2 = pc + (size_t)yp * w;
                // the nine taps in the order dy = -1..1, dx = -1..1 of the replicate-padded window
                const float t0 = r0[xm], t1 = r0[x], t2 = r0[xp], t3 = r1[xm], t4 = r1[x], t5 = r1[xp], t6 = r2[xm], t7 = r2[x], t8 = r2[xp];
                float mean = 0.0f;
                mean = __builtin_fmaf(t0, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t1, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t2, 1.0f / 9.0f, mean);
                mean = __builtin_fmaf(t3, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t4, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t5, 1.0f / 9.0f, mean);
                mean = __builtin_fmaf(t6, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t7, 1.0f / 9.0f, mean); mean = __builtin_fmaf(t8, 1.0f / 9.0f, mean);
                const float pv = t4;
                float a, bb;
                if (!kl) {
                    const float d = pv - mean;
                    l = l + (d < 0.0f ? -d : d);
                    a = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                    bb = -a;
                } else {
                    const float me = mean + 1e-6f, r = pv / me, re = r + 1e-6f, lg = det_logf(re);
                    l = l + pv * lg;
                    a = lg + (pv / me) / re;                 // d/dp [p log(p/me + eps)]
                    bb = -(pv * pv) / (me * me * re);        // d/dmean
                }
                if (ca) { ca[((size_t)b * O + c) * hw + i] = a; cb[((size_t)b * O + c) * hw + i] = bb; }
            }
            ls = (double)l; cnt = 1.0;
        }
    }
    block_sum2(ls, cnt, partials + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// The same forward with a workgroup walking LCL_ROWS rows of a 256-column strip, classes in groups of CG: a thread keeps the taps of
// the two rows above its next one in registers (6 per class), so every row of p is loaded ONCE per strip (+ 2 halo rows per
// LCL_ROWS) instead of by the three workgroups that own it and its neighbours -- which the plain pixel-chunk map puts on three
// different XCDs, i.e. three L2 misses per element.  Same statements per (pixel, class) as k_lcl_fwd; the per-pixel float32 sum over
// the classes is formed per group and the groups are added in float64 (bits differ from k_lcl_fwd's in the last place of a sum).
constexpr int LCL_ROWS = 8, LCL_CG = 10;
__global__ void __launch_bounds__(LTPB) k_lcl_fwd_strip(const float *__restrict__ p, const long long *__restrict__ label, int O, int h, int w,
                                                        int kl, double *__restrict__ partials, float *__restrict__ ca, float *__restrict__ cb,
                                                        unsigned char *__restrict__ mask)
{
    const int b = blockIdx.z, x = blockIdx.x * LTPB + threadIdx.x, y0 = blockIdx.y * LCL_ROWS;
    const int y1 = y0 + LCL_ROWS < h ? y0 + LCL_ROWS : h;
    const long long hw = (long long)h * w;
    const bool col = x < w;
    const int xc = col ? x : w - 1, xm = clampi(xc - 1, 0, w - 1), xp = clampi(xc + 1, 0, w - 1);
    const long long *lab = label + (size_t)b * hw;
    unsigned mbits = 0;                                                  // the strip's mask bits of this column, row y0 + k in bit k
    for (int y = y0; y < y1; ++y) {
        const bool m = col && lcl_mask(lab, h, w, y, xc);
        mbits |= (m ? 1u : 0u) << (y - y0);
        if (mask && col) mask[(size_t)b * hw + (size_t)y * w + x] = m ? 1 : 0;
    }
    double ls = 0.0, cnt = (double)__builtin_popcount(mbits);
    for (int c0 = 0; c0 < O; c0 += LCL_CG) {
        float r0[LCL_CG][3], r1[LCL_CG][3], r2[LCL_CG][3];
        const float *pb = p + ((size_t)b * O + c0) * hw;
        auto load_row = [&](float (&r)[LCL_CG][3], int yy) {
            const size_t ro = (size_t)clampi(yy, 0, h - 1) * w;
#pragma unroll
            for (int q = 0; q < LCL_CG; ++q)
                if (c0 + q < O) { const float *pc = pb + (size_t)q * hw + ro; r[q][0] = pc[xm]; r[q][1] = pc[xc]; r[q][2] = pc[xp]; }
        };
        load_row(r0, y0 - 1);
        load_row(r1, y0);
        for (int y = y0; y < y1; ++y) {
            load_row(r2, y + 1);
            if ((mbits >> (y - y0)) & 1u) {
                const size_t i = (size_t)y * w + x;
                float l = 0.0f;
#pragma unroll
                for (int q = 0; q < LCL_CG; ++q) {
                    if (c0 + q < O) {
                        float mean = 0.0f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) mean = __builtin_fmaf(r0[q][k], 1.0f / 9.0f, mean);
#pragma unroll
                        for (int k = 0; k < 3; ++k) mean = __builtin_fmaf(r1[q][k], 1.0f / 9.0f, mean);
#pragma unroll
                        for (int k = 0; k < 3; ++k) mean = __builtin_fmaf(r2[q][k], 1.0f / 9.0f, mean);
                        const float pv = r1[q][1];
                        float a, bb;
                        if (!kl) {
                            const float d = pv - mean;
                            l = l + (d < 0.0f ? -d : d);
                            a = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                            bb = -a;
                        } else {
                            const float me = mean + 1e-6f, r = pv / me, re = r + 1e-6f, lg = det_logf(re);
                            l = l + pv * lg;
                            a = lg + (pv / me) / re;                     // d/dp [p log(p/me + eps)]
                            bb = -(pv * pv) / (me * me * re);            // d/dmean
                        }
                        if (ca) { ca[((size_t)b * O + c0 + q) * hw + i] = a; cb[((size_t)b * O + c0 + q) * hw + i] = bb; }
                    }
                }
                ls += (double)l;
            }
#pragma unroll
            for (int q = 0; q < LCL_CG; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) { r0[q][k] = r1[q][k]; r1[q][k] = r2[q][k]; }
        }
    }
    block_sum2(ls, cnt, partials + 2 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x));
}

// gx = softmax-backward( gp ),  gp_c(j) = a_c(j) + sum_{i in 3x3(j)} mult(i -> j)/9 * b_c(i),  scaled by g / count.
// mult(i -> j) = number of taps of i's replicate-padded window that land on j.  a / b exist at masked pixels only (mask byte map):
// a pixel with no masked pixel in its 3x3 neighbourhood has gp = 0 for every class, hence gx = 0 -- it writes its zeros and reads
// nothing else (most pixels).  gp of the O classes stays in registers between the dot product and the write (O <= OMAX).
template <int OMAX>
__global__ void __launch_bounds__(LTPB) k_lcl_bwd(const float *__restrict__ p, const float *__restrict__ ca, const float *__restrict__ cb,
                                                  const unsigned char *__restrict__ mask, int O, int h, int w, const double *__restrict__ sums,
                                                  const float *__restrict__ gloss, float *__restrict__ gx)
{
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const long long j = xcd_contiguous_chunk(blockIdx.x, gridDim.x) * LTPB + threadIdx.x;
    if (j >= hw) return;
    const int y = (int)(j / w), x = (int)(j % w);
    const size_t base = (size_t)b * O * hw;
    const unsigned char *mb = mask + (size_t)b * hw;
    // neighbour (y + ny, x + nx): its weight is multiplicity / 9 where it exists AND is masked, else 0; the multiplicity is a
    // product of a row and a column count (how many of the neighbour's three clamped taps along that axis land on this pixel:
    // 1 in the interior, 2 for an edge pixel seen from itself).  Offsets are clamped into the image so that every load below is
    // unconditional (no branch per tap: round 5's first version spent its time in 171 of them per pixel); a zero weight SELECTS
    // zero, because b holds no value at unmasked pixels.
    int cy[3], cx[3];
#pragma unroll
    for (int n = -1; n <= 1; ++n) {
        const int iy = y + n, ix = x + n;
        int ky = 0, kx = 0;
#pragma unroll
        for (int d = -1; d <= 1; ++d) { ky += clampi(iy + d, 0, h - 1) == y ? 1 : 0; kx += clampi(ix + d, 0, w - 1) == x ? 1 : 0; }
        cy[n + 1] = (iy >= 0 && iy < h) ? ky : 0;
        cx[n + 1] = (ix >= 0 && ix < w) ? kx : 0;
    }
    float wgt[9];
    int off[9];
    bool any = false;
#pragma unroll
    for (int ny = -1; ny <= 1; ++ny)
#pragma unroll
        for (int nx = -1; nx <= 1; ++nx) {
            const int o_ = clampi(y + ny, 0, h - 1) * w + clampi(x + nx, 0, w - 1);
            const int mult = mb[o_] ? cy[ny + 1] * cx[nx + 1] : 0;      // (a clamped duplicate of an existing pixel has count 0)
            off[(ny + 1) * 3 + nx + 1] = o_;
            wgt[(ny + 1) * 3 + nx + 1] = (float)mult * (1.0f / 9.0f);
            any = any || mult != 0;
        }
    if (!any) {
#pragma unroll 4
        for (int c = 0; c < O; ++c) gx[base + (size_t)c * hw + j] = 0.0f;
        return;
    }
    const double cnt = sums[1];
    const float scale = cnt > 0.0 ? (float)((double)gloss[0] / cnt) : 0.0f;       // mean over an empty selection: zero gradient
    const bool mj = mb[j] != 0;
    float gpv[OMAX], pvv[OMAX], dot = 0.0f;
#pragma unroll
    for (int c = 0; c < OMAX; ++c) {
        if (c < O) {
            const float *bc = cb + base + (size_t)c * hw;
            const float av = ca[base + (size_t)c * hw + j];
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] = bc[off[q]];
            float gp = mj ? av : 0.0f;
#pragma unroll
            for (int q = 0; q < 9; ++q) gp = __builtin_fmaf(wgt[q], wgt[q] != 0.0f ? t[q] : 0.0f, gp);
            gpv[c] = gp;
            pvv[c] = p[base + (size_t)c * hw + j];
            dot = __builtin_fmaf(pvv[c], gp, dot);
        }
    }
#pragma unroll
    for (int c = 0; c < OMAX; ++c)
        if (c < O) gx[base + (size_t)c * hw + j] = scale * (pvv[c] * (gpv[c] - dot));
}
// any number of classes: the same statements with gp recomputed in the second pass
__global__ void __launch_bounds__(LTPB) k_lcl_bwd_any(const float *__restrict__ p, const float *__restrict__ ca, const float *__restrict__ cb,
                                                      const unsigned char *__restrict__ mask, int O, int h, int w, const double *__restrict__ sums,
                                                      const float *__restrict__ gloss, float *__restrict__ gx)
{
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const long long j = (long long)blockIdx.x * LTPB + threadIdx.x;
    if (j >= hw) return;
    const int y = (int)(j / w), x = (int)(j % w);
    const size_t base = (size_t)b * O * hw;
    const unsigned char *mb = mask + (size_t)b * hw;
    float wgt[3][3];
    for (int ny = -1; ny <= 1; ++ny)
        for (int nx = -1; nx <= 1; ++nx) {
            const int iy = y + ny, ix = x + nx;
            int mult = 0;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w && mb[(size_t)iy * w + ix])
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        mult += (clampi(iy + dy, 0, h - 1) == y && clampi(ix + dx, 0, w - 1) == x) ? 1 : 0;
            wgt[ny + 1][nx + 1] = (float)mult * (1.0f / 9.0f);
        }
    const double cnt = sums[1];
    const float scale = cnt > 0.0 ? (float)((double)gloss[0] / cnt) : 0.0f;
    const bool mj = mb[j] != 0;
    float dot = 0.0f;
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < O; ++c) {
            const float *bc = cb + base + (size_t)c * hw;
            float gp = mj ? ca[base + (size_t)c * hw + j] : 0.0f;
            for (int ny = -1; ny <= 1; ++ny)
                for (int nx = -1; nx <= 1; ++nx) {
                    const float wv = wgt[ny + 1][nx + 1];
                    if (wv != 0.0f) gp = __builtin_fmaf(wv, bc[(size_t)(y + ny) * w + (x + nx)], gp);
                }
            const float pv = p[base + (size_t)c * hw + j];
            if (pass == 0) dot = __builtin_fmaf(pv, gp, dot);
            else gx[base + (size_t)c * hw + j] = scale * (pv * (gp - dot));
        }
}

}  // namespace halo

using namespace halo;

extern "C" size_t halo_loss_workspace_bytes(int64_t n_pixels)
{
    if (n_pixels <= 0) return 0;
    return (size_t)(cdiv(n_pixels, 8) + 8) * 2 * sizeof(double) + 512;       // covers the strip kernel's workgroups (256 columns x 8 rows, ragged edges)
}

// NegativeLearningLoss.forward: sums[0] = sum of loss items, sums[1] = number of selected entries (float64, device)
extern "C" int halo_negative_learning_fwd(const float *p, int64_t n, double threshold, double *sums, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    if (!p || !sums || n <= 0) return fail(HALO_E_ARG, "halo_negative_learning_fwd: null/empty argument");
    const int nblk = (int)(cdiv(n, LTPB) < 2048 ? cdiv(n, LTPB) : 2048);
    if (!workspace || workspace_bytes < (size_t)nblk * 16 + 256) return fail(HALO_E_WORKSPACE, "halo_negative_learning_fwd: workspace too small");
    double *part = (double *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t st = (hipStream_t)stream;
    const long long n4 = ((uintptr_t)p % 16) == 0 ? n / 4 : 0;
    hipLaunchKernelGGL(k_negative_fwd, dim3(nblk), dim3(LTPB), 0, st, p, (long long)n, n4, (float)threshold, part);
    hipLaunchKernelGGL(k_sum2_finalize, dim3(1), dim3(LTPB), 0, st, (const double *)part, nblk, sums);
    return check_launch("halo_negative_learning_fwd");
}

extern "C" int halo_negative_learning_bwd(const float *p, int64_t n, double threshold, const double *sums, const float *gloss,
                                          float *gp, void *stream)
{
    if (!p || !sums || !gloss || !gp || n <= 0) return fail(HALO_E_ARG, "halo_negative_learning_bwd: null/empty argument");
    const long long n4 = (((uintptr_t)p | (uintptr_t)gp) % 16) == 0 ? n / 4 : 0;
    const long long nthreads = n4 > 0 ? (n4 > n - 4 * n4 ? n4 : n - 4 * n4) : n;
    hipLaunchKernelGGL(k_negative_bwd, dim3((unsigned)cdiv(nthreads, LTPB)), dim3(LTPB), 0, (hipStream_t)stream, p, (long long)n, n4, (float)threshold, sums, gloss, gp);
    return check_launch("halo_negative_learning_bwd");
}

// LocalConsistentLoss.forward: x (B,O,h,w) logits, label (B,h,w) i64 -> p (softmax, kept for backward), sums = {sum l, count};
// coef_a / coef_b (B,O,h,w) receive dl/dp and dl/dmean when not NULL.  kl: 0 = 'l1', 1 = 'kl'.
extern "C" int halo_local_consistent_fwd(const float *x, const int64_t *label, int64_t B, int64_t O, int64_t h, int64_t w, int kl,
                                         float *p, double *sums, float *coef_a, float *coef_b, uint8_t *mask, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    if (!x || !label || !p || !sums || B <= 0 || O <= 0 || h <= 0 || w <= 0) return fail(HALO_E_ARG, "halo_local_consistent_fwd: null/empty argument");
    if ((coef_a == nullptr) != (coef_b == nullptr) || (coef_a == nullptr) != (mask == nullptr))
        return fail(HALO_E_ARG, "halo_local_consistent_fwd: coef_a, coef_b and mask go together");
    const long long hw = (long long)h * w;
    const int nb = (int)cdiv(hw, LTPB);
    const bool strip = getenv("HALO_LCL_PLAIN") == nullptr && B <= 65535 && cdiv(h, LCL_ROWS) <= 65535;      // A/B switch: the one-pixel-per-thread forward
    const dim3 gs((unsigned)cdiv(w, LTPB), (unsigned)cdiv(h, LCL_ROWS), (unsigned)B);
    const int nblk = strip ? (int)(gs.x * gs.y * gs.z) : nb * (int)B;
    if (!workspace || workspace_bytes < (size_t)nblk * 16 + 256) return fail(HALO_E_WORKSPACE, "halo_local_consistent_fwd: workspace too small");
    double *part = (double *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nb, (unsigned)B);
    if (O <= 20 && hw % 4 == 0 && (((uintptr_t)x | (uintptr_t)p) % 16) == 0)
        hipLaunchKernelGGL((k_softmax_nchw_v4<20>), dim3((unsigned)cdiv(hw / 4, LTPB), (unsigned)B), dim3(LTPB), 0, st, x, (int)O, hw, p);
    else
        hipLaunchKernelGGL(k_softmax_nchw, grid, dim3(LTPB), 0, st, x, (int)O, hw, p);
    if (strip)
        hipLaunchKernelGGL(k_lcl_fwd_strip, gs, dim3(LTPB), 0, st, (const float *)p, (const long long *)label, (int)O, (int)h, (int)w, kl, part, coef_a, coef_b,
                           (unsigned char *)mask);
    else
        hipLaunchKernelGGL(k_lcl_fwd, grid, dim3(LTPB), 0, st, (const float *)p, (const long long *)label, (int)O, (int)h, (int)w, kl, part, coef_a, coef_b,
                           (unsigned char *)mask);
    hipLaunchKernelGGL(k_sum2_finalize, dim3(1), dim3(LTPB), 0, st, (const double *)part, nblk, sums);
    return check_launch("halo_local_consistent_fwd");
}

extern "C" int halo_local_consistent_bwd(const float *p, const float *coef_a, const float *coef_b, const uint8_t *mask, int64_t B, int64_t O,
                                         int64_t h, int64_t w, const double *sums, const float *gloss, float *gx, void *stream)
{
    if (!p || !coef_a || !coef_b || !mask || !sums || !gloss || !gx || B <= 0 || O <= 0 || h <= 0 || w <= 0)
        return fail(HALO_E_ARG, "halo_local_consistent_bwd: null/empty argument");
    dim3 grid((unsigned)cdiv(h * w, LTPB), (unsigned)B);
    if (O <= 20)
        hipLaunchKernelGGL((k_lcl_bwd<20>), grid, dim3(LTPB), 0, (hipStream_t)stream, p, coef_a, coef_b, (const unsigned char *)mask, (int)O, (int)h, (int)w, sums, gloss, gx);
    else
        hipLaunchKernelGGL(k_lcl_bwd_any, grid, dim3(LTPB), 0, (hipStream_t)stream, p, coef_a, coef_b, (const unsigned char *)mask, (int)O, (int)h, (int)w, sums, gloss, gx);
    return check_launch("halo_local_consistent_bwd");
}
