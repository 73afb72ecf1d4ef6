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
__global__ void __launch_bounds__(LTPB) k_negative_fwd(const float *__restrict__ p, long long n, float thr, double *__restrict__ partials)
{
    double s = 0.0, c = 0.0;
    for (long long i = (long long)blockIdx.x * LTPB + threadIdx.x; i < n; i += (long long)gridDim.x * LTPB) {
        const float v = p[i];
        if (v < thr) { s += (double)(-det_logf((1.0f - v) + 1e-6f)); c += 1.0; }
    }
    block_sum2(s, c, partials + 2 * blockIdx.x);
}
// backward: gp = g * mask / ((1 - p + 1e-6) * count)
__global__ void __launch_bounds__(LTPB) k_negative_bwd(const float *__restrict__ p, long long n, float thr, const double *__restrict__ sums,
                                                       const float *__restrict__ gloss, float *__restrict__ gp)
{
    const long long i = (long long)blockIdx.x * LTPB + threadIdx.x;
    if (i >= n) return;
    const float v = p[i];
    const float scale = (float)((double)gloss[0] / sums[1]);
    gp[i] = v < thr ? scale / ((1.0f - v) + 1e-6f) : 0.0f;
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

// Per masked pixel: l = sum_c |p - mean| (l1) or sum_c p*log(p/(mean+1e-6)+1e-6) (kl), mean = 3x3 replicate-padded box
// mean; block partial sums of (l, 1).  When coef != nullptr also writes, per (pixel, class), a = dl/dp (direct) and
// b = dl/dmean for the backward pass (zeros outside the mask).
__global__ void __launch_bounds__(LTPB) k_lcl_fwd(const float *__restrict__ p, const long long *__restrict__ label, int O, int h, int w,
                                                  int kl, double *__restrict__ partials, float *__restrict__ ca, float *__restrict__ cb)
{
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const long long i = (long long)blockIdx.x * LTPB + threadIdx.x;
    double ls = 0.0, cnt = 0.0;
    if (i < hw) {
        const int y = (int)(i / w), x = (int)(i % w);
        const bool m = lcl_mask(label + (size_t)b * hw, h, w, y, x);
        const float *pb = p + (size_t)b * O * hw;
        float l = 0.0f;
        for (int c = 0; c < O; ++c) {
            float a = 0.0f, bb = 0.0f;
            if (m) {
                const float *pc = pb + (size_t)c * hw;
                float mean = 0.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        mean = __builtin_fmaf(pc[(size_t)clampi(y + dy, 0, h - 1) * w + clampi(x + dx, 0, w - 1)], 1.0f / 9.0f, mean);
                const float pv = pc[i];
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
            }
            if (ca) { ca[((size_t)b * O + c) * hw + i] = a; cb[((size_t)b * O + c) * hw + i] = bb; }
        }
        if (m) { ls = (double)l; cnt = 1.0; }
    }
    block_sum2(ls, cnt, partials + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// gx = softmax-backward( gp ),  gp_c(j) = a_c(j) + sum_{i in 3x3(j)} mult(i -> j)/9 * b_c(i),  scaled by g / count.
// mult(i -> j) = number of taps of i's replicate-padded window that land on j.
__global__ void __launch_bounds__(LTPB) k_lcl_bwd(const float *__restrict__ p, const float *__restrict__ ca, const float *__restrict__ cb,
                                                  int O, int h, int w, const double *__restrict__ sums, const float *__restrict__ gloss,
                                                  float *__restrict__ gx)
{
    const int b = blockIdx.y;
    const long long hw = (long long)h * w;
    const long long j = (long long)blockIdx.x * LTPB + threadIdx.x;
    if (j >= hw) return;
    const int y = (int)(j / w), x = (int)(j % w);
    const double cnt = sums[1];
    const float scale = cnt > 0.0 ? (float)((double)gloss[0] / cnt) : 0.0f;       // mean over an empty selection: zero gradient
    // multiplicities of the <= 9 neighbours
    float wgt[3][3];
    for (int ny = -1; ny <= 1; ++ny)
        for (int nx = -1; nx <= 1; ++nx) {
            const int iy = y + ny, ix = x + nx;
            int mult = 0;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        mult += (clampi(iy + dy, 0, h - 1) == y && clampi(ix + dx, 0, w - 1) == x) ? 1 : 0;
            wgt[ny + 1][nx + 1] = (float)mult * (1.0f / 9.0f);
        }
    const size_t base = (size_t)b * O * hw;
    float dot = 0.0f;
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < O; ++c) {
            const float *bc = cb + base + (size_t)c * hw;
            float gp = ca[base + (size_t)c * hw + j];
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
    return (size_t)(cdiv(n_pixels, LTPB) + 1) * 2 * sizeof(double) + 512;
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
    hipLaunchKernelGGL(k_negative_fwd, dim3(nblk), dim3(LTPB), 0, st, p, (long long)n, (float)threshold, part);
    hipLaunchKernelGGL(k_sum2_finalize, dim3(1), dim3(LTPB), 0, st, (const double *)part, nblk, sums);
    return check_launch("halo_negative_learning_fwd");
}

extern "C" int halo_negative_learning_bwd(const float *p, int64_t n, double threshold, const double *sums, const float *gloss,
                                          float *gp, void *stream)
{
    if (!p || !sums || !gloss || !gp || n <= 0) return fail(HALO_E_ARG, "halo_negative_learning_bwd: null/empty argument");
    hipLaunchKernelGGL(k_negative_bwd, dim3((unsigned)cdiv(n, LTPB)), dim3(LTPB), 0, (hipStream_t)stream, p, (long long)n, (float)threshold, sums, gloss, gp);
    return check_launch("halo_negative_learning_bwd");
}

// LocalConsistentLoss.forward: x (B,O,h,w) logits, label (B,h,w) i64 -> p (softmax, kept for backward), sums = {sum l, count};
// coef_a / coef_b (B,O,h,w) receive dl/dp and dl/dmean when not NULL.  kl: 0 = 'l1', 1 = 'kl'.
extern "C" int halo_local_consistent_fwd(const float *x, const int64_t *label, int64_t B, int64_t O, int64_t h, int64_t w, int kl,
                                         float *p, double *sums, float *coef_a, float *coef_b, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    if (!x || !label || !p || !sums || B <= 0 || O <= 0 || h <= 0 || w <= 0) return fail(HALO_E_ARG, "halo_local_consistent_fwd: null/empty argument");
    if ((coef_a == nullptr) != (coef_b == nullptr)) return fail(HALO_E_ARG, "halo_local_consistent_fwd: coef_a and coef_b go together");
    const long long hw = (long long)h * w;
    const int nb = (int)cdiv(hw, LTPB), nblk = nb * (int)B;
    if (!workspace || workspace_bytes < (size_t)nblk * 16 + 256) return fail(HALO_E_WORKSPACE, "halo_local_consistent_fwd: workspace too small");
    double *part = (double *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nb, (unsigned)B);
    hipLaunchKernelGGL(k_softmax_nchw, grid, dim3(LTPB), 0, st, x, (int)O, hw, p);
    hipLaunchKernelGGL(k_lcl_fwd, grid, dim3(LTPB), 0, st, (const float *)p, (const long long *)label, (int)O, (int)h, (int)w, kl, part, coef_a, coef_b);
    hipLaunchKernelGGL(k_sum2_finalize, dim3(1), dim3(LTPB), 0, st, (const double *)part, nblk, sums);
    return check_launch("halo_local_consistent_fwd");
}

extern "C" int halo_local_consistent_bwd(const float *p, const float *coef_a, const float *coef_b, int64_t B, int64_t O, int64_t h,
                                         int64_t w, const double *sums, const float *gloss, float *gx, void *stream)
{
    if (!p || !coef_a || !coef_b || !sums || !gloss || !gx || B <= 0 || O <= 0 || h <= 0 || w <= 0)
        return fail(HALO_E_ARG, "halo_local_consistent_bwd: null/empty argument");
    dim3 grid((unsigned)cdiv(h * w, LTPB), (unsigned)B);
    hipLaunchKernelGGL(k_lcl_bwd, grid, dim3(LTPB), 0, (hipStream_t)stream, p, coef_a, coef_b, (int)O, (int)h, (int)w, sums, gloss, gx);
    return check_launch("halo_local_consistent_bwd");
}
