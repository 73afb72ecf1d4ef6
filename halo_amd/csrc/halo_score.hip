// FloatingRegionScore.forward as HIP kernels for gfx950 (core/active/floating_region.py:129-217).
//
// Data flow for one batch of B images (everything stays in HBM between kernels, no host sync):
//
//   logit (B,O,H,W) f32 --k_logit_maps--> ent (B,H,W) f32 [, pred i16]
//   feat  (B,C,H,W) T   --k_feat_reduce-> imp_raw (B,H,W) T  + per-block min/max      <- HBM roofline kernel
//   [hyper]  imp_raw --k_quantize--> pred ;  pred --k_region_impurity--> imp_raw f32
//   ent --k_box_unc--> unc_raw f32 + per-block min/max
//   partial min/max --k_minmax_finalize--> stats[b] = {imp_min, imp_max, unc_min, unc_max}
//   imp_raw, unc_raw, stats [, active] --k_combine--> impurity, uncertainty, score
//
// k_feat_reduce is the only kernel that touches the C x H x W tensor: each lane owns 16 bytes
// of consecutive pixels (2 x f64 / 4 x f32), walks the C channel planes with 16-byte
// coalesced loads (a wave reads 1 KiB contiguous per plane) and keeps the running sum of
// squares in registers -- one fma chain per pixel in channel order, which is also bit for bit
// what ATen's CPU norm(dim=1) computes.  Algorithmic traffic: C*sizeof(T) bytes per pixel.
#include "halo_common.hpp"
#include "halo_devmath.hpp"
#include "halo_select_plan.hpp"      // SelHdr / order_key: the score-range record handed to the selector
#include <stdlib.h>

namespace halo {

constexpr int TPB = 256;
#ifndef HALO_FTPB                 // compile-time tuning hooks of k_feat_reduce (variant builds for A/B through HALO_LIB_PATH)
#define HALO_FTPB 128
#endif
#ifndef HALO_FEAT_WAVES
#define HALO_FEAT_WAVES 4
#endif
#ifndef HALO_FEAT_UNROLL
#define HALO_FEAT_UNROLL 8
#endif
constexpr int FTPB = HALO_FTPB;   // k_feat_reduce: 128-thread blocks stream ~3 % faster than 256 (tools/feat_microbench.hip)

// ---------------------------------------------------------------- reductions
// torch .min()/.max() propagate NaN: once a NaN is seen the result is NaN.
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

template <int NT = TPB>
__device__ __forceinline__ void block_minmax(double mn, double mx, double *out2)
{
    __shared__ double smn[NT / 64], smx[NT / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = nan_min(mn, __shfl_xor(mn, off));
        mx = nan_max(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smn[wave] = mn; smx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) { mn = nan_min(mn, smn[i]); mx = nan_max(mx, smx[i]); }
        out2[0] = mn;
        out2[1] = mx;
    }
}

// partials (B, nblk, 2) -> stats[b*4 + slot*2 + {0,1}]
__global__ void __launch_bounds__(TPB) k_minmax_finalize(const double *__restrict__ partials, int nblk,
                                                          double *__restrict__ stats, int slot)
{
    const int b = blockIdx.x;
    const double *p = partials + (size_t)b * nblk * 2;
    double mn = p[0], mx = p[1];
    for (int i = threadIdx.x; i < nblk; i += TPB) { mn = nan_min(mn, p[2 * i]); mx = nan_max(mx, p[2 * i + 1]); }
    block_minmax<TPB>(mn, mx, stats + b * 4 + slot * 2);
}

// both maps of normalize_map in one launch: blockIdx.y = slot (0: impurity partials, 1: uncertainty partials).  1024
// threads: the kernel is a dependent chain of loads on 2 B blocks, i.e. pure latency -- fewer trips per thread.
constexpr int FIN_TPB = 1024;
__global__ void __launch_bounds__(FIN_TPB) k_minmax_finalize2(const double *__restrict__ part0, int nblk0, const double *__restrict__ part1,
                                                               int nblk1, double *__restrict__ stats)
{
    const int b = blockIdx.x, slot = blockIdx.y;
    const int nblk = slot ? nblk1 : nblk0;
    const double2 *p = reinterpret_cast<const double2 *>((slot ? part1 : part0) + (size_t)b * nblk * 2);
    const double2 p0 = p[0];
    double mn = p0.x, mx = p0.y;
    for (int i = threadIdx.x; i < nblk; i += FIN_TPB) { const double2 q = p[i]; mn = nan_min(mn, q.x); mx = nan_max(mx, q.y); }
    block_minmax<FIN_TPB>(mn, mx, stats + b * 4 + slot * 2);
}

// ---------------------------------------------------------------- logits -> entropy / prediction
// Softmax, entropy and arg-max over the classes of a pixel (floating_region.py:72-76, 119, 152).  Two statements of the
// same arithmetic:
//
//  * the general one (px_general: rolled loops over the class planes in memory; softmax_general: registers): the full
//    det_expf / det_logf and a plain division p[c] / s, which hipcc expands into v_div_scale x2, v_rcp_f32, one Newton
//    step, q = n r, two fma corrections (the second one is v_div_fmas), v_div_fixup -- 11 instructions;
//  * the lean one (softmax_lean + finish_px<LEAN>), unrolled over the classes in registers, taken when every lane of the
//    wave has finite logits whose spread is at most 64.  Then x - max lies in [-64, 0], so det_expf needs neither its
//    clamp nor its NaN patch nor its two-step scaling (det_expf_core_small); every exp is >= 2^-93 and s is in [1, O_T], so v_div_scale would rescale
//    nothing (it does only when the numerator is below 2^-104 or the quotient would be denormal) and the division
//    sequence with the reciprocal computed once per pixel returns the identical, correctly rounded bits in 5
//    instructions per class; the probabilities lie in (0, 1], so p + 1e-6 is a positive normal number and the entropy
//    may use det_logf_core.  About 60 VALU operations per class and pixel instead of 95 plus 8 exec-mask branches.
//
// The NP pixels of a lane share the decision and therefore one basic block, in which the scheduler interleaves their
// independent chains (that fills the two-cycle hazard slots between v_cmp / v_cndmask pairs).  Built without SLP
// vectorisation (_build.py): packed v_pk_fma_f32 would save 10 % of the instructions and cost 40 VGPRs.

// The logarithm's 2 KB table: read from an LDS copy in the kernels below (HALO_LOGF_LDS=0 at build time: from device memory through
// the vector cache, the A/B twin -- same values either way).
#ifndef HALO_LOGF_LDS
#define HALO_LOGF_LDS 1
#endif

// torch.sum over the class axis in ATen's order (SumKernel.cpp, multi_row_sum; floating_region.py:72,119): terms enter acc[0]
// one by one; every 16 terms acc[0] is flushed into acc[1], every 256 acc[1] into acc[2], every 4096 acc[2] into acc[3]; the result
// is ((acc[0] + acc[1]) + acc[2]) + acc[3].  For 19 classes: (t16 + t17 + t18) + (t0 + ... + t15).  Stated here on the class
// index alone -- a flush happens when the NEXT term's index lies in another block, and the last block is never flushed: the
// same value, because the flushes ATen does beyond that either add +0 or commute with the final additions -- so that the window
// histograms below can skip their empty classes (an empty class contributes (-0) log(1e-6) = +0).  oracle/halo_oracle.c states
// ATen's loop literally (cascade_sum_f32).
struct ClassSum {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int prev = 0;
    __device__ __forceinline__ void add(int c, float t)
    {
        const int x = c ^ prev;
        if (x >> 4) {
            a1 = a1 + a0; a0 = 0.0f;
            if (x >> 8) {
                a2 = a2 + a1; a1 = 0.0f;
                if (x >> 12) { a3 = a3 + a2; a2 = 0.0f; }
            }
        }
        a0 = a0 + t;
        prev = c;
    }
    __device__ __forceinline__ float result() const { return ((a0 + a1) + a2) + a3; }
};

// General statement for one pixel whose class planes start at lp (stride hw).  is_prob: the planes already hold softmax
// probabilities (helper-method API).
__device__ __forceinline__ void px_general(const float *__restrict__ lp, int O, long long hw, int is_prob, int unc_type,
                                           int pur_type, long long g, float &ent, int &pred)
{
    float m = 0.0f, s = 1.0f;
    if (!is_prob) {
        m = lp[0];
#pragma unroll 1
        for (int c = 1; c < O; ++c) { float x = lp[(size_t)c * hw]; m = x > m ? x : m; }
        s = 0.0f;
#pragma unroll 1
        for (int c = 0; c < O; ++c) s = s + det_expf(lp[(size_t)c * hw] - m);
    }
    float best = 0.0f, pg = 0.0f;
    ClassSum acc;
    int am = 0;               // torch.argmax: first maximal class
#pragma unroll 1
    for (int c = 0; c < O; ++c) {
        const float p = is_prob ? lp[(size_t)c * hw] : det_expf(lp[(size_t)c * hw] - m) / s;
        if (c == 0 || p > best) { best = p; am = c; }
        if (c == (int)g) pg = p;
        acc.add(c, (-p) * det_logf(p + 1e-6f));
    }
    const float a = acc.result();
    if (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_PIXEL_ENTROPY) ent = a / (float)2.9444389791664403;   // math.log(19): hard-coded in the reference (:74-76)
    else if (unc_type == HALO_UNC_ORACLE_ACC) ent = 1.0f - (g == 255 ? best : pg);
    else ent = 0.0f;
    pred = pur_type == HALO_PUR_ORACLE_RIPU ? (g == 255 ? am : (int)g) : am;
}

__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Lean softmax of NP pixels in registers.  Returns false -- p untouched -- when some lane of the wave needs the general
// statement.  NaN logits hide from the two running extrema (a comparison with NaN is false), hence the sum t: it is NaN
// iff a NaN (or both infinities) is among the classes; infinite logits make lo - m infinite or NaN.
template <int O_T, int NP>
__device__ __forceinline__ bool softmax_lean(float (&p)[NP][O_T])
{
    float m[NP], lo[NP], t[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) m[j] = lo[j] = t[j] = p[j][0];
    // the two extrema through v_max3_f32 / v_min3_f32, two classes per instruction (a compare + select pair per class and
    // extremum before: four 4-cycle instructions per class, now one).  They differ from the `>` / `<` scan only where it does
    // not matter: a NaN operand is skipped (t is NaN then and the wave takes the general statement) and max(-0, +0) is +0
    // (x - m is then +-0 either way and exp(+-0) = 1).
#pragma unroll
    for (int c = 1; c + 1 < O_T; c += 2) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            m[j] = vmax3(m[j], p[j][c], p[j][c + 1]);
            lo[j] = vmin3(lo[j], p[j][c], p[j][c + 1]);
            t[j] = (t[j] + p[j][c]) + p[j][c + 1];
        }
    }
    if constexpr ((O_T - 1) % 2 == 1) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            m[j] = vmax3(m[j], p[j][O_T - 1], p[j][O_T - 1]);
            lo[j] = vmin3(lo[j], p[j][O_T - 1], p[j][O_T - 1]);
            t[j] = t[j] + p[j][O_T - 1];
        }
    }
    bool general = false;
#pragma unroll
    for (int j = 0; j < NP; ++j) general = general || !(lo[j] - m[j] >= -64.0f) || t[j] != t[j];
    if (__any(general)) return false;
    float s[NP], r[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) s[j] = 0.0f;
#pragma unroll
    for (int c = 0; c < O_T; ++c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) { p[j][c] = det_expf_core_small(p[j][c] - m[j]); s[j] = s[j] + p[j][c]; }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        r[j] = __builtin_amdgcn_rcpf(s[j]);
        r[j] = __builtin_fmaf(__builtin_fmaf(-s[j], r[j], 1.0f), r[j], r[j]);
    }
#pragma unroll
    for (int c = 0; c < O_T; ++c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float q = p[j][c] * r[j];
            q = __builtin_fmaf(__builtin_fmaf(-s[j], q, p[j][c]), r[j], q);
            p[j][c] = __builtin_fmaf(__builtin_fmaf(-s[j], q, p[j][c]), r[j], q);
        }
    }
    return true;
}

// General softmax in registers, for logits that exist nowhere in memory (the fused low-resolution path).
template <int O_T, int NP>
__device__ __forceinline__ void softmax_general(float (&p)[NP][O_T])
{
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        float m = p[j][0], s = 0.0f;
#pragma unroll
        for (int c = 1; c < O_T; ++c) m = p[j][c] > m ? p[j][c] : m;
#pragma unroll
        for (int c = 0; c < O_T; ++c) { p[j][c] = det_expf(p[j][c] - m); s = s + p[j][c]; }
#pragma unroll
        for (int c = 0; c < O_T; ++c) p[j][c] = p[j][c] / s;
    }
}

// From the probabilities of NP pixels: ent (per unc_type) and pred (per pur_type).  LEAN: p came from softmax_lean.
template <int O_T, int NP, bool LEAN>
__device__ __forceinline__ void finish_px(float (&p)[NP][O_T], int unc_type, int pur_type, const long long (&g)[NP],
                                          float (&ent)[NP], int (&pred)[NP], bool want_pred = true, const double (*ltab)[2] = logf_tab_)
{
    int am[NP];               // torch.argmax: first maximal class
    float best[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { am[j] = 0; best[j] = p[j][0]; }
    // (kernel-uniform: the arg-max is 5 instructions per class, and the radius purities with an entropy uncertainty -- the
    // reference's default configuration -- never read it)
    if (want_pred || unc_type == HALO_UNC_ORACLE_ACC) {
#pragma unroll
        for (int c = 1; c < O_T; ++c)
#pragma unroll
            for (int j = 0; j < NP; ++j) { const bool gtb = p[j][c] > best[j]; am[j] = gtb ? c : am[j]; best[j] = gtb ? p[j][c] : best[j]; }
    }
    if (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_PIXEL_ENTROPY) {
        // torch.sum(dim=0) in ATen's order (ClassSum above, unrolled): full blocks of 16 classes each from +0 into a1, the rest into a0
        static_assert(O_T <= 256, "class count of the unrolled statement");
        float a0[NP], a1[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) a0[j] = a1[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < O_T; ++c) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float q = p[j][c] + 1e-6f;
                a0[j] = a0[j] + (-p[j][c]) * (LEAN ? det_logf_core(q, ltab) : det_logf(q));
            }
            if ((c & 15) == 15 && c + 1 < O_T) {
#pragma unroll
                for (int j = 0; j < NP; ++j) { a1[j] = a1[j] + a0[j]; a0[j] = 0.0f; }
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) ent[j] = (a0[j] + a1[j]) / (float)2.9444389791664403;   // math.log(19): hard-coded in the reference (:74-76)
    } else if (unc_type == HALO_UNC_ORACLE_ACC) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int gi = g[j] == 255 ? am[j] : (int)g[j];
            float pg = 0.0f;
#pragma unroll
            for (int c = 0; c < O_T; ++c) pg = c == gi ? p[j][c] : pg;
            ent[j] = 1.0f - pg;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) ent[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) pred[j] = pur_type == HALO_PUR_ORACLE_RIPU ? (g[j] == 255 ? am[j] : (int)g[j]) : am[j];
}

template <int O_T, int VEC>
__global__ void __launch_bounds__(TPB) k_logit_maps(const float *__restrict__ logit, long long bstride,
                                                    const long long *__restrict__ gt, long long hw,
                                                    int unc_type, int pur_type, float *__restrict__ ent,
                                                    short *__restrict__ pred)
{
    const int b = blockIdx.y;
    const long long i0 = ((long long)blockIdx.x * TPB + threadIdx.x) * VEC;
#if HALO_LOGF_LDS
    __shared__ double s_ltab[128][2];
    stage_logf_table<TPB>(s_ltab);
    const double (*ltab)[2] = s_ltab;
#else
    const double (*ltab)[2] = logf_tab_;
#endif
    if (i0 >= hw) return;
    const float *lp = logit + (size_t)b * bstride + i0;
    const bool need_gt = unc_type == HALO_UNC_ORACLE_ACC || pur_type == HALO_PUR_ORACLE_RIPU;
    float v[VEC][O_T];
    if constexpr (VEC == 4) {
#pragma unroll
        for (int c = 0; c < O_T; ++c) {
            const float4 q = *reinterpret_cast<const float4 *>(lp + (size_t)c * hw);
            v[0][c] = q.x; v[1][c] = q.y; v[2][c] = q.z; v[3][c] = q.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < O_T; ++c) v[0][c] = lp[(size_t)c * hw];
    }
    float e[VEC];
    int pr[VEC];
    long long g[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[j] = need_gt ? gt[(size_t)b * hw + i0 + j] : 0;
    if (softmax_lean<O_T, VEC>(v)) {
        finish_px<O_T, VEC, true>(v, unc_type, pur_type, g, e, pr, pred != nullptr, ltab);
    } else {
#pragma unroll 1
        for (int j = 0; j < VEC; ++j) px_general(lp + j, O_T, hw, 0, unc_type, pur_type, g[j], e[j], pr[j]);
    }
    float *ep = ent + (size_t)b * hw + i0;
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(ep) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
        ep[0] = e[0];
    }
    if (pred) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) pred[(size_t)b * hw + i0 + j] = (short)pr[j];
    }
}

// Any class count: three passes over the O planes (coalesced scalar loads, L2-resident tile).
__global__ void __launch_bounds__(TPB) k_logit_maps_generic(const float *__restrict__ logit, long long bstride,
                                                            const long long *__restrict__ gt, int O, long long hw,
                                                            int unc_type, int pur_type, int is_prob,
                                                            float *__restrict__ ent, short *__restrict__ pred)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= hw) return;
    const bool need_gt = unc_type == HALO_UNC_ORACLE_ACC || pur_type == HALO_PUR_ORACLE_RIPU;
    const long long g = need_gt ? gt[(size_t)b * hw + i] : 0;
    float e;
    int pr;
    px_general(logit + (size_t)b * bstride + i, O, hw, is_prob, unc_type, pur_type, g, e, pr);
    ent[(size_t)b * hw + i] = e;
    if (pred) pred[(size_t)b * hw + i] = (short)pr;
}

// ---------------------------------------------------------------- features -> radius / norm (HBM roofline)
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T, int VEC> struct VecLoad;
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
template <> struct VecLoad<double, 2> {
    static __device__ __forceinline__ void ld(const double *p, double (&v)[2])
    {
        const d2_t q = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p));   // streamed once
        v[0] = q.x; v[1] = q.y;
    }
};
template <> struct VecLoad<float, 4> {
    static __device__ __forceinline__ void ld(const float *p, float (&v)[4])
    {
        const f4_t q = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(p));
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
};
template <typename T> struct VecLoad<T, 1> {
    static __device__ __forceinline__ void ld(const T *p, T (&v)[1]) { v[0] = *p; }
};

// MODE 0: poincare_distance_origin (pur 'radius' / 'hyper'), MODE 1: decoder_out.norm(dim=1) ('euc_norm')
// FO > 0: also compute the per-pixel entropy of the FO-class logits of the same pixels (what
// k_logit_maps does for unc 'entropy'/'pixel_entropy').  That work is ~1.2k VALU instructions per
// pixel against 1-2 KiB of HBM traffic, so it runs in the memory shadow of the other resident waves
// instead of costing a kernel of its own: with the entropy arithmetic compiled out the kernel takes
// the same time (6.10 vs 6.13 ms per 16 float32 images), with the logit loads compiled out it takes
// the time of the feature walk alone -- the fused entropy costs exactly its 76 bytes per pixel.
//
// At most 4 waves per SIMD (the attribute pads the register allocation to 104): the kernel needs 79 VGPRs and would
// run 6, which streams no faster and leaves 32 registers per SIMD lane -- the selection kernels of the other streams
// (k_sel_sweep and k_sel_scatter allocate 40) then wait for several of this kernel's blocks to retire on one CU before
// they can be placed: one image's selection beside the stream took 5.3 ms at 6 waves, 2.4-2.6 at 5 (but 5.6 for four
// images, every other run), 2.0-2.2 at 4 (2.6 for four images), with the same 11.4-11.5 ms per feature launch.
template <typename T, int VEC, int MODE, int UNROLL, int FO>
__global__ void __launch_bounds__(FTPB) __attribute__((amdgpu_waves_per_eu(1, HALO_FEAT_WAVES))) k_feat_reduce(const T *__restrict__ feat, long long bstride, int C,
                                                     long long hw, double ks, double rks, T *__restrict__ out,
                                                     double *__restrict__ partials, const float *__restrict__ logit,
                                                     long long lbstride, int unc_type, float *__restrict__ ent, unsigned xcd_g)
{
    const int b = blockIdx.y;
    // workgroup -> chunk of FTPB * VEC pixels.  Workgroups are dealt to the 8 XCDs round-robin; with the plain map the
    // eight XCDs share every 16 KiB of every plane.  XCD-contiguous in granules of xcd_g chunks instead: the j-th workgroup
    // of XCD x (j = id >> 3) takes chunk (j / G) * 8G + x * G + j % G, so an XCD's consecutive workgroups read adjacent
    // 2 KiB runs (tools/feat_microbench2.hip: +1.6-1.9 % for any granule >= 64 KiB; this kernel, interleaved on one
    // allocation, tools/ab_feat_map.py: -1.5 % f64, -3 % f32 per scoring call; the chunks past the last whole group of 8G
    // keep the plain map).
    unsigned bx = blockIdx.x;
    const unsigned xj = blockIdx.x >> 3;
    if (xcd_g != 0 && blockIdx.x < (gridDim.x / (8 * xcd_g)) * (8 * xcd_g))
        bx = (xj / xcd_g) * 8 * xcd_g + (blockIdx.x & 7) * xcd_g + xj % xcd_g;
    // every address below is (block-uniform base, kept in SGPRs) + (this lane's offset inside the block, one VGPR)
    const long long blk0 = (long long)bx * (FTPB * VEC);
    const unsigned lane0 = threadIdx.x * VEC;
    const long long i0 = blk0 + lane0;
    const bool live = i0 < hw;
#if HALO_LOGF_LDS
    __shared__ double s_ltab[FO > 0 ? 128 : 1][2];
    if constexpr (FO > 0) stage_logf_table<FTPB>(s_ltab);
    const double (*ltab)[2] = s_ltab;
#else
    const double (*ltab)[2] = logf_tab_;
#endif
    T acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = (T)0;
    // fused entropy of this block's pixels (FO > 0).  Odd blocks run it BEFORE the channel loop, even
    // blocks after it, so that at any moment about half of a CU's resident waves are in their VALU
    // phase and half are streaming -- identical blocks launched together would otherwise move
    // through the two phases in lockstep and the VALU work would not hide behind the loads.
    auto entropy_part = [&]() {
        if constexpr (FO > 0) {
            const float *lp = logit + (size_t)b * lbstride + blk0;
            float *eb = ent + (size_t)b * hw + blk0;
            // NP = two pixels at a time (four per lane: two parts), so that the class values of only two pixels are live
            // (38 VGPRs).  The entropy has its own pixel-to-lane map: part k covers the k-th run of FTPB * NP consecutive
            // pixels of the block, lane t the t-th pair of it, so that a wave's load of a class plane is one contiguous
            // 512-byte run (the walk's map -- 4 consecutive pixels per lane -- would use half of every line it touches).
            constexpr int NP = VEC >= 2 ? 2 : 1;
#pragma unroll
            for (int part = 0; part < VEC / NP; ++part) {
                const unsigned off = part * (FTPB * NP) + threadIdx.x * NP;
                if (blk0 + off < hw) {
                    float lv[NP][FO];
                    const float *plane = lp;               // advanced by scalar adds: the plane bases stay in SGPRs
#pragma unroll
                    for (int c2 = 0; c2 < FO; ++c2, plane += hw) {
                        if constexpr (NP == 2) {
                            const float2 q = *reinterpret_cast<const float2 *>(plane + off);
                            lv[0][c2] = q.x; lv[1][c2] = q.y;
                        } else {
                            lv[0][c2] = plane[off];
                        }
                    }
                    float e[NP];
                    int pr[NP];
                    long long gz[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) gz[j] = 0;
                    if (softmax_lean<FO, NP>(lv)) {
                        finish_px<FO, NP, true>(lv, unc_type, HALO_PUR_NONE, gz, e, pr, false, ltab);
                    } else {
#pragma unroll 1
                        for (int j = 0; j < NP; ++j) px_general(lp + off + j, FO, hw, 0, unc_type, HALO_PUR_NONE, 0, e[j], pr[j]);
                    }
                    if constexpr (NP == 2) *reinterpret_cast<float2 *>(eb + off) = make_float2(e[0], e[1]);
                    else eb[off] = e[0];
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the next part's loads (and registers) out of this one
            }
        }
    };
    // (with the XCD-contiguous map the order alternates per granule of an XCD's workgroups -- one resident generation of
    // 32 CUs x 8 blocks --, which measured 0.4 % better than alternating XCDs on the slow plateau and equal on the fast one)
    const bool ent_first = ((xcd_g != 0 ? xj / xcd_g : blockIdx.x) & 1) != 0;
    double mn = 0.0, mx = 0.0;
    // the channel walk, before or after the entropy (ent_first).  (Tried: a per-block pause point INSIDE the walk -- a hash of the
    // chunk index, in steps of UNROLL planes, the sums still one sequential chain -- so that the VALU phases of co-resident
    // blocks spread over the whole walk: 11.42 against 11.13 ms per launch in the bench, five interleaved runs each.)
    const T *p = feat + (size_t)b * bstride + blk0;          // advanced by scalar adds: the plane bases stay in SGPRs
    auto walk_to = [&](int c, const int c_end) {
        for (; c + UNROLL <= c_end; c += UNROLL) {
            T v[UNROLL][VEC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u, p += hw) VecLoad<T, VEC>::ld(p + lane0, v[u]);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = fma_t(v[u][j], v[u][j], acc[j]);
        }
        for (; c < c_end; ++c, p += hw) {
            T v[VEC];
            VecLoad<T, VEC>::ld(p + lane0, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fma_t(v[j], v[j], acc[j]);
        }
    };
    auto finish = [&]() {
        T r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if constexpr (MODE == 0) r[j] = dist0_from_ssq(acc[j], ks, rks);
            else if constexpr (sizeof(T) == 8) r[j] = __builtin_sqrt(acc[j]);
            else r[j] = __builtin_sqrtf(acc[j]);
        }
        T *op = (out + (size_t)b * hw + blk0) + lane0;
        if constexpr (VEC == 2) *reinterpret_cast<double2 *>(op) = make_double2(r[0], r[1]);
        else if constexpr (VEC == 4) *reinterpret_cast<float4 *>(op) = make_float4(r[0], r[1], r[2], r[3]);
        else op[0] = r[0];
        mn = mx = (double)r[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) { mn = nan_min(mn, (double)r[j]); mx = nan_max(mx, (double)r[j]); }
    };
    if constexpr (FO > 0) {
        if (live && !ent_first) { walk_to(0, C); finish(); }
        entropy_part();
        if (live && ent_first) { walk_to(0, C); finish(); }
    } else {
        if (live) { walk_to(0, C); finish(); }
    }
    // dead lanes of the last block take thread 0's value (always live) so they cannot disturb min/max
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<FTPB>(mn, mx, partials + ((size_t)b * gridDim.x + bx) * 2);
}

// ---------------------------------------------------------------- quantize_uncert_map (floating_region.py:94-110)
// r -> (r-min)/(max-min) -> 1-x -> (second min-max: exact no-op, min 0 / max 1) -> x*K-0.5 -> clamp -> round-half-even
template <typename T, typename TO>
__global__ void __launch_bounds__(TPB) k_quantize(const T *__restrict__ r, const double *__restrict__ stats,
                                                  long long hw, int K, TO *__restrict__ pred)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= hw) return;
    const T mn = (T)stats[b * 4 + 0], mx = (T)stats[b * 4 + 1];
    T x;
    if constexpr (sizeof(T) == 8) {
        x = (r[(size_t)b * hw + i] - mn) / (mx - mn);
        x = 1.0 - x;
        double p = x * (double)K - 0.5;
        const double lo = -0.5 + 1e-5, hi = (double)K - 0.5 - 1e-5;
        p = p < lo ? lo : p;
        p = p > hi ? hi : p;
        pred[(size_t)b * hw + i] = (TO)__builtin_rint(p);
    } else {
        const float den = (float)((double)mx - (double)mn);
        x = (r[(size_t)b * hw + i] - mn) / den;
        x = 1.0f - x;
        float p = x * (float)K - 0.5f;
        const float lo = (float)(-0.5 + 1e-5), hi = (float)((double)K - 0.5 - 1e-5);
        p = p < lo ? lo : p;
        p = p > hi ? hi : p;
        // float32 and K > 128: K - 0.5 - 1e-5 rounds to K - 0.5, which rounds half-even to K -- one past the last bin, where the
        // reference's F.one_hot raises.  The last bin takes it here (and in the oracle).
        const float q = __builtin_rintf(p);
        pred[(size_t)b * hw + i] = (TO)(q > (float)(K - 1) ? (float)(K - 1) : q);
    }
}

// ---------------------------------------------------------------- compute_region_impurity (floating_region.py:112-121)
// Window class histogram -> sum_c -d*log(d+1e-6) / log(K), classes visited in ascending order and added in
// torch.sum's order over the one-hot channel axis (ClassSum; empty classes contribute exactly +0).
// No one-hot tensor: the <= k*k window labels are re-scanned once per distinct class.
// nn.Conv2d(padding_mode=...) of the two box filters (floating_region.py:49,63): where a window tap falls outside the image.
// 'zeros' contributes nothing (and does not count in the purity window); the other three modes read an image pixel
// (torch pads the input first: F.pad(mode) then an unpadded convolution), so every tap counts.  Valid for pad < n
// ('reflect') / pad <= n ('circular'), which the host checks as torch does.
__device__ __forceinline__ int pad_index(int t, int n, int mode)
{
    if (t >= 0 && t < n) return t;
    if (mode == HALO_PAD_ZEROS) return -1;
    if (mode == HALO_PAD_REPLICATE) return t < 0 ? 0 : n - 1;
    if (mode == HALO_PAD_REFLECT) return t < 0 ? -t : 2 * (n - 1) - t;
    return t < 0 ? t + n : t - n;                       // circular
}

template <typename TL>
__global__ void __launch_bounds__(TPB) k_region_impurity(const TL *__restrict__ pred, int H, int W, int k,
                                                         float logK, float *__restrict__ imp, float *__restrict__ count, int pad)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= hw) return;
    const int y = (int)(i / W), x = (int)(i % W), r = k / 2;
    // zero padding: the in-image part of the window; any other mode: the whole window, taps mapped into the image
    const int y0 = pad ? y - r : (y - r < 0 ? 0 : y - r), y1 = pad ? y + r : (y + r >= H ? H - 1 : y + r);
    const int x0 = pad ? x - r : (x - r < 0 ? 0 : x - r), x1 = pad ? x + r : (x + r >= W ? W - 1 : x + r);
    const TL *pp = pred + (size_t)b * hw;
    const float cnt = (float)((y1 - y0 + 1) * (x1 - x0 + 1));
    ClassSum acc;
    int cur = -1;
    while (true) {
        int nxt = 0x7fffffff, n = 0;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                const int v = (int)pp[(size_t)pad_index(yy, H, pad) * W + pad_index(xx, W, pad)];
                if (v > cur) {
                    if (v < nxt) { nxt = v; n = 1; }
                    else if (v == nxt) ++n;
                }
            }
        if (n == 0) break;
        const float d = (float)n / cnt;
        acc.add(nxt, (-d) * det_logf(d + 1e-6f));
        cur = nxt;
    }
    imp[(size_t)b * hw + i] = acc.result() / logK;
    if (count) count[(size_t)b * hw + i] = cnt;
}

// 3 x 3 window (the only purity window the reference's drivers use: RADIUS_K = 1, and always for 'hyper'): the
// sliding-window class histogram on an LDS label tile.  A 256-thread block covers 64 x 16 pixels; the 66 x 18 label
// tile (halo included, out-of-image = -1) is staged once, each lane takes the 3 x 6 labels of its 4 consecutive
// pixels into registers and walks the distinct classes of each 3 x 3 window in ascending order -- the same sums in
// the same order as the generic kernel above, with no global re-scan per class.
constexpr int RI_TW = 64, RI_TH = 16;

// partials (optional): per-block min / max of the impurity written, for normalize_map -- saves a pass over the map
template <typename TL>
__global__ void __launch_bounds__(TPB) k_region_impurity3(const TL *__restrict__ pred, int H, int W, float logK,
                                                          float *__restrict__ imp, float *__restrict__ count,
                                                          double *__restrict__ partials)
{
    __shared__ int lab[RI_TH + 2][RI_TW + 2 + 2];                  // +2: row stride 68 words
    // A class with n of the window's cnt pixels contributes (-d) * log(d + 1e-6), d = n / cnt: at most 9 x 9 distinct values,
    // formed once per block by the same expression (same bits) instead of a division and a logarithm per class and pixel.
    __shared__ float term[10][10];
    const int b = blockIdx.z, tid = threadIdx.x;
    if (tid < 100) {
        const int c = tid / 10, n = tid % 10;
        float tv = 0.0f;
        if (c > 0 && n > 0 && n <= c) { const float d = (float)n / (float)c; tv = (-d) * det_logf(d + 1e-6f); }
        term[c][n] = tv;
    }
    const int X0 = blockIdx.x * RI_TW, Y0 = blockIdx.y * RI_TH;
    const long long hw = (long long)H * W;
    const TL *pp = pred + (size_t)b * hw;
    for (int e = tid; e < (RI_TH + 2) * (RI_TW + 2); e += TPB) {
        const int ty = e / (RI_TW + 2), tx = e % (RI_TW + 2);
        const int y = Y0 + ty - 1, x = X0 + tx - 1;
        lab[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? (int)pp[(size_t)y * W + x] : -1;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = (tid & 15) * 4;                  // 16 lanes x 4 pixels per tile row
    const int y = Y0 + ly, xb = X0 + lx;
    const bool live = y < H && xb < W;
    double mn = 0.0, mx = 0.0;
    int v[3][6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 6; ++dx) v[dy][dx] = lab[ly + dy][lx + dx];
    const int ny = (y > 0 ? 1 : 0) + 1 + (y < H - 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xb + j;
        if (!live || x >= W) break;
        const int cnti = ny * ((x > 0 ? 1 : 0) + 1 + (x < W - 1 ? 1 : 0));
        const float cnt = (float)cnti;
        const float *tc = term[cnti];
        ClassSum acc;
        int cur = -1;
        while (true) {
            int nxt = 0x7fffffff;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) { const int q = v[dy][j + dx]; nxt = (q > cur && q < nxt) ? q : nxt; }
            if (nxt == 0x7fffffff) break;
            int n = 0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) n += v[dy][j + dx] == nxt ? 1 : 0;
            acc.add(nxt, tc[n]);
            cur = nxt;
        }
        const float res = acc.result() / logK;
        imp[(size_t)b * hw + (size_t)y * W + x] = res;
        if (count) count[(size_t)b * hw + (size_t)y * W + x] = cnt;
        if (j == 0) mn = mx = (double)res;
        else { mn = nan_min(mn, (double)res); mx = nan_max(mx, (double)res); }
    }
    if (!partials) return;                                          // kernel argument: uniform
    __shared__ double seed[2];
    if (tid == 0) { seed[0] = mn; seed[1] = mx; }                   // thread 0 (pixel X0, Y0) is always live
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + (((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2);
}

// Returns the number of min/max partials per image written to `partials` (0: none -- the caller runs k_minmax_f32).
template <typename TL>
static int launch_region_impurity(const TL *pred, int64_t B, int64_t H, int64_t W, int k, float logK, float *imp, float *count,
                                  hipStream_t st, double *partials = nullptr, int pad = HALO_PAD_ZEROS)
{
    if (pad == HALO_PAD_ZEROS && k == 3 && cdiv(H, RI_TH) <= 65535 && B <= 65535 && !getenv("HALO_IMPURITY_GENERIC")) {     // A/B switch
        const dim3 grid((unsigned)cdiv(W, RI_TW), (unsigned)cdiv(H, RI_TH), (unsigned)B);
        hipLaunchKernelGGL((k_region_impurity3<TL>), grid, dim3(TPB), 0, st, pred, (int)H, (int)W, logK, imp, count, partials);
        return partials ? (int)(grid.x * grid.y) : 0;
    }
    hipLaunchKernelGGL((k_region_impurity<TL>), dim3((unsigned)cdiv(H * W, TPB), (unsigned)B), dim3(TPB), 0, st, pred, (int)H, (int)W, k,
                       logK, imp, count, pad);
    return 0;
}

// ---------------------------------------------------------------- entropy_conv + /count (floating_region.py:42-51,90,204)
// k x k all-ones box SUM with zero padding, taps added in row-major order starting from +0.
// count = in-bounds size of the pk x pk purity window for ripu / oracle_ripu / hyper, else 1.
__global__ void __launch_bounds__(TPB) k_box_unc(const float *__restrict__ ent, int H, int W, int k, int do_box,
                                                 int pk, float *__restrict__ unc, double *__restrict__ partials, int pad)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool live = i < hw;
    double mn = 0.0, mx = 0.0;
    if (live) {
        const int y = (int)(i / W), x = (int)(i % W);
        const float *ep = ent + (size_t)b * hw;
        float a;
        if (do_box) {
            const int r = k / 2;
            a = 0.0f;
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    const int yy = pad_index(y + dy, H, pad), xx = pad_index(x + dx, W, pad);
                    const float v = (yy >= 0 && xx >= 0) ? ep[(size_t)yy * W + xx] : 0.0f;
                    a = a + v;
                }
        } else {
            a = ep[i];
        }
        float cnt = 1.0f;
        if (pk > 0 && pad) {
            cnt = (float)(pk * pk);                    // padded taps are image pixels: the purity window is always full
        } else if (pk > 0) {
            const int r = pk / 2;
            const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
            const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
            cnt = (float)((y1 - y0 + 1) * (x1 - x0 + 1));
        }
        a = a / cnt;
        unc[(size_t)b * hw + i] = a;
        mn = mx = (double)a;
    }
    if (!partials) return;
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

// 3 x 3 fast path (the only window the reference's drivers use: RADIUS_K = 1, build.py:83-88):
// 4 consecutive pixels per lane, three float4 row loads; the two edge neighbours of a row come from the adjacent lanes'
// registers (full-wave DPP shifts) and are loaded only by the first / last lane of a wave.  Same row-major tap order
// (out-of-image taps add +0) as the generic kernel.
__device__ __forceinline__ float lane_prev(float v)      // value of lane - 1 (lane 0: unspecified)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v)      // value of lane + 1 (lane 63: unspecified)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// a / count, count = in-bounds size of the pk x pk purity window (pk = 0: count 1, and a / 1 is a, bit for bit)
__device__ __forceinline__ float box3_over_count(float a, int H, int W, int y, int xx, int pk)
{
    if (pk <= 0) return a;
    const int rr = pk / 2;
    const int y0 = y - rr < 0 ? 0 : y - rr, y1 = y + rr >= H ? H - 1 : y + rr;
    const int x0 = xx - rr < 0 ? 0 : xx - rr, x1 = xx + rr >= W ? W - 1 : xx + rr;
    return a / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
}

// Box sums / count of the 4 pixels (y, x..x+3) of image plane `ep`; x % 4 == 0, W % 4 == 0.  Every lane of the wave
// calls it (dead lanes with live = false: they only take part in the lane shifts).
__device__ __forceinline__ void box3_row4(const float *__restrict__ ep, int H, int W, int y, int x, bool live, int pk, float (&o)[4])
{
    const int lane = threadIdx.x & 63;
    float r[3][6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        const bool in = live && yy >= 0 && yy < H;
        const float *row = ep + (size_t)(in ? yy : 0) * W;
        float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (in) q = *reinterpret_cast<const float4 *>(row + x);
        // the neighbours' registers hold the same row (a lane's left neighbour in the wave is the 4 pixels before it)
        float left = lane_prev(q.w), right = lane_next(q.x);
        if (lane == 0) left = (in && x > 0) ? row[x - 1] : 0.0f;
        if (lane == 63) right = (in && x + 4 < W) ? row[x + 4] : 0.0f;
        r[dy][0] = (in && x > 0) ? left : 0.0f;
        r[dy][1] = q.x; r[dy][2] = q.y; r[dy][3] = q.z; r[dy][4] = q.w;
        r[dy][5] = (in && x + 4 < W) ? right : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a = a + r[dy][j + dx];
        o[j] = box3_over_count(a, H, W, y, x + j, pk);
    }
}

// unc may be NULL: then only the per-block min / max are produced (k_combine_box3 recomputes the sums it normalises)
__global__ void __launch_bounds__(TPB) k_box3_unc(const float *__restrict__ ent, int H, int W, int pk,
                                                  float *__restrict__ unc, double *__restrict__ partials)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i0 = ((long long)blockIdx.x * TPB + threadIdx.x) * 4;
    const bool live = i0 < hw;
    double mn = 0.0, mx = 0.0;
    const int y = live ? (int)(i0 / W) : 0, x = live ? (int)(i0 % W) : 0;          // W % 4 == 0: the 4 pixels share a row
    float o[4];
    box3_row4(ent + (size_t)b * hw, H, W, y, x, live, pk, o);
    if (live) {
        if (unc) *reinterpret_cast<float4 *>(unc + (size_t)b * hw + i0) = make_float4(o[0], o[1], o[2], o[3]);
        mn = mx = (double)o[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) { mn = nan_min(mn, (double)o[j]); mx = nan_max(mx, (double)o[j]); }
    }
    if (!partials) return;
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

// The min / max pass of the fused tail: the same sums as box3_row4 (same taps, same order), nothing stored.  A wave owns a
// strip of 256 columns x BM_RPT rows and slides a three-row window down it: BM_RPT + 2 row loads for BM_RPT rows of sums
// (k_box3_unc: 3 per row) and ONE reduction per BM_RPT * 4 pixels of a lane, in float32 (exact; widened at the end).
constexpr int BM_RPT = 4, BM_TW = 256, BM_TH = BM_RPT * (TPB / 64);

__device__ __forceinline__ float nan_minf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ float nan_maxf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

__global__ void __launch_bounds__(TPB) k_box3_minmax(const float *__restrict__ ent, int H, int W, int pk, double *__restrict__ partials,
                                                     unsigned *__restrict__ hist_zero)
{
    const int b = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (hist_zero) {        // the coarse histogram k_combine_box3 fills behind this kernel: cleared here (NB1 counters per image)
        const unsigned nb = gridDim.x * gridDim.y;
        for (unsigned l = blockIdx.y * gridDim.x + blockIdx.x; l < NB1 / TPB; l += nb) hist_zero[(size_t)b * NB1 + l * TPB + threadIdx.x] = 0u;
    }
    const int x = blockIdx.x * BM_TW + lane * 4, y0 = blockIdx.y * BM_TH + wv * BM_RPT;
    const float *ep = ent + (size_t)b * H * W;
    const bool xin = x < W;
    auto load_row = [&](int yy, float (&r)[6]) {
        const bool in = xin && yy >= 0 && yy < H;
        const float *row = ep + (size_t)(in ? yy : 0) * W;
        float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (in) q = *reinterpret_cast<const float4 *>(row + x);
        float left = lane_prev(q.w), right = lane_next(q.x);
        if (lane == 0) left = (in && x > 0) ? row[x - 1] : 0.0f;
        if (lane == 63) right = (in && x + 4 < W) ? row[x + 4] : 0.0f;
        r[0] = (in && x > 0) ? left : 0.0f;
        r[1] = q.x; r[2] = q.y; r[3] = q.z; r[4] = q.w;
        r[5] = (in && x + 4 < W) ? right : 0.0f;
    };
    float r0[6], r1[6], r2[6];
    load_row(y0 - 1, r0);
    load_row(y0, r1);
    float mn = 0.0f, mx = 0.0f;
    bool have = false;
#pragma unroll
    for (int k = 0; k < BM_RPT; ++k) {
        const int y = y0 + k;
        load_row(y + 1, r2);                 // every lane of the wave, also past the image: the lane shifts are wave-wide
        if (xin && y < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.0f;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = a + r0[j + dx];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = a + r1[j + dx];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = a + r2[j + dx];
                const float o = box3_over_count(a, H, W, y, x + j, pk);
                if (!have) { mn = mx = o; have = true; }
                else { mn = nan_minf(mn, o); mx = nan_maxf(mx, o); }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) { r0[q] = r1[q]; r1[q] = r2[q]; }
    }
    // lanes without a pixel take the block's first pixel (lane 0 of wave 0 always has one)
    __shared__ float seedf[2];
    __shared__ float smn[TPB / 64], smx[TPB / 64];
    if (threadIdx.x == 0) { seedf[0] = mn; seedf[1] = mx; }
    __syncthreads();
    if (!have) { mn = seedf[0]; mx = seedf[1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = nan_minf(mn, __shfl_xor(mn, off));
        mx = nan_maxf(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) { smn[wv] = mn; smx[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < TPB / 64; ++i) { mn = nan_minf(mn, smn[i]); mx = nan_maxf(mx, smx[i]); }
        double *out2 = partials + (((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2;
        out2[0] = (double)mn;
        out2[1] = (double)mx;
    }
}

// per-block min/max of an existing f32 map (impurity of the histogram branches)
__global__ void __launch_bounds__(TPB) k_minmax_f32(const float *__restrict__ x, long long hw,
                                                    double *__restrict__ partials)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool live = i < hw;
    double mn = 0.0, mx = 0.0;
    if (live) mn = mx = (double)x[(size_t)b * hw + i];
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

__global__ void __launch_bounds__(TPB) k_fill_f32(float *__restrict__ x, long long n, float v)
{
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) x[i] = v;
}

// ---------------------------------------------------------------- normalize_map + product (+ prior-pick mask)
// (floating_region.py:22-23,206-210; core/active/build.py:146)
template <typename TI>
__global__ void __launch_bounds__(TPB) k_combine(const TI *__restrict__ imp_raw, const float *__restrict__ unc_raw,
                                                 const double *__restrict__ stats, const unsigned char *__restrict__ active,
                                                 long long hw, int normalize, TI *__restrict__ score,
                                                 TI *__restrict__ imp_out, float *__restrict__ unc_out)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= hw) return;
    const size_t o = (size_t)b * hw + i;
    TI im = imp_raw[o];
    float un = unc_raw[o];
    if (normalize) {
        const float umn = (float)stats[b * 4 + 2], umx = (float)stats[b * 4 + 3];
        un = (un - umn) / (float)((double)umx - (double)umn);
        const TI imn = (TI)stats[b * 4 + 0], imx = (TI)stats[b * 4 + 1];
        if constexpr (sizeof(TI) == 8) im = (im - imn) / (imx - imn);
        else im = (im - imn) / (float)((double)imx - (double)imn);
    }
    TI s = im * (TI)un;
    if (active && active[o]) {
        if constexpr (sizeof(TI) == 8) s = __longlong_as_double(0xfff0000000000000ll);
        else s = __uint_as_float(0xff800000u);
    }
    score[o] = s;
    if (imp_out) imp_out[o] = im;
    if (unc_out) unc_out[o] = un;
}


// k_combine with the 3 x 3 box sum of the uncertainty recomputed from the entropy map (box3_row4: the same operations
// that produced the min / max, so the same bits) instead of read from a stored copy: one kernel and one 4-byte map
// less in the step's tail.  4 pixels per lane, 16-byte loads and stores.
// rng (optional, normalised maps only): the image's score-range record for the selector, written by ONE lane per image.  A
// normalised score is the product of two values in [0, 1] -- (x - min) / (max - min) of finite maps -- or -inf where masked, so
// [0, 1] bounds it without a pass over the map; the only other possibility is a NaN map, and that is a property of the
// min / max themselves (a constant or NaN-carrying map makes max - min zero or NaN): no per-pixel reporting is needed.
// CB_ROWS consecutive rows per workgroup (round 4): the per-row work is what it was -- box3_row4 per row, so the sums are the bits
// k_box3_minmax reduced -- but a workgroup now lives long enough to privatise the selector's coarse histogram of the score it
// writes (NB1 counters in LDS, flushed once: with one row per workgroup the flush would be an atomic per pixel).  For a normalised
// map the score is a product of two values in [0, 1] (or -inf where masked), so the bins are those of the range record this kernel
// writes: lo = 0, scale = NB1 -- exactly what k_sel_hist1 would count from the stored map (same coarse_bin, same exclusions).
constexpr int CB_ROWS = 16;        // at most; fewer when the launch would otherwise not fill the chip (one image at a time)

// NTS: the impurity / uncertainty maps are stored non-temporally (nothing on the device reads them again) and the raw impurity is
// loaded non-temporally (read exactly once); the score map -- the selector's input -- and the entropy rows (shared with the
// neighbouring workgroups) keep the default policy.
template <typename TI, bool NTS>
__global__ void __launch_bounds__(TPB) k_combine_box3(const TI *__restrict__ imp_raw, const float *__restrict__ ent,
                                                      const double *__restrict__ stats, const unsigned char *__restrict__ active,
                                                      int H, int W, int pk, int normalize, TI *__restrict__ score,
                                                      TI *__restrict__ imp_out, float *__restrict__ unc_out, SelHdr *__restrict__ rng,
                                                      unsigned *__restrict__ hist, int rows)
{
    // grid (row segments of TPB * 4 pixels, groups of CB_ROWS rows, images): no index division (a 64-bit i / W was a fifth of the
    // kernel's instructions); lanes past the end of a row idle
    __shared__ unsigned hl[NB1];
    const int b = blockIdx.z;
    const long long hw = (long long)H * W;
    const int x = (blockIdx.x * TPB + threadIdx.x) * 4;
    const bool live = x < W;
    float umn = 0.0f, uden = 1.0f;
    TI imn = (TI)0, iden = (TI)1;
    bool fin = false;
    if (normalize) {
        umn = (float)stats[b * 4 + 2];
        const float umx = (float)stats[b * 4 + 3];
        uden = (float)((double)umx - (double)umn);
        imn = (TI)stats[b * 4 + 0];
        const TI imx = (TI)stats[b * 4 + 1];
        if constexpr (sizeof(TI) == 8) iden = imx - imn;
        else iden = (float)((double)imx - (double)imn);
        // zero, NaN or infinite range: the map holds NaN (0/0, inf/inf); a finite positive range with finite extrema
        // leaves every quotient in [0, 1]
        fin = uden > 0.0f && uden < __builtin_inff() && iden > (TI)0 && iden < (TI)__builtin_inf();
        if (rng && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            SelHdr h;
            memset(&h, 0, sizeof(h));
            h.kmin_inv = ~order_key(0.0);
            h.kmax = order_key(1.0);
            h.nvalid = 1u;                     // "something may be pickable": a fully masked map simply yields no candidates
            h.flags = fin ? (hist ? (unsigned)SEL_F_HIST : 0u) : (unsigned)SEL_F_BAD;
            rng[b] = h;
        }
    }
    const bool count = hist != nullptr && fin;               // kernel-uniform per image
    if (count) {
        for (int j = threadIdx.x; j < NB1; j += TPB) hl[j] = 0u;
        __syncthreads();
    }
    ValRange vr;
    vr.lo = 0.0; vr.scale = (double)NB1; vr.ok = true;       // = sel_range of the record above
    const int ybeg = blockIdx.y * rows, yend = ybeg + rows < H ? ybeg + rows : H;
    // the three-row window slides down the workgroup's rows: CB_ROWS + 2 row loads instead of 3 per row, the same taps added in the
    // same (row-major) order as box3_row4 / k_box3_minmax
    const float *ep = ent + (size_t)b * hw;
    const int lane = threadIdx.x & 63;
    // A row of taps in two steps, so that the loads of row y + 2 are in flight while row y is being combined (round 5: with load and
    // lane shifts in one step every trip of the row loop began with a full memory round trip -- 16 of them per workgroup):
    //   row_issue : the lane's float4 and, for the wave's first / last lane, the two neighbours beyond the wave's span
    //   row_finish: the edge taps from the adjacent lanes' registers (full-wave DPP shifts), out-of-image taps = +0
    struct RowRaw { float4 q; float el, er; bool in; };
    auto row_issue = [&](int yy) {
        RowRaw t;
        t.in = live && yy >= 0 && yy < H;
        const float *row = ep + (size_t)(t.in ? yy : 0) * W;
        t.q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        t.el = 0.0f; t.er = 0.0f;
        if (t.in) t.q = *reinterpret_cast<const float4 *>(row + x);
        if (lane == 0 && t.in && x > 0) t.el = row[x - 1];
        if (lane == 63 && t.in && x + 4 < W) t.er = row[x + 4];
        return t;
    };
    auto row_finish = [&](const RowRaw &t, float (&r)[6]) {
        float left = lane_prev(t.q.w), right = lane_next(t.q.x);
        if (lane == 0) left = t.el;
        if (lane == 63) right = t.er;
        r[0] = (t.in && x > 0) ? left : 0.0f;
        r[1] = t.q.x; r[2] = t.q.y; r[3] = t.q.z; r[4] = t.q.w;
        r[5] = (t.in && x + 4 < W) ? right : 0.0f;
    };
    typedef double cb_d2 __attribute__((ext_vector_type(2)));
    typedef float cb_f4 __attribute__((ext_vector_type(4)));
    struct PixRaw { cb_d2 d0, d1; cb_f4 f; unsigned am; };
    auto pix_issue = [&](int yy) {          // impurity and prior-pick mask of row yy (clamped: the values of a row past the group are not used)
        PixRaw t;
        t.d0 = (cb_d2){0.0, 0.0}; t.d1 = t.d0; t.f = (cb_f4){0.f, 0.f, 0.f, 0.f}; t.am = 0u;
        if (!live || yy >= yend) return t;
        const size_t o = (size_t)b * hw + (size_t)yy * W + x;
        if constexpr (sizeof(TI) == 8) {
            if constexpr (NTS) {
                t.d0 = __builtin_nontemporal_load(reinterpret_cast<const cb_d2 *>(imp_raw + o));
                t.d1 = __builtin_nontemporal_load(reinterpret_cast<const cb_d2 *>(imp_raw + o + 2));
            } else {
                t.d0 = *reinterpret_cast<const cb_d2 *>(imp_raw + o);
                t.d1 = *reinterpret_cast<const cb_d2 *>(imp_raw + o + 2);
            }
        } else {
            if constexpr (NTS) t.f = __builtin_nontemporal_load(reinterpret_cast<const cb_f4 *>(imp_raw + o));
            else t.f = *reinterpret_cast<const cb_f4 *>(imp_raw + o);
        }
        if (active) t.am = *reinterpret_cast<const unsigned *>(active + o);
        return t;
    };
    float r0[6], r1[6], r2[6];
    {
        const RowRaw a0 = row_issue(ybeg - 1), a1 = row_issue(ybeg);
        row_finish(a0, r0);
        row_finish(a1, r1);
    }
    RowRaw nrow = row_issue(ybeg + 1);                            // in flight across a trip: the taps of row y + 1 ...
    PixRaw npix = pix_issue(ybeg);                                // ... and the impurity / mask of row y
    for (int y = ybeg; y < yend; ++y) {
        row_finish(nrow, r2);                                     // every lane of the wave: the lane shifts are wave-wide
        const PixRaw cpix = npix;
        nrow = row_issue(y + 2);                                  // requested now, used in the next trip
        npix = pix_issue(y + 1);
        float un[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.0f;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a = a + r0[j + dx];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a = a + r1[j + dx];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a = a + r2[j + dx];
            un[j] = box3_over_count(a, H, W, y, x + j, pk);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) { r0[q] = r1[q]; r1[q] = r2[q]; }
        if (!live) continue;
        const size_t o = (size_t)b * hw + (size_t)y * W + x;
        TI im[4];
        if constexpr (sizeof(TI) == 8) { im[0] = cpix.d0.x; im[1] = cpix.d0.y; im[2] = cpix.d1.x; im[3] = cpix.d1.y; }
        else { im[0] = cpix.f.x; im[1] = cpix.f.y; im[2] = cpix.f.z; im[3] = cpix.f.w; }
        if (normalize) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { un[j] = (un[j] - umn) / uden; im[j] = (im[j] - imn) / iden; }
        }
        const unsigned am = cpix.am;
        TI sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j] = im[j] * (TI)un[j];
            if ((am >> (8 * j)) & 0xffu) {
                if constexpr (sizeof(TI) == 8) sc[j] = __longlong_as_double(0xfff0000000000000ll);
                else sc[j] = __uint_as_float(0xff800000u);
            }
        }
        if (count) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = (double)sc[j];
                const unsigned long long k = order_key(v);
                if (k < KEY_POS_INF && k != KEY_NEG_INF) {
                    double t;
                    atomicAdd(&hl[coarse_bin(v, vr, t)], 1u);
                }
            }
        }
        if constexpr (sizeof(TI) == 8) {
            *reinterpret_cast<double2 *>(score + o) = make_double2(sc[0], sc[1]);
            *reinterpret_cast<double2 *>(score + o + 2) = make_double2(sc[2], sc[3]);
            if (imp_out) {
                if constexpr (NTS) {
                    __builtin_nontemporal_store((cb_d2){im[0], im[1]}, reinterpret_cast<cb_d2 *>(imp_out + o));
                    __builtin_nontemporal_store((cb_d2){im[2], im[3]}, reinterpret_cast<cb_d2 *>(imp_out + o + 2));
                } else {
                    *reinterpret_cast<double2 *>(imp_out + o) = make_double2(im[0], im[1]);
                    *reinterpret_cast<double2 *>(imp_out + o + 2) = make_double2(im[2], im[3]);
                }
            }
        } else {
            *reinterpret_cast<float4 *>(score + o) = make_float4(sc[0], sc[1], sc[2], sc[3]);
            if (imp_out) {
                if constexpr (NTS) __builtin_nontemporal_store((cb_f4){im[0], im[1], im[2], im[3]}, reinterpret_cast<cb_f4 *>(imp_out + o));
                else *reinterpret_cast<float4 *>(imp_out + o) = make_float4(im[0], im[1], im[2], im[3]);
            }
        }
        if (unc_out) {
            if constexpr (NTS) __builtin_nontemporal_store((cb_f4){un[0], un[1], un[2], un[3]}, reinterpret_cast<cb_f4 *>(unc_out + o));
            else *reinterpret_cast<float4 *>(unc_out + o) = make_float4(un[0], un[1], un[2], un[3]);
        }
    }
    if (count) {
        __syncthreads();
        unsigned *g = hist + (size_t)b * NB1;
        for (int j = threadIdx.x; j < NB1; j += TPB)
            if (hl[j]) atomicAdd(&g[j], hl[j]);
    }
}


// ================================================================ low-resolution sources
// RegionSelection upsamples the head's low-resolution outputs to label size before scoring
// (core/active/build.py:122-135): the float64 embedding becomes a C x H x W tensor (4.3 GB at C=256)
// that is written once and read once.  These kernels consume the LOW-RES tensors directly and
// interpolate on the fly with exactly the arithmetic of k_bilinear (halo_hyperbolic.hip), so their
// outputs are bit-identical to "upsample, then score" while the full-resolution tensor never exists.

// bilinear taps of one output coordinate, weights in the tensor's dtype (align_corners=True)
template <typename T> struct Taps { int i0, i1; T l0, l1; };
template <typename T>
__device__ __forceinline__ Taps<T> make_taps(int o, T scale, int in_size)
{
    Taps<T> t;
    const T f = scale * (T)o;
    int i0 = (int)f;
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    t.i0 = i0;
    t.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    t.l1 = f - (T)i0;
    t.l0 = (T)1 - t.l1;
    return t;
}
constexpr int LR_STAGE_IT = 3, LR_STAGE_G = 4;      // window elements per lane with precomputed offsets (<= 192 taps), channels per group
constexpr int LR_TW = 64, LR_TH = 16, LR_PPT = 4;   // 64 x 16 output pixels per 256-thread block; a wave owns 4 consecutive rows x 64 columns

// bilinear weights of a lane's LR_PPT vertically adjacent output pixels (one column: lx shared, one ly pair per pixel).
// The four taps are combined by bilerp (halo_devmath.hpp): columns first, rows second -- ATen's order.
template <typename T, int NPX = LR_PPT> struct LrW { T lx0, lx1, ly0[NPX], ly1[NPX]; };


// One channel chunk of the window for a lane's LR_PPT vertically adjacent pixels, when their upper tap rows are
// R + {0, PAT bit 0, PAT bit 1, PAT bit 2} (wave-uniform, compile-time): the lane reads the 2 or 3 source rows once per
// channel -- 4 or 6 LDS words instead of the 16 of four independent pixels -- and every pixel picks its rows by register
// name.  The LDS pipe (128 bytes per clock and CU, shared by the four SIMDs) is what bounds this kernel, not the 5 float64
// operations per pixel and channel.  Same bilerp per pixel as the generic loop (the column interpolation of a row is one expression, shared), so the same bits.
// The six LDS words of a lane (rows R..R+2, columns q and q + 1).  For 8-byte elements they are read by hand-written
// ds_read_b64 (2 LDS cycles each, 256 bytes per clock): the compiler fuses each (q, q + 1) pair into one ds_read2_b64,
// which moves the same 16 bytes per lane in 16 LDS cycles and leaves the kernel bound by the LDS pipe at 2.4x its float64
// arithmetic (the pair is only 8-byte aligned, so ds_read_b128 is not an option).  The wait is part of the statement, so
// the values are architecturally ready when the compiler sees them; the latency hides behind the SIMD's other waves.
template <typename T, int PAT> struct LrRows {
    T v[3][2];
    __device__ __forceinline__ void load(const T *__restrict__ tp, int stride)
    {
        if constexpr (sizeof(T) == 8) {
            typedef __attribute__((address_space(3))) const T *lds_ptr;
            const unsigned a0 = (unsigned)(uintptr_t)(lds_ptr)tp, a1 = a0 + (unsigned)stride * 8u;
            if constexpr (PAT != 0) {
                const unsigned a2 = a1 + (unsigned)stride * 8u;
                asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %7\n\tds_read_b64 %3, %7 offset:8\n\t"
                             "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8 offset:8\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]), "=&v"(v[2][1])
                             : "v"(a0), "v"(a1), "v"(a2)
                             : "memory");
            } else {
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1])
                             : "v"(a0), "v"(a1)
                             : "memory");
            }
        } else {
            v[0][0] = tp[0]; v[0][1] = tp[1];
            v[1][0] = tp[stride]; v[1][1] = tp[stride + 1];
            if constexpr (PAT != 0) { v[2][0] = tp[2 * stride]; v[2][1] = tp[2 * stride + 1]; }
        }
    }
    __device__ __forceinline__ void accumulate(const LrW<T> &wt, T (&acc)[LR_PPT]) const
    {
        // the column interpolation of each source row once, shared by the lane's pixels that read the row (2 or 3 rows for 4
        // pixels: 16-18 float64 operations per channel where four independent pixels take 28)
        T t[3];
        t[0] = col_lerp(wt.lx0, wt.lx1, v[0][0], v[0][1]);
        t[1] = col_lerp(wt.lx0, wt.lx1, v[1][0], v[1][1]);
        if constexpr (PAT != 0) t[2] = col_lerp(wt.lx0, wt.lx1, v[2][0], v[2][1]);
#pragma unroll
        for (int j = 0; j < LR_PPT; ++j) {
            const int a = j == 0 ? 0 : (PAT >> (j - 1)) & 1;      // 0 or 1; a + 1 == 2 only when PAT != 0
            const T x = col_lerp(wt.ly0[j], wt.ly1[j], t[a], t[a + 1]);
            acc[j] = fma_t(x, x, acc[j]);
        }
    }
};
template <typename T, int PAT>
__device__ __forceinline__ void lr_rows_chunk(const T *__restrict__ tp, int cc, int plane, int stride, const LrW<T> &wt, T (&acc)[LR_PPT])
{
#pragma unroll 1
    for (int ch = 0; ch < cc; ++ch, tp += plane) {
        LrRows<T, PAT> A;
        A.load(tp, stride);
        A.accumulate(wt, acc);
    }
}

// Features: per output pixel  sum_c (interp_c)^2  with the low-res taps of a channel chunk staged in LDS.
// The staged window carries one extra row and column whose source coordinates are clamped to the grid, so that "the next
// row / column" is always readable and, at the bottom / right edge of the image, holds the clamped tap the reference uses
// there (i1 == i0).
template <typename T, int MODE>
__global__ void __launch_bounds__(TPB) k_feat_reduce_lr(const T *__restrict__ feat, long long bstride, int C, int h, int w,
                                                        int H, int W, T sh, T sw, int max_rows, int max_cols, int CC,
                                                        double ks, double rks, T *__restrict__ out,
                                                        double *__restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_smem[];
    T *tile = reinterpret_cast<T *>(lr_smem);                    // [CC][max_rows][max_cols]
    // Tile <-> block: workgroups go to the 8 XCDs round-robin in launch order, each XCD with its own L2.  Tiles that are
    // neighbours along x share the source lines at their common border (a tile's 19-element window rows straddle two
    // 128-byte lines), so in launch order every line would be fetched by two XCDs.  Give the blocks that share an XCD
    // (id % 8) one contiguous eighth of the row-major tile list instead -- bijective for any tile count.
    const unsigned ntx = gridDim.x, nty = gridDim.y, ntiles = ntx * nty * gridDim.z;
    const unsigned lin = blockIdx.x + ntx * (blockIdx.y + nty * blockIdx.z);
    const unsigned xq = ntiles / 8, xr = ntiles % 8, xk = lin % 8;
    const unsigned tile_id = (xk < xr ? xk * (xq + 1) : xr * (xq + 1) + (xk - xr) * xq) + lin / 8;
    const int bx = (int)(tile_id % ntx), by = (int)((tile_id / ntx) % nty);
    const int b = (int)(tile_id / (ntx * nty));
    const int X0 = bx * LR_TW, Y0 = by * LR_TH;
    const int lx = threadIdx.x & (LR_TW - 1), ly = threadIdx.x / LR_TW;      // ly = wave in 0..3
    const int x = X0 + lx;
    // tap window of this block in the low-res grid
    const int ylast = (Y0 + LR_TH - 1 < H ? Y0 + LR_TH - 1 : H - 1), xlast = (X0 + LR_TW - 1 < W ? X0 + LR_TW - 1 : W - 1);
    const int ty_lo = make_taps<T>(Y0, sh, h).i0, ty_hi = make_taps<T>(ylast, sh, h).i1;
    const int tx_lo = make_taps<T>(X0, sw, w).i0, tx_hi = make_taps<T>(xlast, sw, w).i1;
    const int rows = ty_hi - ty_lo + 2, cols = tx_hi - tx_lo + 2;            // with the extra row / column; <= max_rows / max_cols
    const bool xin = x < W;
    const Taps<T> tx = make_taps<T>(xin ? x : W - 1, sw, w);
    T acc[LR_PPT];
    LrW<T> wt;
    wt.lx0 = tx.l0; wt.lx1 = tx.l1;
    int o00[LR_PPT], o10[LR_PPT];
    bool live[LR_PPT];
    int a0 = 0, pat = 0;
    bool regular = true;         // wave-uniform: the rows' upper taps are R + {0, 1}
#pragma unroll
    for (int j = 0; j < LR_PPT; ++j) {
        const int y = Y0 + ly * LR_PPT + j;
        live[j] = xin && y < H;
        const Taps<T> ty = make_taps<T>(y < H ? y : H - 1, sh, h);
        wt.ly0[j] = ty.l0; wt.ly1[j] = ty.l1;
        o00[j] = (ty.i0 - ty_lo) * max_cols + (tx.i0 - tx_lo);
        o10[j] = (ty.i1 - ty_lo) * max_cols + (tx.i0 - tx_lo);
        acc[j] = (T)0;
        if (j == 0) a0 = ty.i0;
        const int d = ty.i0 - a0;                                  // non-decreasing in j
        regular = regular && (d == 0 || d == 1);
        if (j > 0) pat |= (d & 1) << (j - 1);
    }
    const int dx1 = tx.i1 - tx.i0;
    const T *fb = feat + (size_t)b * bstride;
    const int plane = max_rows * max_cols;
    // staging offsets of this lane inside one channel plane of the window
    int st_src[LR_STAGE_IT], st_dst[LR_STAGE_IT];
    bool st_ok[LR_STAGE_IT];
#pragma unroll
    for (int it = 0; it < LR_STAGE_IT; ++it) {
        const int e = (threadIdx.x & 63) + 64 * it;
        const int r = e / cols, q = e % cols;
        const int sr = ty_lo + r < h - 1 ? ty_lo + r : h - 1, sq = tx_lo + q < w - 1 ? tx_lo + q : w - 1;
        st_ok[it] = e < rows * cols;
        st_src[it] = sr * w + sq;
        st_dst[it] = r * max_cols + q;
    }
    const int upat = __builtin_amdgcn_readfirstlane(regular ? pat : -1);       // the rows of a wave are shared by its lanes
    // Staging: wave k of the block owns channels k, k + 4, ... of a chunk (CC <= 4 * LR_STAGE_G, so at most LR_STAGE_G
    // of them), its lanes walk the window with precomputed offsets; all loads are issued before the first LDS write so
    // that the chunk costs one memory round trip.  (Issuing the loads of chunk n + 1 before the arithmetic of chunk n
    // gained 7 %, less than the 24 registers it holds are worth to the LDS double buffer of lr_rows_chunk.)
    const int wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += CC) {
        const int cc = C - c0 < CC ? C - c0 : CC;
        __syncthreads();                                          // previous chunk fully consumed
        {
            T pre[LR_STAGE_G][LR_STAGE_IT];
#pragma unroll
            for (int gI = 0; gI < LR_STAGE_G; ++gI) {
                const int ch = wv + gI * (TPB / 64);
                const T *src = fb + (size_t)(c0 + (ch < cc ? ch : 0)) * h * w;
#pragma unroll
                for (int it = 0; it < LR_STAGE_IT; ++it) pre[gI][it] = (st_ok[it] && ch < cc) ? src[st_src[it]] : (T)0;
            }
#pragma unroll
            for (int gI = 0; gI < LR_STAGE_G; ++gI) {
                const int ch = wv + gI * (TPB / 64);
                if (ch < cc) {
                    T *dst = tile + ch * plane;
#pragma unroll
                    for (int it = 0; it < LR_STAGE_IT; ++it)
                        if (st_ok[it]) dst[st_dst[it]] = pre[gI][it];
                    for (int e = (threadIdx.x & 63) + 64 * LR_STAGE_IT; e < rows * cols; e += 64) {   // very large windows
                        const int r = e / cols, q = e % cols;
                        const int sr = ty_lo + r < h - 1 ? ty_lo + r : h - 1, sq = tx_lo + q < w - 1 ? tx_lo + q : w - 1;
                        dst[r * max_cols + q] = fb[(size_t)(c0 + ch) * h * w + (size_t)sr * w + sq];
                    }
                }
            }
        }
        __syncthreads();
        const T *t0 = tile + o00[0];
        if (upat == 0) lr_rows_chunk<T, 0>(t0, cc, plane, max_cols, wt, acc);
        else if (upat == 4) lr_rows_chunk<T, 4>(t0, cc, plane, max_cols, wt, acc);
        else if (upat == 6) lr_rows_chunk<T, 6>(t0, cc, plane, max_cols, wt, acc);
        else if (upat == 7) lr_rows_chunk<T, 7>(t0, cc, plane, max_cols, wt, acc);
        else {
            // any other geometry (factors below 3, rows clamped past the image): four independent pixels
#pragma unroll 1
            for (int ch = 0; ch < cc; ++ch) {
                const T *tp = tile + ch * plane;
#pragma unroll
                for (int j = 0; j < LR_PPT; ++j) {
                    const T v = bilerp<T>(tp[o00[j]], tp[o00[j] + dx1], tp[o10[j]], tp[o10[j] + dx1], wt.lx0, wt.lx1, wt.ly0[j], wt.ly1[j]);
                    acc[j] = fma_t(v, v, acc[j]);
                }
            }
        }
    }
    double mn = 0.0, mx = 0.0;
    bool have = false;
#pragma unroll
    for (int j = 0; j < LR_PPT; ++j) {
        if (!live[j]) continue;
        const int y = Y0 + ly * LR_PPT + j;
        T r;
        if constexpr (MODE == 0) r = dist0_from_ssq(acc[j], ks, rks);
        else if constexpr (sizeof(T) == 8) r = __builtin_sqrt(acc[j]);
        else r = __builtin_sqrtf(acc[j]);
        out[(size_t)b * H * W + (size_t)y * W + x] = r;
        if (!have) { mn = mx = (double)r; have = true; }
        else { mn = nan_min(mn, (double)r); mx = nan_max(mx, (double)r); }
    }
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }         // thread 0 (pixel X0,Y0) is always live
    __syncthreads();
    if (!have) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + (size_t)tile_id * 2);
}

// ---- k_feat_reduce_lr with the window staged by LDS-DMA into a double buffer (float64, even source width).
// The kernel above is bound by its staging, not by its arithmetic (round 2: 2.4 ms without the arithmetic, 1.9 ms without the
// staging, 3.05 ms complete, per 16 images): every 16-channel chunk is one dependent global round trip between two barriers,
// through registers and ds_write_b64.  Here the window of chunk n + 1 travels global -> LDS by global_load_lds_dwordx4 (16 bytes
// per lane, no registers, no LDS write instructions) WHILE chunk n is being interpolated: two LDS images of DMA_UNITS 16-byte
// units, DMA_PER_WAVE instructions per wave and chunk, a counted s_waitcnt vmcnt(DMA_PER_WAVE) + raw s_barrier so that the
// prefetch stays in flight across the barriers.
//   * LDS image of a chunk: [channel][row][pair] dense, lane-linear in the order the DMA writes it (the destination of an
//     LDS-DMA is wave-uniform base + lane x 16); the window starts at an EVEN source column so that every pair is a 16-byte
//     aligned load (w even);
//   * lanes past the end of a chunk re-read a valid address into the image's padding (no exec masking: every wave issues
//     exactly DMA_PER_WAVE instructions per chunk, which is what the counted wait counts);
//   * a pair that would start past the row's last column (only the clamped extra column of a right-edge tile) is loaded from
//     (w - 2, w - 1) and its first half overwritten with the second after the data has landed: both halves then hold the
//     clamped tap v[w - 1] the arithmetic expects there.
// Same bilerp / fma chains per pixel, so the same bits as k_feat_reduce_lr and as upsample-then-reduce.
constexpr int DMA_PER_WAVE = 4, DMA_UNITS = DMA_PER_WAVE * TPB;      // 1024 units = 16 KiB per image

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(TPB) k_feat_reduce_lr_dma(const double *__restrict__ feat, long long bstride, int C, int h, int w,
                                                            int H, int W, double sh, double sw, int CC, double ks, double rks,
                                                            double *__restrict__ out, double *__restrict__ partials)
{
    typedef double T;
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_smem[];
    T *img = reinterpret_cast<T *>(lr_smem);                     // two images of DMA_UNITS * 2 doubles
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_bytes)lr_smem;
    // tile <-> block: XCD-contiguous eighths of the row-major tile list (see k_feat_reduce_lr)
    const unsigned ntx = gridDim.x, nty = gridDim.y, ntiles = ntx * nty * gridDim.z;
    const unsigned lin = blockIdx.x + ntx * (blockIdx.y + nty * blockIdx.z);
    const unsigned xq = ntiles / 8, xr = ntiles % 8, xk = lin % 8;
    const unsigned tile_id = (xk < xr ? xk * (xq + 1) : xr * (xq + 1) + (xk - xr) * xq) + lin / 8;
    const int bx = (int)(tile_id % ntx), by = (int)((tile_id / ntx) % nty);
    const int b = (int)(tile_id / (ntx * nty));
    const int X0 = bx * LR_TW, Y0 = by * LR_TH;
    const int lx = threadIdx.x & (LR_TW - 1), ly = threadIdx.x / LR_TW;      // ly = wave in 0..3
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = X0 + lx;
    const int ylast = (Y0 + LR_TH - 1 < H ? Y0 + LR_TH - 1 : H - 1), xlast = (X0 + LR_TW - 1 < W ? X0 + LR_TW - 1 : W - 1);
    const int ty_lo = make_taps<T>(Y0, sh, h).i0, ty_hi = make_taps<T>(ylast, sh, h).i1;
    const int tx_lo = make_taps<T>(X0, sw, w).i0 & ~1, tx_hi = make_taps<T>(xlast, sw, w).i1;       // even start column
    const int rows = ty_hi - ty_lo + 2;                          // with the clamped extra row
    const int U = (tx_hi - tx_lo + 2 + 1) >> 1;                  // pairs per row, covering the clamped extra column
    const int per = rows * U, stride = 2 * U, plane = 2 * per;   // units per channel; row / channel strides in doubles
    const bool xin = x < W;
    const Taps<T> tx = make_taps<T>(xin ? x : W - 1, sw, w);
    T acc[LR_PPT];
    LrW<T> wt;
    wt.lx0 = tx.l0; wt.lx1 = tx.l1;
    int o00[LR_PPT], o10[LR_PPT];
    bool live[LR_PPT];
    int a0 = 0, pat = 0;
    bool regular = true;
#pragma unroll
    for (int j = 0; j < LR_PPT; ++j) {
        const int y = Y0 + ly * LR_PPT + j;
        live[j] = xin && y < H;
        const Taps<T> ty = make_taps<T>(y < H ? y : H - 1, sh, h);
        wt.ly0[j] = ty.l0; wt.ly1[j] = ty.l1;
        o00[j] = (ty.i0 - ty_lo) * stride + (tx.i0 - tx_lo);
        o10[j] = (ty.i1 - ty_lo) * stride + (tx.i0 - tx_lo);
        acc[j] = (T)0;
        if (j == 0) a0 = ty.i0;
        const int d = ty.i0 - a0;
        regular = regular && (d == 0 || d == 1);
        if (j > 0) pat |= (d & 1) << (j - 1);
    }
    const int dx1 = tx.i1 - tx.i0;
    const int upat = __builtin_amdgcn_readfirstlane(regular ? pat : -1);
    // this lane's DMA_PER_WAVE units of a chunk: source offset (doubles, from the chunk's first plane) and channel inside the chunk
    const long long hwl = (long long)h * w;
    unsigned soff[DMA_PER_WAVE], sch[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int u = (i * (TPB / 64) + wv) * 64 + lane;
        const int ch = u / per, rem = u - ch * per, r = rem / U, pp = rem - r * U;
        const int sr = ty_lo + r < h - 1 ? ty_lo + r : h - 1;
        int q0 = tx_lo + 2 * pp;
        q0 = q0 + 1 > w - 1 ? w - 2 : q0;                        // a pair past the last column: (w - 2, w - 1), patched after it lands
        sch[i] = (unsigned)ch;
        soff[i] = (unsigned)(sr * w + q0);
    }
    const T *fb = feat + (size_t)b * bstride;
    auto issue = [&](int c0, int buf) {
        const int cc = C - c0 < CC ? C - c0 : CC;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const unsigned ch = sch[i] < (unsigned)cc ? sch[i] : (unsigned)(cc - 1);       // past the chunk: any valid plane (padding)
            const T *src = fb + (size_t)(c0 + ch) * hwl + soff[i];
            const unsigned dst = lds0 + (unsigned)buf * (DMA_UNITS * 16) + (unsigned)((i * (TPB / 64) + wv) * 64) * 16u;
            glds16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
        }
    };
    const bool patch = tx_lo + 2 * U > w;                        // block-uniform: the image holds a pair past the last source column
    const int nchunks = (C + CC - 1) / CC;
    issue(0, 0);
    for (int n = 0; n < nchunks; ++n) {
        const int c0 = n * CC, cc = C - c0 < CC ? C - c0 : CC;
        if (n + 1 < nchunks) {
            issue(c0 + CC, (n + 1) & 1);                          // its image was last read two iterations ago, behind a barrier
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");      // all but the newest chunk of THIS wave have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                             // ... and every other wave's part of chunk n too
        T *tile = img + (size_t)(n & 1) * (DMA_UNITS * 2);
        if (patch) {
            const int pu = (w - tx_lo) >> 1;                      // the pair that starts at column w
            for (int e = threadIdx.x; e < cc * rows; e += TPB) {
                T *q = tile + (size_t)(e / rows) * plane + (size_t)(e % rows) * stride + 2 * pu;
                q[0] = q[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const T *t0 = tile + o00[0];
        if (upat == 0) lr_rows_chunk<T, 0>(t0, cc, plane, stride, wt, acc);
        else if (upat == 4) lr_rows_chunk<T, 4>(t0, cc, plane, stride, wt, acc);
        else if (upat == 6) lr_rows_chunk<T, 6>(t0, cc, plane, stride, wt, acc);
        else if (upat == 7) lr_rows_chunk<T, 7>(t0, cc, plane, stride, wt, acc);
        else {
#pragma unroll 1
            for (int ch = 0; ch < cc; ++ch) {
                const T *tp = tile + (size_t)ch * plane;
#pragma unroll
                for (int j = 0; j < LR_PPT; ++j) {
                    const T v = bilerp<T>(tp[o00[j]], tp[o00[j] + dx1], tp[o10[j]], tp[o10[j] + dx1], wt.lx0, wt.lx1, wt.ly0[j], wt.ly1[j]);
                    acc[j] = fma_t(v, v, acc[j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of the image are retired ...
        __builtin_amdgcn_s_barrier();                             // ... and everybody's, before the next iteration restages it
    }
    double mn = 0.0, mx = 0.0;
    bool have = false;
#pragma unroll
    for (int j = 0; j < LR_PPT; ++j) {
        if (!live[j]) continue;
        const int y = Y0 + ly * LR_PPT + j;
        T r;
        if constexpr (MODE == 0) r = dist0_from_ssq(acc[j], ks, rks);
        else r = __builtin_sqrt(acc[j]);
        out[(size_t)b * H * W + (size_t)y * W + x] = r;
        if (!have) { mn = mx = (double)r; have = true; }
        else { mn = nan_min(mn, (double)r); mx = nan_max(mx, (double)r); }
    }
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }         // thread 0 (pixel X0,Y0) is always live
    __syncthreads();
    if (!have) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + (size_t)tile_id * 2);
}

// ---- k_feat_reduce_lr_dma with a COMPILE-TIME image geometry: ROWS x UU pairs per channel, CCF = DMA_UNITS / (ROWS * UU)
// channels per image.  Every LDS address of the interpolation loop is then one base register plus an immediate offset
// (channel * plane + row * stride + {0, 8}): no address arithmetic in the loop (3 of 26 instructions per channel before), the
// loop is fully unrolled over the chunk, and the six row words of channel c + 1 are requested BEFORE channel c is interpolated
// (two register sets, counted lgkmcnt) -- a wave no longer parks on its own LDS latency once per channel.
template <int PAT, int OFF, int STRIDE_B> struct LrRowsImm {
    double v[3][2];
    // request the row words of the channel whose plane starts OFF bytes behind `base` (no wait)
    __device__ __forceinline__ void issue(unsigned base)
    {
        if constexpr (PAT != 0) {
            asm volatile("ds_read_b64 %0, %6 offset:%7\n\tds_read_b64 %1, %6 offset:%8\n\tds_read_b64 %2, %6 offset:%9\n\t"
                         "ds_read_b64 %3, %6 offset:%10\n\tds_read_b64 %4, %6 offset:%11\n\tds_read_b64 %5, %6 offset:%12"
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]), "=&v"(v[2][1])
                         : "v"(base), "n"(OFF), "n"(OFF + 8), "n"(OFF + STRIDE_B), "n"(OFF + STRIDE_B + 8), "n"(OFF + 2 * STRIDE_B),
                           "n"(OFF + 2 * STRIDE_B + 8)
                         : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\t"
                         "ds_read_b64 %3, %4 offset:%8"
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1])
                         : "v"(base), "n"(OFF), "n"(OFF + 8), "n"(OFF + STRIDE_B), "n"(OFF + STRIDE_B + 8)
                         : "memory");
        }
    }
    // the values become readable: N = LDS reads issued AFTER this set's (they may stay in flight)
    template <int N> __device__ __forceinline__ void wait()
    {
        if constexpr (PAT != 0)
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]) : "n"(N) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]) : "n"(N) : "memory");
    }
    __device__ __forceinline__ void accumulate(const LrW<double> &wt, double (&acc)[LR_PPT]) const
    {
        // the column interpolation of each source row once, shared by the lane's pixels that read the row (2 or 3 rows for 4
        // pixels: 16-18 float64 operations per channel where four independent pixels take 28)
        double t[3];
        t[0] = col_lerp(wt.lx0, wt.lx1, v[0][0], v[0][1]);
        t[1] = col_lerp(wt.lx0, wt.lx1, v[1][0], v[1][1]);
        if constexpr (PAT != 0) t[2] = col_lerp(wt.lx0, wt.lx1, v[2][0], v[2][1]);
#pragma unroll
        for (int j = 0; j < LR_PPT; ++j) {
            const int a = j == 0 ? 0 : (PAT >> (j - 1)) & 1;
            const double x = col_lerp(wt.ly0[j], wt.ly1[j], t[a], t[a + 1]);
            acc[j] = fma_t(x, x, acc[j]);
        }
    }
};

template <int PAT, int ROWS, int UU, int CH, int CCF, bool FULL>
__device__ __forceinline__ void lr_imm_steps(unsigned base, int cc, LrRowsImm<PAT, (CH % CCF) * ROWS * UU * 16, UU * 16> &cur,
                                             const LrW<double> &wt, double (&acc)[LR_PPT])
{
    constexpr int NREAD = PAT != 0 ? 6 : 4;
    if constexpr (CH + 1 < CCF) {
        LrRowsImm<PAT, ((CH + 1) % CCF) * ROWS * UU * 16, UU * 16> nxt;
        nxt.issue(base);                                   // channel CH + 1 on its way ...
        cur.template wait<NREAD>();                        // ... channel CH has landed
        if (FULL || CH < cc) cur.accumulate(wt, acc);
        lr_imm_steps<PAT, ROWS, UU, CH + 1, CCF, FULL>(base, cc, nxt, wt, acc);
    } else {
        cur.template wait<0>();
        if (FULL || CH < cc) cur.accumulate(wt, acc);
    }
}

// FULL: the chunk holds CCF channels (every chunk but possibly the last): no per-channel test of the channel count
template <int PAT, int ROWS, int UU, int CCF>
__device__ __forceinline__ void lr_imm_chunk(unsigned base, int cc, const LrW<double> &wt, double (&acc)[LR_PPT])
{
    // The first channel's reads are issued INSIDE each branch: between a set's issue and its wait there must be no control-flow
    // merge.  The compiler takes an asm output for a value that exists when the statement ends; issued before the branch, the two
    // arms wanted the set in different registers and the copies (v_mov_b64 of registers whose LDS data was still in flight) were
    // placed in front of the wait: channel 0 of every partial chunk read stale registers, now and then
    // (test_lowres_exact_mode_staging_variants_agree_bitwise found it one day after the branch went in).
    if (cc == CCF) {
        LrRowsImm<PAT, 0, UU * 16> first;
        first.issue(base);
        lr_imm_steps<PAT, ROWS, UU, 0, CCF, true>(base, cc, first, wt, acc);
    } else {
        LrRowsImm<PAT, 0, UU * 16> first;
        first.issue(base);
        lr_imm_steps<PAT, ROWS, UU, 0, CCF, false>(base, cc, first, wt, acc);
    }
}

// ---- the same with NPX (8) vertically adjacent pixels per lane.  CODE holds the row of every pixel's upper tap relative to the
// lane's first row, 2 bits per pixel (wave-uniform, compile-time, non-decreasing): the lane reads NR = last row + 2 source rows
// (2 words each) per channel and interpolates every row's columns once -- for 8 pixels over 3-4 rows that is 30-32 float64
// instructions and 6-8 LDS words per channel where two 4-pixel groups take 34-36 and 10-12; the A/B of round 4 (arithmetic removed
// / LDS reads removed) found this kernel bound by float64 issue first and the LDS pipe second.  Same expressions per pixel: same bits.
template <int NPX, unsigned CODE, int OFF, int STRIDE_B> struct LrRowsImmN {
    static constexpr int row(int j) { return (int)((CODE >> (2 * j)) & 3u); }
    static constexpr int NR = row(NPX - 1) + 2;
    double v[NR][2];
    template <int R> __device__ __forceinline__ void issue_rows(unsigned base)
    {
        if constexpr (R < NR) {
            asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                         : "=&v"(v[R][0]), "=&v"(v[R][1]) : "v"(base), "n"(OFF + R * STRIDE_B), "n"(OFF + R * STRIDE_B + 8) : "memory");
            issue_rows<R + 1>(base);
        }
    }
    __device__ __forceinline__ void issue(unsigned base) { issue_rows<0>(base); }
    // the values become readable: N = LDS reads issued AFTER this set's (they may stay in flight); the empty statements pin every
    // use of a value behind the wait
    template <int N> __device__ __forceinline__ void wait()
    {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
        for (int r = 0; r < NR; ++r) { asm volatile("" : "+v"(v[r][0])); asm volatile("" : "+v"(v[r][1])); }
    }
    __device__ __forceinline__ void accumulate(const LrW<double, NPX> &wt, double (&acc)[NPX]) const
    {
        double t[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) t[r] = col_lerp(wt.lx0, wt.lx1, v[r][0], v[r][1]);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int a = row(j);
            const double x = col_lerp(wt.ly0[j], wt.ly1[j], t[a], t[a + 1]);
            acc[j] = fma_t(x, x, acc[j]);
        }
    }
};

template <int NPX, unsigned CODE, int ROWS, int UU, int CH, int CCF, bool FULL>
__device__ __forceinline__ void lr_immn_steps(unsigned base, int cc, LrRowsImmN<NPX, CODE, (CH % CCF) * ROWS * UU * 16, UU * 16> &cur,
                                              const LrW<double, NPX> &wt, double (&acc)[NPX])
{
    constexpr int NREAD = 2 * LrRowsImmN<NPX, CODE, 0, UU * 16>::NR;
    if constexpr (CH + 1 < CCF) {
        LrRowsImmN<NPX, CODE, ((CH + 1) % CCF) * ROWS * UU * 16, UU * 16> nxt;
        nxt.issue(base);
        cur.template wait<NREAD>();
        if (FULL || CH < cc) cur.accumulate(wt, acc);
        lr_immn_steps<NPX, CODE, ROWS, UU, CH + 1, CCF, FULL>(base, cc, nxt, wt, acc);
    } else {
        cur.template wait<0>();
        if (FULL || CH < cc) cur.accumulate(wt, acc);
    }
}
template <int NPX, unsigned CODE, int ROWS, int UU, int CCF>
__device__ __forceinline__ void lr_immn_chunk(unsigned base, int cc, const LrW<double, NPX> &wt, double (&acc)[NPX])
{
    if (cc == CCF) {                          // (issue inside each arm: see lr_imm_chunk)
        LrRowsImmN<NPX, CODE, 0, UU * 16> first;
        first.issue(base);
        lr_immn_steps<NPX, CODE, ROWS, UU, 0, CCF, true>(base, cc, first, wt, acc);
    } else {
        LrRowsImmN<NPX, CODE, 0, UU * 16> first;
        first.issue(base);
        lr_immn_steps<NPX, CODE, ROWS, UU, 0, CCF, false>(base, cc, first, wt, acc);
    }
}

// row codes of 8 pixels the kernel is instantiated for: no step, one step before pixel g, two steps at least three pixels apart
// (source rows are >= 3 output rows tall for the scales the host sends here, <= 1/3).  X(code)
#ifndef HALO_LR8_WAVES
#define HALO_LR8_WAVES 3
#endif
#define HALO_LR_CODES8(X) \
    X(0x0000u) \
    X(0x5554u) X(0x5550u) X(0x5540u) X(0x5500u) X(0x5400u) X(0x5000u) X(0x4000u) \
    X(0xa954u) X(0xa554u) X(0x9554u) X(0xa550u) X(0x9550u) X(0x9540u) \
    X(0xaa54u) X(0xa950u) X(0xa540u) X(0x9500u)

template <int MODE, int ROWS, int UU, int PPT = LR_PPT>
__global__ void __launch_bounds__(TPB, (PPT > LR_PPT ? HALO_LR8_WAVES : 1)) k_feat_reduce_lr_dmaf(const double *__restrict__ feat, long long bstride, int C, int h, int w,
                                                             int H, int W, double sh, double sw, double ks, double rks,
                                                             double *__restrict__ out, double *__restrict__ partials)
{
    typedef double T;
    constexpr int PER = ROWS * UU, CCF = DMA_UNITS / PER, STRIDE = 2 * UU, PLANE = 2 * PER;
    constexpr int TH = PPT * (TPB / 64);                        // output rows per block: a wave owns PPT consecutive rows x 64 columns
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_smem[];
    T *img = reinterpret_cast<T *>(lr_smem);
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_bytes)lr_smem;
    const unsigned ntx = gridDim.x, nty = gridDim.y, ntiles = ntx * nty * gridDim.z;
    const unsigned lin = blockIdx.x + ntx * (blockIdx.y + nty * blockIdx.z);
    const unsigned xq = ntiles / 8, xr = ntiles % 8, xk = lin % 8;
    const unsigned tile_id = (xk < xr ? xk * (xq + 1) : xr * (xq + 1) + (xk - xr) * xq) + lin / 8;
    const int bx = (int)(tile_id % ntx), by = (int)((tile_id / ntx) % nty);
    const int b = (int)(tile_id / (ntx * nty));
    const int X0 = bx * LR_TW, Y0 = by * TH;
    const int lx = threadIdx.x & (LR_TW - 1), ly = threadIdx.x / LR_TW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = X0 + lx;
    const int xlast = (X0 + LR_TW - 1 < W ? X0 + LR_TW - 1 : W - 1);
    const int ty_lo = make_taps<T>(Y0, sh, h).i0;
    const int tx_lo = make_taps<T>(X0, sw, w).i0 & ~1, tx_hi = make_taps<T>(xlast, sw, w).i1;
    const bool xin = x < W;
    const Taps<T> tx = make_taps<T>(xin ? x : W - 1, sw, w);
    T acc[PPT];
    LrW<T, PPT> wt;
    wt.lx0 = tx.l0; wt.lx1 = tx.l1;
    int o00[PPT], o10[PPT];
    bool live[PPT];
    int a0 = 0, pat = 0;
    bool regular = true;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int y = Y0 + ly * PPT + j;
        live[j] = xin && y < H;
        const Taps<T> ty = make_taps<T>(y < H ? y : H - 1, sh, h);
        wt.ly0[j] = ty.l0; wt.ly1[j] = ty.l1;
        o00[j] = (ty.i0 - ty_lo) * STRIDE + (tx.i0 - tx_lo);
        o10[j] = (ty.i1 - ty_lo) * STRIDE + (tx.i0 - tx_lo);
        acc[j] = (T)0;
        if (j == 0) a0 = ty.i0;
        const int d = ty.i0 - a0;
        if constexpr (PPT == LR_PPT) {
            regular = regular && (d == 0 || d == 1);
            if (j > 0) pat |= (d & 1) << (j - 1);
        } else {                                                    // 2 bits per pixel: LrRowsImmN's row code
            regular = regular && d >= 0 && d <= 3;
            pat |= (d & 3) << (2 * j);
        }
    }
    const int dx1 = tx.i1 - tx.i0;
    const int upat = __builtin_amdgcn_readfirstlane(regular ? pat : -1);
    const long long hwl = (long long)h * w;
    unsigned soff[DMA_PER_WAVE], sch[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int u = (i * (TPB / 64) + wv) * 64 + lane;
        const int ch = u / PER, rem = u - ch * PER, r = rem / UU, pp = rem - r * UU;      // compile-time divisors
        const int sr = ty_lo + r < h - 1 ? ty_lo + r : h - 1;
        int q0 = tx_lo + 2 * pp;
        q0 = q0 + 1 > w - 1 ? w - 2 : q0;                        // pairs past the last column (one of them is read: patched below)
        // CCF * PER < DMA_UNITS: the trailing units of a wave's last DMA belong to no channel of the chunk (their LDS slots are
        // never read).  They re-load the chunk's last channel instead of channel c0 + CCF, which for a FULL last chunk
        // (C a multiple of CCF) would be one plane past the image -- past the tensor for the last image of the batch
        sch[i] = (unsigned)(ch < CCF ? ch : CCF - 1);
        soff[i] = (unsigned)(sr * w + q0);
    }
    const T *fb = feat + (size_t)b * bstride;
    // per-lane source pointers of the chunk to be issued next, advanced by CCF planes per chunk (two VALU instructions per
    // pointer and chunk instead of re-deriving channel * plane + offset); the last, partial chunk clamps its channel
    const T *sp[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) sp[i] = fb + (size_t)sch[i] * hwl + soff[i];
    const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(wv * 64) * 16u));
    // ONE straight-line block of DMA_PER_WAVE instructions per chunk, whatever the chunk: the counted wait below counts them, and
    // halo_amd/_asmcheck.py (run by the build) verifies on the emitted assembly that nothing else is outstanding at that wait.  A
    // partial last chunk steps the lanes whose channel lies past it back onto its last channel (round 4 had one issue block per
    // case: the compiler spilled the second block's pointers and reloaded each with s_waitcnt vmcnt(0) between two DMAs)
    auto issue = [&](int c0, int buf) {
        const int cc = C - c0 < CCF ? C - c0 : CCF;
        long long back[DMA_PER_WAVE];
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) back[i] = 0;
        if (cc != CCF) {
#pragma unroll
            for (int i = 0; i < DMA_PER_WAVE; ++i) {
                const int ex = (int)sch[i] - (cc - 1);
                back[i] = ex > 0 ? (long long)ex * hwl : 0ll;
            }
        }
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            glds16(sp[i] - back[i], dst0 + (unsigned)buf * (DMA_UNITS * 16) + (unsigned)(i * (TPB / 64) * 64) * 16u);
            sp[i] += (size_t)CCF * hwl;
        }
    };
    const int pu = (w - tx_lo) >> 1;                              // the pair that starts at column w, if the image holds it
    const bool patch = tx_hi + 1 >= w && pu < UU;                 // block-uniform: the clamped extra column lies past the row
    const int nchunks = (C + CCF - 1) / CCF;
    issue(0, 0);
    for (int n = 0; n < nchunks; ++n) {
        const int c0 = n * CCF, cc = C - c0 < CCF ? C - c0 : CCF;
        if (n + 1 < nchunks) {
            issue(c0 + CCF, (n + 1) & 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        T *tile = img + (size_t)(n & 1) * (DMA_UNITS * 2);
        if (patch) {
            for (int e = threadIdx.x; e < cc * ROWS; e += TPB) {
                T *q = tile + (size_t)(e / ROWS) * PLANE + (size_t)(e % ROWS) * STRIDE + 2 * pu;
                q[0] = q[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const unsigned base = lds0 + (unsigned)(n & 1) * (DMA_UNITS * 16) + (unsigned)o00[0] * 8u;
        bool done = false;
        if constexpr (PPT == LR_PPT) {
            done = true;
            if (upat == 0) lr_imm_chunk<0, ROWS, UU, CCF>(base, cc, wt, acc);
            else if (upat == 4) lr_imm_chunk<4, ROWS, UU, CCF>(base, cc, wt, acc);
            else if (upat == 6) lr_imm_chunk<6, ROWS, UU, CCF>(base, cc, wt, acc);
            else if (upat == 7) lr_imm_chunk<7, ROWS, UU, CCF>(base, cc, wt, acc);
            else done = false;
        } else {
            switch (upat) {
#define HALO_LR_CASE(CODE_) case (int)CODE_: lr_immn_chunk<PPT, CODE_, ROWS, UU, CCF>(base, cc, wt, acc); done = true; break;
                HALO_LR_CODES8(HALO_LR_CASE)
#undef HALO_LR_CASE
                default: break;
            }
        }
        if (!done) {
#pragma unroll 1
            for (int ch = 0; ch < cc; ++ch) {
                const T *tp = tile + (size_t)ch * PLANE;
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const T v = bilerp<T>(tp[o00[j]], tp[o00[j] + dx1], tp[o10[j]], tp[o10[j] + dx1], wt.lx0, wt.lx1, wt.ly0[j], wt.ly1[j]);
                    acc[j] = fma_t(v, v, acc[j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    double mn = 0.0, mx = 0.0;
    bool have = false;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        if (!live[j]) continue;
        const int y = Y0 + ly * PPT + j;
        T r;
        if constexpr (MODE == 0) r = dist0_from_ssq(acc[j], ks, rks);
        else r = __builtin_sqrt(acc[j]);
        out[(size_t)b * H * W + (size_t)y * W + x] = r;
        if (!have) { mn = mx = (double)r; have = true; }
        else { mn = nan_min(mn, (double)r); mx = nan_max(mx, (double)r); }
    }
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }
    __syncthreads();
    if (!have) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + (size_t)tile_id * 2);
}

// ---- Gram form of the low-resolution radius ("precompute the 4-neighbour Gram terms per low-res cell,
// then each output pixel costs O(10) not O(C)").  The interpolated embedding of an output pixel is sum_i w_i v_i over the
// four corner vectors of its low-res cell, so  ||.||^2 = sum_{i<=j} (2 - [i==j]) w_i w_j <v_i, v_j>.  The 10 inner products
// of a cell are 5 maps over the low-res grid read at the cell's corners (n(.) = neighbour clamped to the grid, exactly
// the i1 of make_taps):
//     S(y,x) = <v,v>   Hh(y,x) = <v(y,x), v(y,n(x))>   Vv(y,x) = <v(y,x), v(n(y),x)>
//     D1(y,x) = <v(y,x), v(n(y),n(x))>   D2(y,x) = <v(y,n(x)), v(n(y),x)>
// k_gram_lr computes the maps in one pass over the low-res tensor (sequential fma chains over the channels);
// k_radius_gram evaluates the 10-term form per output pixel.  Mathematically the same number as k_feat_reduce_lr,
// rounded differently (a few 1e-16 of the largest corner norm; pixels whose terms cancel take the exact order: GRAM_GUARD),
// so this mode is NOT bit-identical to upsample-then-score: it is opt-in (halo_score_maps_lr_gram) and float64 only.
constexpr int GRAM_MAPS = 5;
constexpr int GRAM_COLS = 63;          // columns per wave: lane 63 only supplies the right neighbour of lane 62

// value of the next lane (lane 63: unspecified): one full-wave DPP shift per 32-bit half
__device__ __forceinline__ double next_lane(double v)
{
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffll), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// A wave owns 63 columns x 2 rows of the low-res grid: per channel it loads its columns of three rows (the third is the
// lower neighbour of the second) and takes the right neighbours from the next lane -- 3 loads for 126 pixels' 5 products
// each, where a lane per pixel with four loads made the kernel L1-bound (2.9 TB/s of the tensor).  Blocks that share an
// XCD take a contiguous eighth of the wave list, so the row shared by two row pairs is fetched by one L2.
__global__ void __launch_bounds__(TPB) k_gram_lr(const double *__restrict__ feat, long long bstride, int C, int h, int w,
                                                 double *__restrict__ gram)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const unsigned per = gridDim.x / 8;                       // the host pads the grid's x extent to a multiple of 8
    const unsigned wid = ((blockIdx.x % 8) * per + blockIdx.x / 8) * (TPB / 64) + (threadIdx.x >> 6);
    const unsigned nwx = (unsigned)((w + GRAM_COLS - 1) / GRAM_COLS);
    const int y = 2 * (int)(wid / nwx);
    if (y >= h) return;                                       // wave-uniform
    const int xu = (int)(wid % nwx) * GRAM_COLS + lane, x = xu < w - 1 ? xu : w - 1;
    const int y1 = y + 1 < h - 1 ? y + 1 : h - 1, y2 = y + 2 < h - 1 ? y + 2 : h - 1;
    const unsigned a0 = (unsigned)(y * w + x), a1 = (unsigned)(y1 * w + x), a2 = (unsigned)(y2 * w + x);
    const int hwl = h * w;
    const double *p = feat + (size_t)b * bstride;             // plane base: scalar, advanced by scalar adds
    double g0[GRAM_MAPS], g1[GRAM_MAPS];
#pragma unroll
    for (int k = 0; k < GRAM_MAPS; ++k) g0[k] = g1[k] = 0.0;
    auto accumulate = [&](double v0, double v1, double v2) {
        const double r0 = next_lane(v0), r1 = next_lane(v1), r2 = next_lane(v2);
        g0[0] = __builtin_fma(v0, v0, g0[0]); g0[1] = __builtin_fma(v0, r0, g0[1]); g0[2] = __builtin_fma(v0, v1, g0[2]);
        g0[3] = __builtin_fma(v0, r1, g0[3]); g0[4] = __builtin_fma(r0, v1, g0[4]);
        g1[0] = __builtin_fma(v1, v1, g1[0]); g1[1] = __builtin_fma(v1, r1, g1[1]); g1[2] = __builtin_fma(v1, v2, g1[2]);
        g1[3] = __builtin_fma(v1, r2, g1[3]); g1[4] = __builtin_fma(r1, v2, g1[4]);
    };
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u, p += hwl) { v[u][0] = p[a0]; v[u][1] = p[a1]; v[u][2] = p[a2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) accumulate(v[u][0], v[u][1], v[u][2]);
    }
    for (; c < C; ++c, p += hwl) accumulate(p[a0], p[a1], p[a2]);
    if (lane < GRAM_COLS && xu < w) {
        double *gb = gram + (size_t)b * GRAM_MAPS * hwl;
#pragma unroll
        for (int k = 0; k < GRAM_MAPS; ++k) gb[(size_t)k * hwl + a0] = g0[k];
        if (y + 1 < h) {
#pragma unroll
            for (int k = 0; k < GRAM_MAPS; ++k) gb[(size_t)k * hwl + a1] = g1[k];
        }
    }
}

// k_gram_lr with 16 bytes per lane (even source width, 16-byte aligned planes): a lane owns TWO adjacent columns of two rows,
// a wave 128 columns x 2 rows.  The right neighbour of a lane's first column is its own second column, that of its second
// column the next lane's first (full-wave DPP shift); only lane 63 loads its right neighbour itself (8 bytes per row and
// channel, clamped at the row's end).  Half the load instructions per byte of k_gram_lr, no idle 64th lane, and 512-column rows
// split into four full waves.  Same fma chains, same bits.
template <int UCH, bool NT, int R>
__global__ void __launch_bounds__(TPB) k_gram_lr2(const double *__restrict__ feat, long long bstride, int C, int h, int w,
                                                  double *__restrict__ gram)
{
    // R rows per wave (+ the lower neighbour of the last one): every source row is loaded (R + 1) / R times
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const unsigned per = gridDim.x / 8;                       // the host pads the grid's x extent to a multiple of 8
    const unsigned wid = ((blockIdx.x % 8) * per + blockIdx.x / 8) * (TPB / 64) + (threadIdx.x >> 6);
    const unsigned nwx = (unsigned)((w + 127) / 128);
    const int y = R * (int)(wid / nwx);
    if (y >= h) return;                                       // wave-uniform
    const int xu = (int)(wid % nwx) * 128 + 2 * lane;         // this lane's first column (even)
    const int x = xu < w - 2 ? xu : w - 2;                    // lanes past the row re-read its last pair (never stored)
    const int xr = xu + 2 < w - 1 ? xu + 2 : w - 1;           // right neighbour of the second column, clamped (lane 63 loads it)
    unsigned a[R + 1], e[R + 1];
#pragma unroll
    for (int r = 0; r <= R; ++r) {
        const int yr = y + r < h - 1 ? y + r : h - 1;
        a[r] = (unsigned)(yr * w + x);
        e[r] = (unsigned)(yr * w + xr);
    }
    const int hwl = h * w;
    const double *p = feat + (size_t)b * bstride;             // plane base: scalar, advanced by scalar adds
    double g[R][2][GRAM_MAPS];                                // [row][column of the pair][map]
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < GRAM_MAPS; ++k) g[r][q][k] = 0.0;
    const bool last = lane == 63, edge = xu + 2 >= w;         // edge: the pair ends the row, its right neighbour is clamped to itself
    auto accumulate = [&](const d2_t (&v)[R + 1], const double (&xx)[R + 1]) {
        // right neighbours of the pair's second column: the next lane's first column, (lane 63) the value it loaded itself, or
        // (last pair of the row) the column itself
        double cc[R + 1][2], rr[R + 1][2];
#pragma unroll
        for (int r = 0; r <= R; ++r) {
            const double sh = next_lane(v[r].x);
            const double nb = edge ? v[r].y : (last ? xx[r] : sh);
            cc[r][0] = v[r].x; cc[r][1] = v[r].y; rr[r][0] = v[r].y; rr[r][1] = nb;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                g[r][q][0] = __builtin_fma(cc[r][q], cc[r][q], g[r][q][0]); g[r][q][1] = __builtin_fma(cc[r][q], rr[r][q], g[r][q][1]);
                g[r][q][2] = __builtin_fma(cc[r][q], cc[r + 1][q], g[r][q][2]); g[r][q][3] = __builtin_fma(cc[r][q], rr[r + 1][q], g[r][q][3]);
                g[r][q][4] = __builtin_fma(rr[r][q], cc[r + 1][q], g[r][q][4]);
            }
    };
    // UCH channels ((R + 1) x 16 bytes each) in flight per lane.  NT (non-temporal loads) measured 20 % SLOWER: every row is read
    // a second time, right away, as the neighbour row of the wave above -- an L2 hit that `nt` gives up (profiles/archive/r04_gram_ab.txt)
    auto ld2 = [](const double *q) -> d2_t {
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(q));
        else return *reinterpret_cast<const d2_t *>(q);
    };
    int c = 0;
    for (; c + UCH <= C; c += UCH) {
        d2_t v[UCH][R + 1];
        double xe[UCH][R + 1];
#pragma unroll
        for (int u = 0; u < UCH; ++u, p += hwl) {
#pragma unroll
            for (int r = 0; r <= R; ++r) { v[u][r] = ld2(p + a[r]); xe[u][r] = 0.0; }
            if (last) {
#pragma unroll
                for (int r = 0; r <= R; ++r) xe[u][r] = p[e[r]];
            }
        }
#pragma unroll
        for (int u = 0; u < UCH; ++u) accumulate(v[u], xe[u]);
    }
    for (; c < C; ++c, p += hwl) {
        d2_t v[R + 1];
        double xx[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) { v[r] = *reinterpret_cast<const d2_t *>(p + a[r]); xx[r] = last ? p[e[r]] : 0.0; }
        accumulate(v, xx);
    }
    if (xu < w) {                                             // w even: both columns of the pair exist
        double *gb = gram + (size_t)b * GRAM_MAPS * hwl;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (y + r < h) {
#pragma unroll
                for (int k = 0; k < GRAM_MAPS; ++k) *reinterpret_cast<d2_t *>(gb + (size_t)k * hwl + a[r]) = (d2_t){g[r][0][k], g[r][1][k]};
            }
    }
}

// Cancellation guard: when neighbouring low-res vectors point in opposing directions the 10 terms cancel and the
// Gram form keeps only the bits of s that stand above ~2^-53 of the terms' magnitudes t = sum |coef * G|.  A pixel with
// s < 2^-10 t takes the EXACT evaluation order instead (interpolate every channel, then the fma chain over the squares --
// what upsample-then-score computes, bit for bit), so the Gram value is only ever used where it carries >= 43 good bits:
// |s_gram - s_exact| <= 1.3e-10 s for C <= 256 (4 C u 2^10; observed < 1e-12).  oracle/halo_oracle.c:halo_o_gram_radius
// states the same rule.
constexpr double GRAM_GUARD = 0x1p-10;

template <int MODE>
__global__ void __launch_bounds__(TPB) k_radius_gram(const double *__restrict__ gram, const double *__restrict__ feat, long long bstride,
                                                     int C, int h, int w, int H, int W, double sh, double sw,
                                                     double ks, double rks, double *__restrict__ out, double *__restrict__ partials)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool live = i < hw;
    double mn = 0.0, mx = 0.0;
    if (live) {
        // pixel coordinates: 32-bit division where the map allows it (a 64-bit one is ~40 instructions per pixel)
        int y, x;
        if (hw <= 0x7fffffffll) { const unsigned iu = (unsigned)i; y = (int)(iu / (unsigned)W); x = (int)(iu - (unsigned)y * (unsigned)W); }
        else { y = (int)(i / W); x = (int)(i % W); }
        const Taps<double> ty = make_taps<double>(y, sh, h), tx = make_taps<double>(x, sw, w);
        const double wt[4] = {ty.l0 * tx.l0, ty.l0 * tx.l1, ty.l1 * tx.l0, ty.l1 * tx.l1};      // the weights of the four corner vectors
        const size_t hwl = (size_t)h * w;
        const double *S = gram + (size_t)b * GRAM_MAPS * hwl, *Hh = S + hwl, *Vv = Hh + hwl, *D1 = Vv + hwl, *D2 = D1 + hwl;
        const unsigned c00 = (unsigned)(ty.i0 * w + tx.i0), c01 = (unsigned)(ty.i0 * w + tx.i1), c10 = (unsigned)(ty.i1 * w + tx.i0),
                       c11 = (unsigned)(ty.i1 * w + tx.i1);      // h * w < 2^31 (checked by the host)
        // <v_a, v_b> for corners a <= b in the order (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
        const double G[10] = {S[c00], Hh[c00], Vv[c00], D1[c00], S[c01], D2[c00], Vv[c01], S[c10], Hh[c10], S[c11]};
        double s = 0.0, t = 0.0;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c2 = a; c2 < 4; ++c2, ++k) {
                double coef = wt[a] * wt[c2];
                if (c2 != a) coef = coef + coef;
                s = __builtin_fma(coef, G[k], s);
                t = __builtin_fma(coef, __builtin_fabs(G[k]), t);
            }
        if (s < t * GRAM_GUARD) {                 // cancellation (covers every negative residue); NaN compares false and stays NaN
            const double *pl = feat + (size_t)b * bstride;
            double acc = 0.0;
            for (int ch = 0; ch < C; ++ch, pl += hwl) {
                const double v = bilerp<double>(pl[c00], pl[c01], pl[c10], pl[c11], tx.l0, tx.l1, ty.l0, ty.l1);
                acc = __builtin_fma(v, v, acc);
            }
            s = acc;
        }
        double r;
        if constexpr (MODE == 0) r = dist0_from_ssq(s, ks, rks);
        else r = __builtin_sqrt(s);
        out[(size_t)b * hw + i] = r;
        mn = mx = r;
    }
    __shared__ double seed[2];
    if (threadIdx.x == 0) { seed[0] = mn; seed[1] = mx; }         // thread 0 of every block is live
    __syncthreads();
    if (!live) { mn = seed[0]; mx = seed[1]; }
    block_minmax<TPB>(mn, mx, partials + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

// Logits: interpolate the O class planes at one output pixel, then the same entropy / prediction code.
template <int O_T>
__global__ void __launch_bounds__(TPB) k_logit_maps_lr(const float *__restrict__ logit, long long bstride, int h, int w, int H, int W,
                                                       float sh, float sw, const long long *__restrict__ gt, int unc_type,
                                                       int pur_type, float *__restrict__ ent, short *__restrict__ pred)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
#if HALO_LOGF_LDS
    __shared__ double s_ltab[128][2];
    stage_logf_table<TPB>(s_ltab);
    const double (*ltab)[2] = s_ltab;
#else
    const double (*ltab)[2] = logf_tab_;
#endif
    if (i >= hw) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const Taps<float> ty = make_taps<float>(y, sh, h), tx = make_taps<float>(x, sw, w);
    // plane base in SGPRs (advanced by scalar adds) + four 32-bit tap offsets: no 64-bit address arithmetic per load
    // (byte offsets in 32 bits: a low-res class plane is far below 4 GiB)
    const char *pl = reinterpret_cast<const char *>(logit + (size_t)b * bstride);
    const unsigned a00 = (unsigned)(ty.i0 * w + tx.i0) * 4u, a01 = (unsigned)(ty.i0 * w + tx.i1) * 4u,
                   a10 = (unsigned)(ty.i1 * w + tx.i0) * 4u, a11 = (unsigned)(ty.i1 * w + tx.i1) * 4u;
    const size_t plane_bytes = (size_t)h * w * 4;
    auto at = [](const char *base, unsigned off) { return *reinterpret_cast<const float *>(base + off); };
    float p[1][O_T];
#pragma unroll
    for (int c = 0; c < O_T; ++c, pl += plane_bytes)
        p[0][c] = bilerp<float>(at(pl, a00), at(pl, a01), at(pl, a10), at(pl, a11), tx.l0, tx.l1, ty.l0, ty.l1);
    const bool need_gt = unc_type == HALO_UNC_ORACLE_ACC || pur_type == HALO_PUR_ORACLE_RIPU;
    const long long g[1] = {need_gt ? gt[(size_t)b * hw + i] : 0};
    float e[1];
    int pr[1];
    if (softmax_lean<O_T, 1>(p)) {
        finish_px<O_T, 1, true>(p, unc_type, pur_type, g, e, pr, pred != nullptr, ltab);
    } else {
        softmax_general<O_T, 1>(p);
        finish_px<O_T, 1, false>(p, unc_type, pur_type, g, e, pr, pred != nullptr);
    }
    ent[(size_t)b * hw + i] = e[0];
    if (pred) pred[(size_t)b * hw + i] = (short)pr[0];
}

// any class count: materialise the interpolated logits of a strip into scratch, then the generic kernel
__global__ void __launch_bounds__(TPB) k_logit_interp_lr(const float *__restrict__ logit, long long bstride, int O, int h, int w, int H,
                                                         int W, float sh, float sw, float *__restrict__ dst)
{
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= hw) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const Taps<float> ty = make_taps<float>(y, sh, h), tx = make_taps<float>(x, sw, w);
    const float *lb = logit + (size_t)b * bstride;
    for (int c = 0; c < O; ++c) {
        const float *pl = lb + (size_t)c * h * w;
        dst[((size_t)b * O + c) * hw + i] = bilerp<float>(pl[(size_t)ty.i0 * w + tx.i0], pl[(size_t)ty.i0 * w + tx.i1],
                                                          pl[(size_t)ty.i1 * w + tx.i0], pl[(size_t)ty.i1 * w + tx.i1], tx.l0, tx.l1, ty.l0, ty.l1);
    }
}

// ---------------------------------------------------------------- host side
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

struct FusedLogit { const float *logit; long long bstride; int O; int unc_type; float *ent; };

template <typename T, int VEC>
static void launch_feat(const T *feat, long long bstride, int C, long long hw, int B, int mode, double ks, double rks,
                        T *out, double *partials, int nblk, hipStream_t st, const FusedLogit *fl = nullptr)
{
    dim3 grid(nblk, B), block(FTPB);
    constexpr int UNROLL = HALO_FEAT_UNROLL;      // channel planes in flight per lane (16: no gain beside the selection kernels, 6 % slower alone)
    // granule of the XCD-contiguous chunk map: 256 chunks (512 KiB per plane), halved until a group of 8 granules fits
    const char *eg = getenv("HALO_FEAT_XCD_GRANULE");      // A/B switch, read per call: -1 = the plain map
    const int env_g = eg ? atoi(eg) : 256;
    unsigned xcd_g = (unsigned)(env_g < 0 ? 0 : env_g);
    while (xcd_g > 1 && 8 * xcd_g > (unsigned)nblk) xcd_g >>= 1;
    if (8 * xcd_g > (unsigned)nblk) xcd_g = 0;
#define HALO_FEAT(M, FO_)                                                                                               \
    hipLaunchKernelGGL((k_feat_reduce<T, VEC, M, UNROLL, FO_>), grid, block, 0, st, feat, bstride, C, hw, ks, rks, out, \
                       partials, fl ? fl->logit : nullptr, fl ? fl->bstride : 0ll, fl ? fl->unc_type : 0, fl ? fl->ent : nullptr, \
                       xcd_g)
    const int fo = fl ? fl->O : 0;
    if (mode == 0) {
        if (fo == 19) HALO_FEAT(0, 19); else if (fo == 16) HALO_FEAT(0, 16); else HALO_FEAT(0, 0);
    } else {
        if (fo == 19) HALO_FEAT(1, 19); else if (fo == 16) HALO_FEAT(1, 16); else HALO_FEAT(1, 0);
    }
#undef HALO_FEAT
}

}  // namespace halo

using namespace halo;

static_assert(RI_TW == LR_TW && RI_TH == LR_TH, "k_region_impurity3 writes one min/max partial per tile of the same shape");
static size_t partial_slots(int64_t H, int64_t W)
{
    const size_t a = (size_t)cdiv(H * W, FTPB), b = (size_t)(cdiv(W, LR_TW) * cdiv(H, LR_TH));
    return a > b ? a : b;
}

extern "C" size_t halo_score_workspace_bytes(int64_t B, int64_t H, int64_t W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t n = (size_t)B * H * W;
    // per-block min/max slots: one per FTPB pixels, or one per 64 x 16 tile of the low-res scorer (narrow maps
    // such as 4096 x 4 have more tiles than 128-pixel blocks)
    const size_t nblk = partial_slots(H, W);
    size_t s = 0;
    s += align_up(n * 4, 256);                   // ent
    s += align_up(n * 4, 256);                   // unc_raw
    s += align_up(n * 8, 256);                   // imp_raw
    s += align_up(n * 2, 256);                   // pred
    s += 2 * align_up((size_t)B * nblk * 2 * 8, 256);  // partials (imp, unc)
    s += align_up((size_t)B * 4 * 8, 256);       // stats
    return s + 1024;
}

extern "C" int halo_score_maps(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                               int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                               int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                               int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    return halo_score_maps_timed(logit, logit_bstride, feat, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type,
                                 pur_type, normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace,
                                 workspace_bytes, stream, nullptr, nullptr, nullptr);
}

extern "C" size_t halo_score_lr_workspace_bytes(int64_t B, int64_t O, int64_t H, int64_t W)
{
    const size_t base = halo_score_workspace_bytes(B, H, W);
    if (base == 0 || O <= 0) return 0;
    return base + ((O == 19 || O == 16) ? 0 : (size_t)B * O * H * W * 4 + 512);
}

extern "C" size_t halo_score_lr_gram_workspace_bytes(int64_t B, int64_t O, int64_t H, int64_t W, int64_t hf, int64_t wf)
{
    const size_t base = halo_score_lr_workspace_bytes(B, O, H, W);
    if (base == 0 || hf <= 0 || wf <= 0) return 0;
    return base + (size_t)B * 5 * hf * wf * 8 + 512;
}

// low-res source geometry (halo_score_maps_lr); gram: the embedding's radius through the Gram form (float64 only)
struct LrDims { int hl, wl, hf, wf; bool gram; };

template <typename T>
static void lr_window(int out_size, int in_size, int tile, int &max_span)
{
    const T sc = out_size > 1 ? (T)(in_size - 1) / (T)(out_size - 1) : (T)0;
    max_span = 1;
    for (int o0 = 0; o0 < out_size; o0 += tile) {
        const int ol = o0 + tile - 1 < out_size ? o0 + tile - 1 : out_size - 1;
        T f = sc * (T)o0; int lo = (int)f; lo = lo > in_size - 1 ? in_size - 1 : lo;
        f = sc * (T)ol; int hi = (int)f; hi = hi > in_size - 1 ? in_size - 1 : hi; hi = hi + (hi < in_size - 1 ? 1 : 0);
        if (hi - lo + 1 > max_span) max_span = hi - lo + 1;
    }
}

template <typename T>
static int launch_feat_lr(const T *feat, long long bstride, int C, const LrDims &lr, int H, int W, int B, int mode, double ks,
                          double rks, T *out, double *partials, int &nblk, hipStream_t st)
{
    int max_rows, max_cols;
    lr_window<T>(H, lr.hf, LR_TH, max_rows);
    lr_window<T>(W, lr.wf, LR_TW, max_cols);
    ++max_rows; ++max_cols;       // the staged window carries one clamped extra row and column (k_feat_reduce_lr)
    const T sh = H > 1 ? (T)(lr.hf - 1) / (T)(H - 1) : (T)0, sw = W > 1 ? (T)(lr.wf - 1) / (T)(W - 1) : (T)0;
    dim3 grid((unsigned)cdiv(W, LR_TW), (unsigned)cdiv(H, LR_TH), (unsigned)B), block(TPB);
    nblk = (int)(grid.x * grid.y);
    if constexpr (sizeof(T) == 8) {
        // LDS-DMA double buffer (k_feat_reduce_lr_dma): float64, even source width (16-byte aligned pairs), at least 4 channels
        // of the largest window per 16 KiB image; HALO_LR_NODMA=1 keeps the register-staged kernel (A/B switch, same bits)
        const int units_per_ch = max_rows * ((max_cols + 2) / 2 + 1);        // even start column: up to one more pair per row
        int CCd = DMA_UNITS / units_per_ch;
        CCd = CCd > C ? C : CCd;
        const bool dma_ok = lr.wf >= 2 && lr.wf % 2 == 0 && bstride % 2 == 0 && aligned16(feat) && CCd >= 4 && getenv("HALO_LR_NODMA") == nullptr;
        if (dma_ok) {
            const size_t lds = 2 * (size_t)DMA_UNITS * 16;
            // compile-time image geometries (k_feat_reduce_lr_dmaf): rows x pairs that cover the launch's largest window
            const int need_rows = max_rows, need_u = (max_cols + 2) / 2;
            const bool nofixed = getenv("HALO_LR_NOFIXED") != nullptr;                 // A/B switch: runtime strides
            // 8 pixels per lane (64 x 32 output tiles) where a source row is at least 3 output rows tall: the row codes the kernel
            // is instantiated for (HALO_LR_CODES8) then cover every wave; HALO_LR_PPT4=1 keeps the 4-pixel kernel (A/B, same bits)
            if (!nofixed && (double)sh <= 1.0 / 3.0 && H >= 2 * LR_TH && getenv("HALO_LR_PPT4") == nullptr) {
                int max_rows8;
                lr_window<T>(H, lr.hf, 2 * LR_TH, max_rows8);
                ++max_rows8;
                dim3 grid8((unsigned)cdiv(W, LR_TW), (unsigned)cdiv(H, 2 * LR_TH), (unsigned)B);
#define HALO_LR_FIXED8(R_, U_)                                                                                                             \
                if (max_rows8 <= R_ && need_u <= U_) {                                                                                     \
                    nblk = (int)(grid8.x * grid8.y);                                                                                       \
                    if (mode == 0) hipLaunchKernelGGL((k_feat_reduce_lr_dmaf<0, R_, U_, 2 * LR_PPT>), grid8, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, ks, rks, (double *)out, partials); \
                    else hipLaunchKernelGGL((k_feat_reduce_lr_dmaf<1, R_, U_, 2 * LR_PPT>), grid8, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, ks, rks, (double *)out, partials);           \
                    return HALO_OK;                                                                                                        \
                }
                HALO_LR_FIXED8(8, 8)       // x6.4 (160x320 -> 1024x2048): 16 channels per image
                HALO_LR_FIXED8(11, 10)     // x4: 9 channels per image
#undef HALO_LR_FIXED8
            }
#define HALO_LR_FIXED(R_, U_)                                                                                                              \
            if (!nofixed && need_rows <= R_ && need_u <= U_) {                                                                             \
                if (mode == 0) hipLaunchKernelGGL((k_feat_reduce_lr_dmaf<0, R_, U_>), grid, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, ks, rks, (double *)out, partials); \
                else hipLaunchKernelGGL((k_feat_reduce_lr_dmaf<1, R_, U_>), grid, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, ks, rks, (double *)out, partials);           \
                return HALO_OK;                                                                                                            \
            }
            HALO_LR_FIXED(6, 8)        // x6.4 (160x320 -> 1024x2048): 21 channels per image
            HALO_LR_FIXED(7, 10)       // x4: 14 channels per image
            HALO_LR_FIXED(8, 12)
#undef HALO_LR_FIXED
            if (mode == 0) hipLaunchKernelGGL((k_feat_reduce_lr_dma<0>), grid, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, CCd, ks, rks, (double *)out, partials);
            else hipLaunchKernelGGL((k_feat_reduce_lr_dma<1>), grid, block, lds, st, (const double *)feat, bstride, C, lr.hf, lr.wf, H, W, (double)sh, (double)sw, CCd, ks, rks, (double *)out, partials);
            return HALO_OK;
        }
    }
    const size_t plane_bytes = (size_t)max_rows * max_cols * sizeof(T);
    if (plane_bytes > 48 * 1024) return fail(HALO_E_UNSUPPORTED, "halo_score_maps_lr: source window too large for LDS (downsampling?)");
    // channels per chunk: what the block's four waves prefetch in one group each (k_feat_reduce_lr), LDS permitting
    int CC = (int)((48 * 1024) / plane_bytes);
    CC = CC > (TPB / 64) * LR_STAGE_G ? (TPB / 64) * LR_STAGE_G : CC;
    CC = CC > C ? C : CC;
    const size_t lds = (size_t)CC * plane_bytes;
    if (mode == 0) hipLaunchKernelGGL((k_feat_reduce_lr<T, 0>), grid, block, lds, st, feat, bstride, C, lr.hf, lr.wf, H, W, sh, sw, max_rows, max_cols, CC, ks, rks, out, partials);
    else hipLaunchKernelGGL((k_feat_reduce_lr<T, 1>), grid, block, lds, st, feat, bstride, C, lr.hf, lr.wf, H, W, sh, sw, max_rows, max_cols, CC, ks, rks, out, partials);
    return HALO_OK;
}

// torch's own limits on the padded convolution (F.pad): 'reflect' needs pad < size, 'circular' pad <= size, in both dimensions
static int check_padding(int pad, int k, int64_t H, int64_t W, const char *who)
{
    if (pad < HALO_PAD_ZEROS || pad > HALO_PAD_CIRCULAR) return fail(HALO_E_ARG, "%s: bad padding mode %d", who, pad);
    const int64_t r = k / 2, m = H < W ? H : W;
    if (pad == HALO_PAD_REFLECT && r >= m)
        return fail(HALO_E_ARG, "%s: padding_mode='reflect' needs the padding (%lld) to be smaller than the map (%lld x %lld)", who, (long long)r, (long long)H, (long long)W);
    if (pad == HALO_PAD_CIRCULAR && r > m)
        return fail(HALO_E_ARG, "%s: padding_mode='circular' needs the padding (%lld) to be at most the map size (%lld x %lld)", who, (long long)r, (long long)H, (long long)W);
    return HALO_OK;
}

static int score_impl(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                      int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                      int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                      int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                      void *workspace, size_t workspace_bytes, void *stream, void *ev_feat_start,
                      void *ev_feat_stop, const LrDims *lr, void *ev_logit_start = nullptr, void *ev_logit_stop = nullptr,
                      void *score_range = nullptr, void *tail_stream = nullptr, void *ev_feat_mid = nullptr, void *ev_tail_stop = nullptr)
{
    hipStream_t st = (hipStream_t)stream;
    // `normalize` is the flags word of include/halo_hip.h: bit 0 = normalise, bits 8-9 = padding mode of the two box windows
    const int pad = (normalize >> 8) & 3;
    if (normalize & ~(HALO_FLAG_NORMALIZE | (3 << 8))) return fail(HALO_E_ARG, "halo_score_maps: unknown flag bits 0x%x", normalize);
    normalize &= HALO_FLAG_NORMALIZE;
    if (!logit || !score || B <= 0 || O <= 0 || H <= 0 || W <= 0) return fail(HALO_E_ARG, "halo_score_maps: null/empty argument");
    if (unc_type < 0 || unc_type > HALO_UNC_ZEROS) return fail(HALO_E_ARG, "halo_score_maps: bad unc_type %d", unc_type);
    if (pur_type < 0 || pur_type > HALO_PUR_EUC_NORM) return fail(HALO_E_UNSUPPORTED, "Error: purity type '%d' not implemented", pur_type);
    if (ksize < 1 || !(ksize & 1) || pksize < 1 || !(pksize & 1)) return fail(HALO_E_ARG, "halo_score_maps: window sizes must be odd");
    {
        const int rc = check_padding(pad, ksize > pksize ? ksize : pksize, H, W, "halo_score_maps");
        if (rc != HALO_OK) return rc;
    }
    const bool need_feat = pur_type == HALO_PUR_HYPER || pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM;
    if (need_feat && (!feat || C <= 0)) return fail(HALO_E_ARG, "halo_score_maps: decoder_out required for this purity type");
    if (need_feat && feat_dtype != HALO_F32 && feat_dtype != HALO_F64) return fail(HALO_E_ARG, "halo_score_maps: bad feat dtype");
    const bool need_gt = unc_type == HALO_UNC_ORACLE_ACC || pur_type == HALO_PUR_ORACLE_RIPU;
    if (need_gt && !gt) return fail(HALO_E_ARG, "halo_score_maps: ground_truth required");
    if (pur_type == HALO_PUR_HYPER && (K < 1 || K > 32767)) return fail(HALO_E_UNSUPPORTED, "halo_score_maps: K out of range");
    if (O > 32767) return fail(HALO_E_UNSUPPORTED, "halo_score_maps: too many classes");
    if (workspace_bytes < halo_score_workspace_bytes(B, H, W) || !workspace) return fail(HALO_E_WORKSPACE, "halo_score_maps: workspace too small");
    const bool lr_generic_O = lr && !(O == 19 || O == 16);
    if (lr_generic_O && workspace_bytes < halo_score_workspace_bytes(B, H, W) + (size_t)B * O * H * W * 4 + 256)
        return fail(HALO_E_WORKSPACE, "halo_score_maps_lr: workspace too small for %d classes", (int)O);

    const long long hw = (long long)H * W;
    const int nblk1 = (int)cdiv(hw, TPB);
    Arena ar(workspace, workspace_bytes);
    float *ent = ar.take<float>((size_t)B * hw);
    float *unc_raw = ar.take<float>((size_t)B * hw);
    double *imp_raw = ar.take<double>((size_t)B * hw);
    short *pred = ar.take<short>((size_t)B * hw);
    double *part_imp = ar.take<double>((size_t)B * partial_slots(H, W) * 2);
    double *part_unc = ar.take<double>((size_t)B * partial_slots(H, W) * 2);
    double *stats = ar.take<double>((size_t)B * 4);
    float *lr_logit_full = lr_generic_O ? ar.take<float>((size_t)B * O * hw) : nullptr;
    const bool gram_mode = lr && lr->gram && need_feat;
    if (gram_mode && feat_dtype != HALO_F64) return fail(HALO_E_UNSUPPORTED, "halo_score_maps_lr_gram: float64 embeddings only");
    double *gram = gram_mode ? ar.take<double>((size_t)B * GRAM_MAPS * lr->hf * lr->wf) : nullptr;
    if (!ar.ok()) return fail(HALO_E_WORKSPACE, "halo_score_maps: workspace too small");

    const bool f64out = (pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM) && feat_dtype == HALO_F64;
    const bool hist = pur_type == HALO_PUR_RIPU || pur_type == HALO_PUR_ORACLE_RIPU || pur_type == HALO_PUR_HYPER;
    const double ks = sqrt(fabs(-c) + 1e-15), rks = 1.0 / ks;
    dim3 block(TPB);

    // ---- logits -> ent (+ pred for ripu / oracle_ripu); fused into the feature stream when possible
    const bool ent_only = (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_PIXEL_ENTROPY) &&
                          pur_type != HALO_PUR_RIPU && pur_type != HALO_PUR_ORACLE_RIPU;
    const int fvec = !need_feat ? 0 : (feat_dtype == HALO_F64
        ? (((hw % 2 == 0) && (feat_bstride % 2 == 0) && aligned16(feat) && aligned16(imp_raw)) ? 2 : 1)
        : (((hw % 4 == 0) && (feat_bstride % 4 == 0) && aligned16(feat) && aligned16(imp_raw)) ? 4 : 1));
    const bool fuse = !lr && need_feat && ent_only && (O == 19 || O == 16) && fvec > 1 && (logit_bstride % fvec == 0) &&
                      (((uintptr_t)logit) % (4 * fvec) == 0) && getenv("HALO_NO_FUSE") == nullptr;
    FusedLogit fl{logit, (long long)logit_bstride, (int)O, unc_type, ent};
    const FusedLogit *flp = fuse ? &fl : nullptr;
    const bool need_logit_pass = !fuse && ( unc_type != HALO_UNC_ZEROS || pur_type == HALO_PUR_RIPU || pur_type == HALO_PUR_ORACLE_RIPU);
    short *pred_from_logits = (pur_type == HALO_PUR_RIPU || pur_type == HALO_PUR_ORACLE_RIPU) ? pred : nullptr;
    if (ev_logit_start) (void)hipEventRecord((hipEvent_t)ev_logit_start, st);
    if (need_logit_pass && lr) {
        const float shl = H > 1 ? (float)(lr->hl - 1) / (float)(H - 1) : 0.0f, swl = W > 1 ? (float)(lr->wl - 1) / (float)(W - 1) : 0.0f;
        dim3 grid((unsigned)nblk1, (unsigned)B);
        if (O == 19)
            hipLaunchKernelGGL((k_logit_maps_lr<19>), grid, block, 0, st, logit, (long long)logit_bstride, lr->hl, lr->wl, (int)H, (int)W, shl, swl, (const long long *)gt, unc_type, pur_type, ent, pred_from_logits);
        else if (O == 16)
            hipLaunchKernelGGL((k_logit_maps_lr<16>), grid, block, 0, st, logit, (long long)logit_bstride, lr->hl, lr->wl, (int)H, (int)W, shl, swl, (const long long *)gt, unc_type, pur_type, ent, pred_from_logits);
        else {
            hipLaunchKernelGGL(k_logit_interp_lr, grid, block, 0, st, logit, (long long)logit_bstride, (int)O, lr->hl, lr->wl, (int)H, (int)W, shl, swl, lr_logit_full);
            hipLaunchKernelGGL(k_logit_maps_generic, grid, block, 0, st, (const float *)lr_logit_full, (long long)(O * hw), (const long long *)gt, (int)O, hw, unc_type, pur_type, 0, ent, pred_from_logits);
        }
    } else if (need_logit_pass) {
        const bool vec4 = (hw % 4 == 0) && (logit_bstride % 4 == 0) && aligned16(logit) && aligned16(ent);
        if (O == 19 && vec4) {
            dim3 grid((unsigned)cdiv(hw, TPB * 4), (unsigned)B);
            hipLaunchKernelGGL((k_logit_maps<19, 4>), grid, block, 0, st, logit, (long long)logit_bstride, (const long long *)gt, hw, unc_type, pur_type, ent, pred_from_logits);
        } else if (O == 16 && vec4) {
            dim3 grid((unsigned)cdiv(hw, TPB * 4), (unsigned)B);
            hipLaunchKernelGGL((k_logit_maps<16, 4>), grid, block, 0, st, logit, (long long)logit_bstride, (const long long *)gt, hw, unc_type, pur_type, ent, pred_from_logits);
        } else {
            dim3 grid((unsigned)nblk1, (unsigned)B);
            hipLaunchKernelGGL(k_logit_maps_generic, grid, block, 0, st, logit, (long long)logit_bstride, (const long long *)gt, (int)O, hw, unc_type, pur_type, 0, ent, pred_from_logits);
        }
    } else if (!fuse) {
        hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)cdiv(B * hw, TPB)), block, 0, st, ent, (long long)(B * hw), 0.0f);
    }

    if (ev_logit_stop) (void)hipEventRecord((hipEvent_t)ev_logit_stop, st);

    // ---- features -> radius / norm  (+ min/max partials)
    int nblk_imp = nblk1;
    if (need_feat) {
        if (ev_feat_start) (void)hipEventRecord((hipEvent_t)ev_feat_start, st);
        const int mode = pur_type == HALO_PUR_EUC_NORM ? 1 : 0;
        if (gram_mode) {
            // 16 bytes per lane where the planes allow it (even width, 16-byte aligned); HALO_GRAM_8B=1: the 8-byte kernel (A/B)
            const bool wide = lr->wf >= 2 && lr->wf % 2 == 0 && feat_bstride % 2 == 0 && aligned16(feat) && aligned16(gram) && getenv("HALO_GRAM_8B") == nullptr;
            if (wide) {
                // rows per wave: every source row is loaded (R + 1) / R times (its own strip + as lower neighbour of the strip above),
                // and that redundancy, not the bytes in flight, is what the kernel's time follows (profiles/archive/r04_gram_ab.txt: R = 2 / 4 /
                // 8 -> 788 / 745 / 705 us per 16 images; 2, 3 or 4 channels in flight: equal; non-temporal loads: 20 % slower).  The
                // widest strip that still gives every SIMD of the chip a wave; HALO_GRAM_ROWS / _UCH / _NT are A/B switches (same bits)
                const char *eu = getenv("HALO_GRAM_UCH"), *en = getenv("HALO_GRAM_NT"), *er = getenv("HALO_GRAM_ROWS");
                const long long nwx = cdiv(lr->wf, 128);
                int rows = 2;
                for (int cand = 8; cand > 2; cand >>= 1)
                    if (B * nwx * cdiv(lr->hf, cand) >= 1024) { rows = cand; break; }
                if (er) rows = atoi(er);
                rows = rows >= 8 ? 8 : (rows >= 4 ? 4 : 2);
                const int uch = eu ? atoi(eu) : (rows == 8 ? 1 : 2), nt = en ? atoi(en) : 0;
                const long long nwaves = nwx * cdiv(lr->hf, rows);
                const dim3 gg((unsigned)(cdiv(cdiv(nwaves, TPB / 64), 8) * 8), (unsigned)B);
#define HALO_GRAM2(U_, N_, R_) hipLaunchKernelGGL((k_gram_lr2<U_, N_, R_>), gg, block, 0, st, (const double *)feat, (long long)feat_bstride, (int)C, lr->hf, lr->wf, gram)
                if (rows == 8) HALO_GRAM2(1, false, 8);
                else if (rows == 4) { if (uch >= 3) HALO_GRAM2(3, false, 4); else if (uch == 2) HALO_GRAM2(2, false, 4); else HALO_GRAM2(1, false, 4); }
                else if (uch >= 4) { if (nt) HALO_GRAM2(4, true, 2); else HALO_GRAM2(4, false, 2); }
                else { if (nt) HALO_GRAM2(2, true, 2); else HALO_GRAM2(2, false, 2); }
#undef HALO_GRAM2
            } else {
                const long long nwaves = cdiv(lr->wf, GRAM_COLS) * cdiv(lr->hf, 2);
                hipLaunchKernelGGL(k_gram_lr, dim3((unsigned)(cdiv(cdiv(nwaves, TPB / 64), 8) * 8), (unsigned)B), block, 0, st, (const double *)feat,
                                   (long long)feat_bstride, (int)C, lr->hf, lr->wf, gram);
            }
            if (ev_feat_mid) (void)hipEventRecord((hipEvent_t)ev_feat_mid, st);       // between the Gram pass and the radius pass
            const double shd = H > 1 ? (double)(lr->hf - 1) / (double)(H - 1) : 0.0, swd = W > 1 ? (double)(lr->wf - 1) / (double)(W - 1) : 0.0;
            nblk_imp = nblk1;
            if (mode == 0) hipLaunchKernelGGL(k_radius_gram<0>, dim3((unsigned)nblk1, (unsigned)B), block, 0, st, gram, (const double *)feat, (long long)feat_bstride, (int)C, lr->hf, lr->wf, (int)H, (int)W, shd, swd, ks, rks, imp_raw, part_imp);
            else hipLaunchKernelGGL(k_radius_gram<1>, dim3((unsigned)nblk1, (unsigned)B), block, 0, st, gram, (const double *)feat, (long long)feat_bstride, (int)C, lr->hf, lr->wf, (int)H, (int)W, shd, swd, ks, rks, imp_raw, part_imp);
        } else if (lr) {
            const int rc = feat_dtype == HALO_F64
                ? launch_feat_lr<double>((const double *)feat, feat_bstride, (int)C, *lr, (int)H, (int)W, (int)B, mode, ks, rks, imp_raw, part_imp, nblk_imp, st)
                : launch_feat_lr<float>((const float *)feat, feat_bstride, (int)C, *lr, (int)H, (int)W, (int)B, mode, ks, rks, (float *)imp_raw, part_imp, nblk_imp, st);
            if (rc != HALO_OK) return rc;
        } else if (feat_dtype == HALO_F64) {
            const bool v2 = (hw % 2 == 0) && (feat_bstride % 2 == 0) && aligned16(feat) && aligned16(imp_raw);
            if (v2) { nblk_imp = (int)cdiv(hw, FTPB * 2); launch_feat<double, 2>((const double *)feat, feat_bstride, (int)C, hw, (int)B, mode, ks, rks, imp_raw, part_imp, nblk_imp, st, flp); }
            else { nblk_imp = (int)cdiv(hw, FTPB); launch_feat<double, 1>((const double *)feat, feat_bstride, (int)C, hw, (int)B, mode, ks, rks, imp_raw, part_imp, nblk_imp, st); }
        } else {
            const bool v4 = (hw % 4 == 0) && (feat_bstride % 4 == 0) && aligned16(feat) && aligned16(imp_raw);
            if (v4) { nblk_imp = (int)cdiv(hw, FTPB * 4); launch_feat<float, 4>((const float *)feat, feat_bstride, (int)C, hw, (int)B, mode, ks, rks, (float *)imp_raw, part_imp, nblk_imp, st, flp); }
            else { nblk_imp = (int)cdiv(hw, FTPB); launch_feat<float, 1>((const float *)feat, feat_bstride, (int)C, hw, (int)B, mode, ks, rks, (float *)imp_raw, part_imp, nblk_imp, st); }
        }
        if (ev_feat_stop) (void)hipEventRecord((hipEvent_t)ev_feat_stop, st);
    }
    // everything behind the passes over the inputs may run on another stream of the caller (halo_score_maps_split): the small
    // kernels of the tail then overlap the NEXT call's feature pass instead of standing between two of them
    if (tail_stream && (hipStream_t)tail_stream != st) {
        if (!need_feat || !ev_feat_stop) return fail(HALO_E_ARG, "halo_score_maps_split: a purity type that reads decoder_out and ev_feat_stop are required with a tail stream");
        if (hipStreamWaitEvent((hipStream_t)tail_stream, (hipEvent_t)ev_feat_stop, 0) != hipSuccess)
            return fail(HALO_E_LAUNCH, "halo_score_maps_split: hipStreamWaitEvent: %s", hipGetErrorString(hipGetLastError()));
        st = (hipStream_t)tail_stream;
    }
    dim3 grid1((unsigned)nblk1, (unsigned)B);
    if (pur_type == HALO_PUR_HYPER) {
        hipLaunchKernelGGL(k_minmax_finalize, dim3((unsigned)B), block, 0, st, part_imp, nblk_imp, stats, 0);
        if (feat_dtype == HALO_F64) hipLaunchKernelGGL((k_quantize<double, short>), grid1, block, 0, st, (const double *)imp_raw, stats, hw, (int)K, pred);
        else hipLaunchKernelGGL((k_quantize<float, short>), grid1, block, 0, st, (const float *)imp_raw, stats, hw, (int)K, pred);
    }
    // the impurity's min / max are only needed by normalize_map: per-block partials straight from the 3x3 histogram kernel,
    // a separate pass for the other sources of the map, nothing at all when the branch does not normalise (ripu.yaml)
    if (hist) {
        const float logK = (float)log((double)(pur_type == HALO_PUR_HYPER ? K : O));
        const int np = launch_region_impurity<short>((const short *)pred, B, H, W, pksize, logK, (float *)imp_raw, (float *)nullptr, st,
                                                     normalize ? part_imp : nullptr, pad);
        nblk_imp = np ? np : nblk1;
        if (normalize && !np) hipLaunchKernelGGL(k_minmax_f32, grid1, block, 0, st, (const float *)imp_raw, hw, part_imp);
    } else if (pur_type == HALO_PUR_NONE) {
        hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)cdiv(B * hw, TPB)), block, 0, st, (float *)imp_raw, (long long)(B * hw), 0.0f);
        if (normalize) hipLaunchKernelGGL(k_minmax_f32, grid1, block, 0, st, (const float *)imp_raw, hw, part_imp);
        nblk_imp = nblk1;
    }

    // ---- box-sum of the uncertainty, / count
    const int do_box = (unc_type == HALO_UNC_ENTROPY || unc_type == HALO_UNC_ORACLE_ACC) ? 1 : 0;
    int nblk_unc = nblk1;
    const bool box3 = do_box && pad == HALO_PAD_ZEROS && ksize == 3 && W % 4 == 0 && aligned16(ent) && aligned16(unc_raw);   // (the 3x3 fast paths pad with zeros)
    // 3 x 3 window + 16-byte aligned maps: the box sum is recomputed inside the combine kernel (no stored copy);
    // HALO_NO_FUSE_TAIL=1 keeps the round-2 sequence (A/B switch, identical results)
    const bool fuse_tail = box3 && B <= 65535 && H <= 65535 && aligned16(imp_raw) && aligned16(score) && (!impurity || aligned16(impurity)) &&
                           (!uncertainty || aligned16(uncertainty)) && (!active || ((uintptr_t)active & 3) == 0) &&
                           getenv("HALO_NO_FUSE_TAIL") == nullptr;
    const int nblk_c3 = (int)cdiv(hw, TPB * 4);
    // the selector's coarse histogram of a normalised score map, counted by the combine kernel while it writes the map and handed
    // over behind the range records (HALO_NO_FUSE_HIST=1: A/B switch, the selector then counts it itself -- same picks)
    unsigned *rng_hist = (score_range && normalize && fuse_tail && getenv("HALO_NO_FUSE_HIST") == nullptr)
                             ? (unsigned *)((char *)score_range + range_hist_offset(B)) : nullptr;
    if (fuse_tail) {
        if (normalize) {            // only the min / max are needed before the combine kernel
            const dim3 gridm((unsigned)cdiv(W, BM_TW), (unsigned)cdiv(H, BM_TH), (unsigned)B);
            nblk_unc = (int)(gridm.x * gridm.y);
            hipLaunchKernelGGL(k_box3_minmax, gridm, block, 0, st, (const float *)ent, (int)H, (int)W, hist ? pksize : 0, part_unc, rng_hist);
        }
    } else if (box3) {
        nblk_unc = nblk_c3;
        hipLaunchKernelGGL(k_box3_unc, dim3((unsigned)nblk_unc, (unsigned)B), block, 0, st, (const float *)ent, (int)H, (int)W,
                           hist ? pksize : 0, unc_raw, normalize ? part_unc : nullptr);
    } else {
        hipLaunchKernelGGL(k_box_unc, grid1, block, 0, st, ent, (int)H, (int)W, ksize, do_box, hist ? pksize : 0, unc_raw,
                           normalize ? part_unc : nullptr, pad);
    }

    // ---- global min/max (normalize_map only), then normalise + product
    // the score's value range for the selector: free when the maps are normalised and the fused combine kernel runs (the
    // product of two values in [0, 1]); otherwise the exact reduction, here instead of in the selector
    SelHdr *rng_free = (score_range && normalize && fuse_tail) ? (SelHdr *)score_range : nullptr;
    if (normalize)
        hipLaunchKernelGGL(k_minmax_finalize2, dim3((unsigned)B, 2u), dim3(FIN_TPB), 0, st, (const double *)part_imp, nblk_imp,
                           (const double *)part_unc, nblk_unc, stats);
    if (fuse_tail) {
        int crows = CB_ROWS;                 // rows per workgroup: as many as leave >= 1024 workgroups in the launch
        while (crows > 1 && cdiv(W, TPB * 4) * cdiv(H, crows) * B < 1024) crows >>= 1;
        dim3 gridc((unsigned)cdiv(W, TPB * 4), (unsigned)cdiv(H, crows), (unsigned)B);
        // non-temporal impurity / uncertainty stores measured SLOWER (233 against 214 us per 16 images, gpurun_out/r05f): off unless asked for
        static const bool cb_nt = getenv("HALO_COMBINE_NT") != nullptr;      // A/B switch (same bits)
#define HALO_CB(T, NT_) hipLaunchKernelGGL((k_combine_box3<T, NT_>), gridc, block, 0, st, (const T *)imp_raw, (const float *)ent, stats, active, (int)H, (int)W, hist ? pksize : 0, normalize, (T *)score, (T *)impurity, uncertainty, rng_free, rng_hist, crows)
        if (f64out) { if (cb_nt) HALO_CB(double, true); else HALO_CB(double, false); }
        else { if (cb_nt) HALO_CB(float, true); else HALO_CB(float, false); }
#undef HALO_CB
    } else if (f64out) hipLaunchKernelGGL((k_combine<double>), grid1, block, 0, st, (const double *)imp_raw, unc_raw, stats, active, hw, normalize, (double *)score, (double *)impurity, uncertainty);
    else hipLaunchKernelGGL((k_combine<float>), grid1, block, 0, st, (const float *)imp_raw, unc_raw, stats, active, hw, normalize, (float *)score, (float *)impurity, uncertainty);
    if (score_range && !rng_free) {
        const int rc = score_range_exact(score, f64out ? HALO_F64 : HALO_F32, B, hw, score_range, st);
        if (rc != HALO_OK) return rc;
    }
    if (ev_tail_stop) (void)hipEventRecord((hipEvent_t)ev_tail_stop, st);
    return check_launch("halo_score_maps");
}

// ---------------------------------------------------------------- helper-method entry points
// FloatingRegionScore.compute_region_uncertainty / compute_pixel_entropy (floating_region.py:70-92,123-127)
extern "C" int halo_region_uncertainty(const float *x, int64_t bstride, int is_prob, const int64_t *gt, int64_t B, int64_t O,
                                       int64_t H, int64_t W, int unc_type, int ksize, int do_box, float *out,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int pad = (do_box >> 8) & 3;                    // do_box: bit 0 = box-sum the map, bits 8-9 = padding mode (halo_hip.h)
    do_box &= 1;
    if (!x || !out || B <= 0 || O <= 0 || H <= 0 || W <= 0) return fail(HALO_E_ARG, "halo_region_uncertainty: null/empty argument");
    if (unc_type < 0 || unc_type > HALO_UNC_ZEROS) return fail(HALO_E_ARG, "halo_region_uncertainty: bad unc_type");
    if (unc_type == HALO_UNC_ORACLE_ACC && !gt) return fail(HALO_E_ARG, "halo_region_uncertainty: ground_truth required");
    if (ksize < 1 || !(ksize & 1)) return fail(HALO_E_ARG, "halo_region_uncertainty: window size must be odd");
    if (do_box) { const int rc = check_padding(pad, ksize, H, W, "halo_region_uncertainty"); if (rc != HALO_OK) return rc; }
    const long long hw = (long long)H * W;
    if (!workspace || workspace_bytes < (size_t)B * hw * 4 + 256) return fail(HALO_E_WORKSPACE, "halo_region_uncertainty: workspace too small");
    float *ent = (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    dim3 block(TPB), grid1((unsigned)cdiv(hw, TPB), (unsigned)B);
    float *dst = do_box ? ent : out;
    if (unc_type == HALO_UNC_ZEROS)
        hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)cdiv(B * hw, TPB)), block, 0, st, dst, (long long)(B * hw), 0.0f);
    else
        hipLaunchKernelGGL(k_logit_maps_generic, grid1, block, 0, st, x, (long long)bstride, (const long long *)gt, (int)O, hw,
                           unc_type, HALO_PUR_NONE, is_prob, dst, (short *)nullptr);
    if (do_box)
        hipLaunchKernelGGL(k_box_unc, grid1, block, 0, st, (const float *)ent, (int)H, (int)W, ksize, 1, 0, out, (double *)nullptr, pad);
    return check_launch("halo_region_uncertainty");
}

// FloatingRegionScore.compute_region_impurity(predict, K) (floating_region.py:112-121)
extern "C" int halo_region_impurity(const int64_t *pred, int64_t B, int64_t H, int64_t W, int ksize, int64_t K, float *impurity,
                                    float *count, int pad_mode, void *stream)
{
    if (!pred || !impurity || B <= 0 || H <= 0 || W <= 0 || K < 1) return fail(HALO_E_ARG, "halo_region_impurity: null/empty argument");
    if (ksize < 1 || !(ksize & 1)) return fail(HALO_E_ARG, "halo_region_impurity: window size must be odd");
    { const int rc = check_padding(pad_mode, ksize, H, W, "halo_region_impurity"); if (rc != HALO_OK) return rc; }
    launch_region_impurity<long long>((const long long *)pred, B, H, W, ksize, (float)log((double)K), impurity, count, (hipStream_t)stream,
                                      nullptr, pad_mode);
    return check_launch("halo_region_impurity");
}

// FloatingRegionScore.quantize_uncert_map(decoder_out) (floating_region.py:94-110) -> int64 bins
extern "C" int halo_quantize_radius(const void *feat, int feat_dtype, int64_t feat_bstride, int64_t B, int64_t C, int64_t H,
                                    int64_t W, int64_t K, double c, int64_t *pred, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!feat || !pred || B <= 0 || C <= 0 || H <= 0 || W <= 0 || K < 1) return fail(HALO_E_ARG, "halo_quantize_radius: null/empty argument");
    if (feat_dtype != HALO_F32 && feat_dtype != HALO_F64) return fail(HALO_E_ARG, "halo_quantize_radius: bad dtype");
    if (!workspace || workspace_bytes < halo_score_workspace_bytes(B, H, W)) return fail(HALO_E_WORKSPACE, "halo_quantize_radius: workspace too small");
    const long long hw = (long long)H * W;
    const int nblk1 = (int)cdiv(hw, TPB), nblkf = (int)cdiv(hw, FTPB);
    Arena ar(workspace, workspace_bytes);
    double *imp_raw = ar.take<double>((size_t)B * hw);
    double *part = ar.take<double>((size_t)B * nblkf * 2);
    double *stats = ar.take<double>((size_t)B * 4);
    if (!ar.ok()) return fail(HALO_E_WORKSPACE, "halo_quantize_radius: workspace too small");
    const double ks = sqrt(fabs(-c) + 1e-15), rks = 1.0 / ks;
    dim3 block(TPB), grid1((unsigned)nblk1, (unsigned)B);
    if (feat_dtype == HALO_F64) {
        launch_feat<double, 1>((const double *)feat, feat_bstride, (int)C, hw, (int)B, 0, ks, rks, imp_raw, part, nblkf, st);
        hipLaunchKernelGGL(k_minmax_finalize, dim3((unsigned)B), block, 0, st, (const double *)part, nblkf, stats, 0);
        hipLaunchKernelGGL((k_quantize<double, long long>), grid1, block, 0, st, (const double *)imp_raw, (const double *)stats, hw, (int)K, (long long *)pred);
    } else {
        launch_feat<float, 1>((const float *)feat, feat_bstride, (int)C, hw, (int)B, 0, ks, rks, (float *)imp_raw, part, nblkf, st);
        hipLaunchKernelGGL(k_minmax_finalize, dim3((unsigned)B), block, 0, st, (const double *)part, nblkf, stats, 0);
        hipLaunchKernelGGL((k_quantize<float, long long>), grid1, block, 0, st, (const float *)imp_raw, (const double *)stats, hw, (int)K, (long long *)pred);
    }
    return check_launch("halo_quantize_radius");
}

extern "C" int halo_score_maps_timed(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                                     int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                                     int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                                     int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                                     void *workspace, size_t workspace_bytes, void *stream, void *ev_feat_start,
                                     void *ev_feat_stop, void *score_range)
{
    return score_impl(logit, logit_bstride, feat, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type, pur_type,
                      normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace, workspace_bytes, stream,
                      ev_feat_start, ev_feat_stop, nullptr, nullptr, nullptr, score_range);
}

// halo_score_maps_timed with the tail (min / max, normalisation, product, mask: everything behind the passes over logit and
// decoder_out) enqueued on `tail_stream`, which first waits for ev_feat_stop (recorded on `stream` behind the feature pass).
// The outputs are complete on tail_stream; the workspace belongs to the call until then.
extern "C" int halo_score_maps_split(const float *logit, int64_t logit_bstride, const void *feat, int feat_dtype,
                                     int64_t feat_bstride, const int64_t *gt, const uint8_t *active, int64_t B, int64_t O,
                                     int64_t C, int64_t H, int64_t W, int unc_type, int pur_type, int normalize, int ksize,
                                     int pksize, int64_t K, double c, void *score, void *impurity, float *uncertainty,
                                     void *workspace, size_t workspace_bytes, void *stream, void *tail_stream, void *ev_feat_start,
                                     void *ev_feat_stop, void *score_range)
{
    return score_impl(logit, logit_bstride, feat, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type, pur_type,
                      normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace, workspace_bytes, stream,
                      ev_feat_start, ev_feat_stop, nullptr, nullptr, nullptr, score_range, tail_stream);
}

// FloatingRegionScore.forward on bilinearly upsampled (align_corners=True) low-resolution sources
// without materialising them: logit_lr (B,O,hl,wl) f32, feat_lr (B,C,hf,wf) f64|f32 -> maps (B,H,W).
// = core/active/build.py:122-144 for B images (resize of output and decoder_out + the scorer).
extern "C" int halo_score_maps_lr(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                                  int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                                  const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                                  int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                                  void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream)
{
    if (hl <= 0 || wl <= 0) return fail(HALO_E_ARG, "halo_score_maps_lr: bad low-res logit size");
    const bool need_feat = pur_type == HALO_PUR_HYPER || pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM;
    if (need_feat && (hf <= 0 || wf <= 0)) return fail(HALO_E_ARG, "halo_score_maps_lr: bad low-res embedding size");
    LrDims lr{(int)hl, (int)wl, (int)hf, (int)wf, false};
    return score_impl(logit_lr, logit_bstride, feat_lr, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type, pur_type,
                      normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace, workspace_bytes, stream, nullptr,
                      nullptr, &lr);
}

extern "C" int halo_score_maps_lr_gram(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                                       int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                                       const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                                       int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                                       void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream)
{
    if (hl <= 0 || wl <= 0) return fail(HALO_E_ARG, "halo_score_maps_lr_gram: bad low-res logit size");
    const bool need_feat = pur_type == HALO_PUR_HYPER || pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM;
    if (need_feat && (hf <= 0 || wf <= 0)) return fail(HALO_E_ARG, "halo_score_maps_lr_gram: bad low-res embedding size");
    if (need_feat && hf * wf > 0x7fffffffll) return fail(HALO_E_UNSUPPORTED, "halo_score_maps_lr_gram: low-res planes of more than 2^31 elements");
    if (need_feat && workspace_bytes < halo_score_lr_gram_workspace_bytes(B, O, H, W, hf, wf))
        return fail(HALO_E_WORKSPACE, "halo_score_maps_lr_gram: workspace too small");
    LrDims lr{(int)hl, (int)wl, (int)hf, (int)wf, true};
    return score_impl(logit_lr, logit_bstride, feat_lr, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type, pur_type,
                      normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace, workspace_bytes, stream, nullptr,
                      nullptr, &lr);
}

// halo_score_maps_lr (gram = 0) / halo_score_maps_lr_gram (gram = 1) with optional hipEvent_t handles (halo_event_create) recorded on
// `stream` around the logit pass (k_logit_maps_lr), around the embedding pass (k_feat_reduce_lr, or k_gram_lr + k_radius_gram, with
// ev_feat_mid between the two) and behind the tail (ev_tail_stop) -- bench.py's live per-kernel times for the low-resolution boundary.
extern "C" int halo_score_maps_lr_timed(const float *logit_lr, int64_t logit_bstride, int64_t hl, int64_t wl, const void *feat_lr,
                                        int feat_dtype, int64_t feat_bstride, int64_t hf, int64_t wf, const int64_t *gt,
                                        const uint8_t *active, int64_t B, int64_t O, int64_t C, int64_t H, int64_t W, int unc_type,
                                        int pur_type, int normalize, int ksize, int pksize, int64_t K, double c, void *score,
                                        void *impurity, float *uncertainty, void *workspace, size_t workspace_bytes, void *stream,
                                        int gram, void *ev_logit_start, void *ev_logit_stop, void *ev_feat_start, void *ev_feat_stop,
                                        void *score_range, void *ev_feat_mid, void *ev_tail_stop)
{
    if (hl <= 0 || wl <= 0) return fail(HALO_E_ARG, "halo_score_maps_lr_timed: bad low-res logit size");
    const bool need_feat = pur_type == HALO_PUR_HYPER || pur_type == HALO_PUR_RADIUS || pur_type == HALO_PUR_EUC_NORM;
    if (need_feat && (hf <= 0 || wf <= 0)) return fail(HALO_E_ARG, "halo_score_maps_lr_timed: bad low-res embedding size");
    if (gram && need_feat && workspace_bytes < halo_score_lr_gram_workspace_bytes(B, O, H, W, hf, wf))
        return fail(HALO_E_WORKSPACE, "halo_score_maps_lr_timed: workspace too small");
    LrDims lr{(int)hl, (int)wl, (int)hf, (int)wf, gram != 0};
    return score_impl(logit_lr, logit_bstride, feat_lr, feat_dtype, feat_bstride, gt, active, B, O, C, H, W, unc_type, pur_type,
                      normalize, ksize, pksize, K, c, score, impurity, uncertainty, workspace, workspace_bytes, stream, ev_feat_start,
                      ev_feat_stop, &lr, ev_logit_start, ev_logit_stop, score_range, nullptr, ev_feat_mid, ev_tail_stop);
}
