// select_pixels_to_label (core/active/build.py:27-64) without the per-pick memory round trip:
// the value-binned sweep.
//
// The reference picks, n times, the global arg-max of the score map and suppresses a
// (2*mask_radius+1)^2 window around it.  That sequence is the prefix, in priority order, of the
// lexicographically-first maximal independent set of the "within one window" relation: a pixel is
// picked iff no pixel of HIGHER priority that was itself picked lies within its window.  Whether a
// pixel is picked therefore depends only on higher-priority pixels, so the picks can be found by
// visiting pixels in descending priority and testing each against the picks made so far -- no
// arg-max over the map, no dependent HBM access per pick.  Two facts bound the work:
//   * the n-th pick has at most (n-1)*(2r+1)^2 pixels above it (each one is a pick or inside the
//     window of an earlier pick), so only the top K = n*(2r+1)^2 values are ever visited;
//   * two picks are more than r apart (Chebyshev), so a cell of (r+1)^2 pixels holds at most one:
//     "the picks so far" is a byte per cell in LDS and a test reads the 3 x 3 cells around a pixel.
//
// Pipeline (all asynchronous on one stream, no host synchronisation):
//   k_sel_range    min / max of the finite values, count of pickable pixels
//   k_sel_hist1    2048 equal-width coarse bins over [min, max]                      (LDS histograms)
//   k_sel_scan1    threshold bin t1 (top-K), every coarse bin split into count/target equal sub-bins
//   k_sel_place    pixels >= t1 -> a slot of their fine bin (one returned atomic per candidate; bins have BIN_CAP slots
//                  and are numbered in descending value order) -- round 2 went through a staging list, a fine histogram,
//                  a prefix scan and a scatter pass
//   k_sel_sweep    ONE workgroup per image walks the bins: filter a bin's candidates against the
//                  pick grid (4 waves, one candidate per lane), then wave 0 takes the survivors
//                  in exact (value, w, h) order with a register-resident arg-max loop
//   k_sel_apply    one wave per pick writes its (h, w, value) row of the pick table and its windows
//                  (score = -inf, active, selected, active_mask)
// Order inside a bin never matters (the resolve step is an exact arg-max over the bin's survivors),
// bins are monotone in the value, so the result is the reference's sequence bit for bit.
//
// The sweep gives up ("bail") where its assumptions do not hold -- NaN or +inf in the map, a
// constant map, a fine bin with more candidates than slots or more than 256 unsuppressed ones (plateaus of exact ties),
// or candidates exhausted after the threshold bin had to be dropped -- and leaves the image, in the
// exact state the reference would have after `np` picks, to the serial kernel of halo_select.hip,
// which is enqueued behind it and returns immediately for images that are done.
#include <stdlib.h>
#include <string.h>

#include "halo_select_plan.hpp"

namespace halo {

constexpr int SW_TPB = 256;      // sweep workgroup: 4 waves, one per SIMD
constexpr int SW_SURV = 256;     // survivor capacity of one bin (4 per lane of the resolving wave)
constexpr int FWIN = 1024;       // fine-bin offsets staged in LDS at a time

// per-image arrays: element [b * stride + i]
constexpr int BIN_CAP = 128;     // slots per fine bin (expected occupancy <= target = 64: equal-width sub-bins of a 1/2048 slice of
                                 // the value range are Poisson-filled; a fuller bin -- a plateau of ties -- hands the image over)
constexpr int PL_U = 8;          // row segments (loads, then returned atomics) a k_sel_place thread keeps in flight
constexpr int SW_MB = 4;         // the sweep filters up to SW_MB consecutive bins (<= 256 candidates together) in one pass

struct BinWs {
    SelHdr *hdr;
    SelHdr *rng;                             // where the value range lives: hdr, or the scorer's range record (halo_score_range_t)
    const unsigned *rng_hist;                // the scorer's coarse histograms behind its records (valid per image: SEL_F_HIST), or NULL
    unsigned *hist1, *cbase, *cm;            // NB1 each
    unsigned *fcur;                          // nfmax: candidates placed in each fine bin
    unsigned long long *ckey;                // nfmax * BIN_CAP: bin f owns slots [f * BIN_CAP, (f + 1) * BIN_CAP)
    unsigned *cpos;                          // nfmax * BIN_CAP
    unsigned long long *okmin_inv, *okmax;   // nfmax each: max of ~key / of key over the candidates of the bins that ran out of slots (zero = none)
    unsigned *plist;                         // n_regions: picks as (w << 16) | h
    int *handover;                           // optional (B, 2): {HALO_SWEEP_* reason, picks the sweep made}; NULL = not reported
};

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ------------------------------------------------------------------ counters to zero
// A kernel, not hipMemsetAsync: inside a captured hipGraph (ROCm 7.2) a memset node between kernel nodes was observed not to be
// ordered against them (the histogram kernels of the same replay found counters of the previous one: different picks in 1 of ~3
// replays of a score + select graph, tests/test_gpu_parity.py::test_score_and_select_replay_from_a_hip_graph); a kernel node is.
__global__ void __launch_bounds__(256) k_sel_zero(uint4 *__restrict__ p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------ value range
template <typename T>
__global__ void __launch_bounds__(256) k_sel_range(const T *__restrict__ score, long long hw, BinWs ws)
{
    const int b = blockIdx.y;
    const T *sc = score + (size_t)b * hw;
    unsigned long long kmin_inv = 0, kmax = 0;
    unsigned nval = 0, bad = 0;
    // four independent loads per iteration: beside a bandwidth-bound kernel a round trip takes microseconds, and a thread's
    // iterations would otherwise pay them one after the other
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw; i += 4 * stride) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * stride < hw ? (double)sc[i + u * stride] : __longlong_as_double(0xfff0000000000000ll);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long k = order_key(v[u]);
            const bool isbad = k >= KEY_POS_INF;                         // +inf or NaN
            const bool ok = !isbad && k != KEY_NEG_INF;
            bad |= isbad ? 1u : 0u;
            nval += ok ? 1u : 0u;
            const unsigned long long ki = ok ? ~k : 0ull, kx = ok ? k : 0ull;
            kmin_inv = ki > kmin_inv ? ki : kmin_inv;
            kmax = kx > kmax ? kx : kmax;
        }
    }
    kmin_inv = wave_max_u64(kmin_inv);
    kmax = wave_max_u64(kmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nval += __shfl_xor(nval, o, 64); bad |= __shfl_xor(bad, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        SelHdr *h = ws.hdr + b;
        if (nval) {
            atomicMax(&h->kmin_inv, kmin_inv);
            atomicMax(&h->kmax, kmax);
            atomicAdd(&h->nvalid, nval);
        }
        if (bad) atomicOr(&h->flags, (unsigned)SEL_F_BAD);
    }
}

// ------------------------------------------------------------------ coarse histogram
template <typename T>
__global__ void __launch_bounds__(256) k_sel_hist1(const T *__restrict__ score, long long hw, BinWs ws)
{
    __shared__ unsigned h[NB1];
    const int b = blockIdx.y;
    const ValRange r = sel_range(ws.rng[b]);
    if (!r.ok) return;
    if (ws.rng_hist && (ws.rng[b].flags & SEL_F_HIST)) return;      // the scorer counted this map while it wrote it
    for (int j = threadIdx.x; j < NB1; j += 256) h[j] = 0;
    __syncthreads();
    const T *sc = score + (size_t)b * hw;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < hw; i += PL_U * stride) {
        double v[PL_U];
#pragma unroll
        for (int u = 0; u < PL_U; ++u) v[u] = i + u * stride < hw ? (double)sc[i + u * stride] : __longlong_as_double(0xfff0000000000000ll);
#pragma unroll
        for (int u = 0; u < PL_U; ++u) {
            const unsigned long long k = order_key(v[u]);
            if (k < KEY_POS_INF && k != KEY_NEG_INF) {
                double t;
                atomicAdd(&h[coarse_bin(v[u], r, t)], 1u);
            }
        }
    }
    __syncthreads();
    unsigned *g = ws.hist1 + (size_t)b * NB1;
    for (int j = threadIdx.x; j < NB1; j += 256)
        if (h[j]) atomicAdd(&g[j], h[j]);
}

// ------------------------------------------------------------------ threshold + sub-bin layout
// One workgroup per image.  S[j] = number of values in coarse bins >= j.  t1 = the highest bin with
// S[t1] >= kneed (0 if even S[0] is smaller: every pickable pixel is a candidate).  If S[t1] exceeds
// the staging capacity the threshold bin is dropped (t1 + 1, `truncated`).  Coarse bin j is split into
// m[j] = ceil(count / target) equal-width sub-bins; fine bins are numbered from the TOP of the value
// range downwards, cbase[j] = sum of m over the bins above j.
__global__ void __launch_bounds__(256) k_sel_scan1(BinWs ws, BinGeom g)
{
    __shared__ unsigned sc_c[256], sc_m[256], s_t1;
    const int b = blockIdx.x, tid = threadIdx.x;
    SelHdr *hdr = ws.hdr + b;
    const ValRange r = sel_range(ws.rng[b]);
    // the scorer's histogram describes the map as it was when it was scored: used once (the flag is cleared below), and should the
    // map have changed since (fewer candidates above the threshold than the counts promise) an exhausted sweep hands the image over
    // instead of concluding that nothing is left (`truncated`)
    const bool ext = ws.rng_hist && (ws.rng[b].flags & SEL_F_HIST);
    const unsigned *hist = (ext ? ws.rng_hist : ws.hist1) + (size_t)b * NB1;
    unsigned c[8], m[8], tc = 0, tm = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c[i] = r.ok ? hist[tid * 8 + i] : 0u;
        m[i] = (c[i] + g.target - 1) / g.target;
        tc += c[i];
        tm += m[i];
    }
    sc_c[tid] = tc;
    sc_m[tid] = tm;
    if (tid == 0) s_t1 = 0;
    __syncthreads();
    // inclusive suffix scan over the 256 per-thread totals (Hillis-Steele)
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned ac = tid + d < 256 ? sc_c[tid + d] : 0u, am = tid + d < 256 ? sc_m[tid + d] : 0u;
        __syncthreads();
        sc_c[tid] += ac;
        sc_m[tid] += am;
        __syncthreads();
    }
    unsigned above_c = tid + 1 < 256 ? sc_c[tid + 1] : 0u, above_m = tid + 1 < 256 ? sc_m[tid + 1] : 0u;
    unsigned S[8], M[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        M[i] = above_m;                      // fine bins above coarse bin j
        above_c += c[i];
        above_m += m[i];
        S[i] = above_c;                      // values in bins >= j
    }
    int mine = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (S[i] >= g.kneed) mine = tid * 8 + i;
    if (mine > 0) atomicMax(&s_t1, (unsigned)mine);
    unsigned *cb = ws.cbase + (size_t)b * NB1, *cmm = ws.cm + (size_t)b * NB1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { cb[tid * 8 + i] = M[i]; cmm[tid * 8 + i] = m[i]; }
    __syncthreads();
    unsigned t1 = s_t1;
    // the thread owning bin t1 decides about truncation and publishes the layout
    if ((int)(t1 >> 3) == tid) {
        const int i = t1 & 7;
        unsigned trunc = 0, nf;
        unsigned s_here = 0, m_here = 0, M_here = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q == i) { s_here = S[q]; m_here = m[q]; M_here = M[q]; }
        if (!r.ok) { t1 = NB1; trunc = 1; nf = 0; }
        else if (s_here > g.captot) { t1 += 1; trunc = 1; nf = M_here; }
        else nf = M_here + m_here;
        if (ext && t1 > 0) trunc = 1;
        hdr->t1 = t1;
        hdr->truncated = trunc;
        hdr->nf = nf;
    }
    __syncthreads();
    if (ext && tid == 0) ws.rng[b].flags &= ~(unsigned)SEL_F_HIST;      // consumed
}

// ------------------------------------------------------------------ candidates -> their fine bins, in one pass
// Round 3: a candidate goes straight to a slot of its fine bin -- one returned atomic on the bin's counter, two stores --
// instead of staging list -> fine histogram -> prefix scan -> scatter (round 2: two atomics and a 16-byte round trip per
// candidate, three launches).  Bins have BIN_CAP slots; a candidate that finds its bin full marks the image SEL_F_OVERFLOW and
// the sweep hands it over untouched.
template <typename T>
__global__ void __launch_bounds__(256) k_sel_place(const T *__restrict__ score, BinWs ws, BinGeom g)
{
    __shared__ unsigned s_base[NB1], s_m[NB1];
    const int b = blockIdx.y, tid = threadIdx.x;
    SelHdr *hdr = ws.hdr + b;
    const ValRange r = sel_range(ws.rng[b]);
    const unsigned t1 = hdr->t1;
    if (!r.ok || t1 >= NB1) return;
    for (int j = tid; j < NB1; j += 256) { s_base[j] = ws.cbase[(size_t)b * NB1 + j]; s_m[j] = ws.cm[(size_t)b * NB1 + j]; }
    __syncthreads();
    const T *sc = score + (size_t)b * g.H * g.W;
    unsigned *fcur = ws.fcur + (size_t)b * g.nfmax;
    unsigned long long *ckey = ws.ckey + (size_t)b * g.nfmax * BIN_CAP;
    unsigned *cpos = ws.cpos + (size_t)b * g.nfmax * BIN_CAP;
    bool overflow = false;
    // PL_U row segments per iteration: their loads and their returned atomics are in flight together (beside a
    // bandwidth-bound kernel each dependent round trip costs microseconds)
    for (int y = blockIdx.x; y < g.H; y += gridDim.x)
        for (int x0 = 0; x0 < g.W; x0 += PL_U * 256) {
            double v[PL_U];
#pragma unroll
            for (int u = 0; u < PL_U; ++u) {
                const int x = x0 + u * 256 + tid;
                v[u] = x < g.W ? (double)sc[(size_t)y * g.W + x] : __longlong_as_double(0xfff0000000000000ll);    // -inf: never a candidate
            }
            unsigned long long k[PL_U];
            unsigned f[PL_U], slot[PL_U];
            bool cand[PL_U];
#pragma unroll
            for (int u = 0; u < PL_U; ++u) {
                k[u] = order_key(v[u]);
                cand[u] = k[u] < KEY_POS_INF && k[u] != KEY_NEG_INF;
                double t = 0.0;
                const int j = cand[u] ? coarse_bin(v[u], r, t) : 0;
                cand[u] = cand[u] && (unsigned)j >= t1;
                f[u] = 0;
                if (cand[u]) {
                    const unsigned mj = s_m[j];
                    unsigned sb = (unsigned)((t - (double)j) * (double)mj);      // sub-bin inside the coarse bin, monotone in v
                    sb = sb > mj - 1 ? mj - 1 : sb;
                    f[u] = s_base[j] + (mj - 1 - sb);
                }
            }
            // One returned atomic per candidate -- unless ALL candidates of the wave in this trip go to one bin (a plateau of ties: half a
            // million atomics on one counter otherwise, 18 ms per 16 such images): then one atomic for the wave's count, checked once
            // per trip with scalar bookkeeping only (a per-candidate version of this test cost 10-20 registers and with them the
            // second wave per SIMD beside the feature kernel: 850 -> 1340 us per 16 ordinary images).
            unsigned long long cm[PL_U];
            unsigned f_first = 0u, tot = 0u;
            cm[0] = __ballot(cand[0]);
            bool uni = __builtin_popcountll(cm[0]) >= 32;                       // ordinary maps leave here: one ballot, one count
            if (uni) {
                f_first = (unsigned)__builtin_amdgcn_readlane((int)f[0], (int)__builtin_ctzll(cm[0]));
                tot = (unsigned)__builtin_popcountll(cm[0]);
#pragma unroll
                for (int u = 1; u < PL_U; ++u) { cm[u] = __ballot(cand[u]); tot += (unsigned)__builtin_popcountll(cm[u]); }
#pragma unroll
                for (int u = 0; u < PL_U; ++u) uni = uni && __ballot(cand[u] && f[u] != f_first) == 0ull;
            }
            if (uni) {
                unsigned base = 0u;
                if ((tid & 63) == 0) base = atomicAdd(&fcur[f_first], tot);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                for (int u = 0; u < PL_U; ++u) {
                    slot[u] = base + __builtin_amdgcn_mbcnt_hi((unsigned)(cm[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm[u], 0u));
                    base += (unsigned)__builtin_popcountll(cm[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < PL_U; ++u) slot[u] = cand[u] ? atomicAdd(&fcur[f[u]], 1u) : 0u;      // PL_U returns in flight
            }
#pragma unroll
            for (int u = 0; u < PL_U; ++u)
                if (cand[u]) {
                    if (slot[u] < (unsigned)BIN_CAP) {
                        const size_t sl = (size_t)f[u] * BIN_CAP + slot[u];
                        ckey[sl] = k[u];
                        cpos[sl] = ((unsigned)(x0 + u * 256 + tid) << 16) | (unsigned)y;
                    } else overflow = true;
                }
        }
    if (__any(overflow) && (tid & 63) == 0) atomicOr(&hdr->flags, (unsigned)SEL_F_OVERFLOW);
}

// ------------------------------------------------------------------ key extrema of the bins that ran out of slots
// Only for images in which k_sel_place dropped a candidate (SEL_F_OVERFLOW: a plateau of ties; the launch returns at once otherwise):
// one more pass over the map with the same binning arithmetic, and every candidate whose bin is full adds its key to the bin's
// extrema -- the lanes of a wave that share a bin reduce their keys first and send ONE pair of atomics (a plateau: all 64 lanes, one
// bin).  The sweep walks such a bin through the map itself, key by key, between those extrema.  (A first version kept the extrema in
// k_sel_place itself, with one slot atomic per wave for plateaus: 48 -> 58-70 registers, one wave per SIMD instead of two beside
// the feature kernel, 850 -> 1340 us per 16 ordinary images.)
template <typename T>
__global__ void __launch_bounds__(256) k_sel_full_extrema(const T *__restrict__ score, BinWs ws, BinGeom g)
{
    __shared__ unsigned s_base[NB1], s_m[NB1];
    const int b = blockIdx.y, tid = threadIdx.x;
    const SelHdr *hdr = ws.hdr + b;
    if (!(hdr->flags & SEL_F_OVERFLOW)) return;
    const ValRange r = sel_range(ws.rng[b]);
    const unsigned t1 = hdr->t1;
    if (!r.ok || t1 >= NB1) return;
    for (int j = tid; j < NB1; j += 256) { s_base[j] = ws.cbase[(size_t)b * NB1 + j]; s_m[j] = ws.cm[(size_t)b * NB1 + j]; }
    __syncthreads();
    const T *sc = score + (size_t)b * g.H * g.W;
    const unsigned *fcur = ws.fcur + (size_t)b * g.nfmax;
    unsigned long long *okmin_inv = ws.okmin_inv + (size_t)b * g.nfmax, *okmax = ws.okmax + (size_t)b * g.nfmax;
    for (int y = blockIdx.x; y < g.H; y += gridDim.x)
        for (int x0 = 0; x0 < g.W; x0 += 256) {
            const int x = x0 + tid;
            const double v = x < g.W ? (double)sc[(size_t)y * g.W + x] : __longlong_as_double(0xfff0000000000000ll);
            const unsigned long long k = order_key(v);
            bool cand = k < KEY_POS_INF && k != KEY_NEG_INF;
            double t = 0.0;
            const int j = cand ? coarse_bin(v, r, t) : 0;
            cand = cand && (unsigned)j >= t1;
            unsigned f = 0;
            if (cand) {
                const unsigned mj = s_m[j];
                unsigned sb = (unsigned)((t - (double)j) * (double)mj);
                sb = sb > mj - 1 ? mj - 1 : sb;
                f = s_base[j] + (mj - 1 - sb);
            }
            const bool full = cand && fcur[f] > (unsigned)BIN_CAP;
            unsigned long long todo = __ballot(full);
            while (todo) {
                const int l0 = (int)__builtin_ctzll(todo);
                const unsigned f0 = (unsigned)__builtin_amdgcn_readlane((int)f, l0);
                const bool mine = full && f == f0;
                const unsigned long long kx = wave_max_u64(mine ? k : 0ull), kn = wave_max_u64(mine ? ~k : 0ull);
                if ((tid & 63) == l0) { atomicMax(&okmax[f0], kx); atomicMax(&okmin_inv[f0], kn); }
                todo &= ~__ballot(mine);
            }
        }
}

// ------------------------------------------------------------------ resolve one bin (one wave)
// The survivors of a bin -- candidates no earlier pick suppresses -- are taken in exact (value desc, w asc, h asc)
// order: arg-max over the wave, commit (pick list, pick grid), kill the survivors inside the new window, repeat.
// Returns 0 (bin done) or 1 (n_regions reached).  With a single survivor alive the arg-max is a lane read.
template <int R>
__device__ __forceinline__ int resolve_bin(const unsigned long long *skey, const unsigned *spos, unsigned sc, int lane, int r, int cs,
                                           const BinGeom &g, unsigned char *grid, unsigned *plist, int &np)
{
    Cand e[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const unsigned idx = lane + 64 * i;
        const bool ok = idx < sc;
        const unsigned ic = ok ? idx : 0u;
        const unsigned long long k = skey[ic];
        const unsigned p = spos[ic];
        e[i].key = ok ? k : 0ull;                                // 0 = dead (below every real key)
        e[i].pos = ok ? p : 0xffffffffu;
    }
    while (true) {
        Cand best = e[0];
#pragma unroll
        for (int i = 1; i < R; ++i) {
            const bool take = better(e[i], best);
            best.key = take ? e[i].key : best.key;
            best.pos = take ? e[i].pos : best.pos;
        }
        const unsigned long long am = __ballot(best.key != 0ull);
        if (am == 0ull) return 0;
        // arg-max under (key desc, pos asc).  Survivors of one bin lie in one narrow value slice: their keys usually
        // share the high word and are distinct, so the common case is ONE 32-bit DPP reduction (the general form is three)
        const unsigned hi = (unsigned)(best.key >> 32), lo = (unsigned)best.key;
        const bool alive = best.key != 0ull;
        Cand top;
        const int l0 = (int)__builtin_ctzll(am);
        if ((am & (am - 1ull)) == 0ull) {                        // exactly one lane holds a live survivor
            top.key = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, l0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, l0);
            top.pos = (unsigned)__builtin_amdgcn_readlane((int)best.pos, l0);
        } else {
            unsigned mh = (unsigned)__builtin_amdgcn_readlane((int)hi, l0);
            if (__ballot(alive && hi != mh) != 0ull) mh = wave_umax(alive ? hi : 0u);
            const unsigned ml = wave_umax((alive && hi == mh) ? lo : 0u);
            const bool tie = alive && hi == mh && lo == ml;
            const unsigned long long tm = __ballot(tie);
            top.key = ((unsigned long long)mh << 32) | ml;
            if ((tm & (tm - 1ull)) == 0ull) top.pos = (unsigned)__builtin_amdgcn_readlane((int)best.pos, (int)__builtin_ctzll(tm));
            else top.pos = ~wave_umax(tie ? ~best.pos : 0u);
        }
        const int px = (int)(top.pos >> 16), py = (int)(top.pos & 0xffffu);
        if (lane == 0) {                                         // the pick table's (h, w, value) rows are written by k_sel_apply
            plist[np] = top.pos;
            const int pcx = (int)__umulhi((unsigned)px, g.cmul), pcy = (int)__umulhi((unsigned)py, g.cmul);
            grid[(pcy + 1) * g.gstride + pcx + 1] = (unsigned char)(1 + (((py - pcy * cs) << 4) | (px - pcx * cs)));
        }
        ++np;
        if (np >= g.n_regions) return 1;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ex = (int)(e[i].pos >> 16), ey = (int)(e[i].pos & 0xffffu);
            const bool hit = (unsigned)(ex - px + r) <= (unsigned)(2 * r) && (unsigned)(ey - py + r) <= (unsigned)(2 * r);
            e[i].key = hit ? 0ull : e[i].key;
        }
    }
}

// ------------------------------------------------------------------ resolve one bin, all survivors at once (<= 64)
// In the maps this path sees, a bin leaves a handful of survivors (6.7 on average at 1024 x 2048) and almost every one
// of them becomes a pick: they passed the filter against all earlier picks and rarely lie within a window of each
// other.  So instead of one arg-max round per pick (about 740 cycles each for a lone wave: DPP reductions, lane reads,
// exec-mask branches), every lane learns in ONE pass over the survivors which of them precede it (-> its rank) and which
// of those lie within its window (-> who can suppress it), and the greedy rule is then iterated on wave-uniform masks:
//     dead  : a preceding neighbour is a pick            pick : every preceding neighbour is dead
// The highest-priority undecided survivor is decided in every round, so the loop ends (one round when nothing
// conflicts).  The result is the sequential rule's, pick for pick and in the same order: pick index = np + rank among picks.
// Returns 0 (bin done) or 1 (n_regions reached).
__device__ __forceinline__ int resolve_bin_parallel(const unsigned long long *skey, const unsigned *spos, unsigned sc, int lane, int r,
                                                    int cs, const BinGeom &g, unsigned char *grid, unsigned *plist, int &np)
{
    const bool alive = (unsigned)lane < sc;
    const unsigned ic = alive ? (unsigned)lane : 0u;
    const unsigned long long key = skey[ic];
    const unsigned pos = spos[ic];
    const unsigned khi = (unsigned)(key >> 32), klo = (unsigned)key;
    const int x = (int)(pos >> 16), y = (int)(pos & 0xffffu);
    unsigned long long before = 0ull, nb = 0ull;                 // survivors that precede this one / ... and lie within its window
    for (unsigned j = 0; j < sc; ++j) {
        const unsigned jh = (unsigned)__builtin_amdgcn_readlane((int)khi, (int)j), jl = (unsigned)__builtin_amdgcn_readlane((int)klo, (int)j);
        const unsigned jp = (unsigned)__builtin_amdgcn_readlane((int)pos, (int)j);
        const unsigned long long jk = ((unsigned long long)jh << 32) | jl;
        const bool prec = jk > key || (jk == key && jp < pos);
        const int jx = (int)(jp >> 16), jy = (int)(jp & 0xffffu);
        const bool near = (unsigned)(jx - x + r) <= (unsigned)(2 * r) && (unsigned)(jy - y + r) <= (unsigned)(2 * r);
        const unsigned long long bit = 1ull << j;
        before |= prec ? bit : 0ull;
        nb |= (prec && near) ? bit : 0ull;
    }
    unsigned long long pickm = 0ull, deadm = 0ull;
    bool undecided = alive;
    while (true) {
        const bool dies = undecided && (nb & pickm) != 0ull;
        const bool wins = undecided && !dies && (nb & ~deadm) == 0ull;
        const unsigned long long np_m = __ballot(wins), nd_m = __ballot(dies);
        pickm |= np_m;
        deadm |= nd_m;
        undecided = undecided && !wins && !dies;
        if (__ballot(undecided) == 0ull) break;
    }
    const bool is_pick = alive && ((pickm >> lane) & 1ull) != 0ull;
    const int idx = np + (int)__builtin_popcountll(before & pickm);
    if (is_pick && idx < g.n_regions) {
        plist[idx] = pos;
        const int pcx = (int)__umulhi((unsigned)x, g.cmul), pcy = (int)__umulhi((unsigned)y, g.cmul);
        grid[(pcy + 1) * g.gstride + pcx + 1] = (unsigned char)(1 + (((y - pcy * cs) << 4) | (x - pcx * cs)));
    }
    np += (int)__builtin_popcountll(pickm);
    if (np >= g.n_regions) { np = g.n_regions; return 1; }
    return 0;
}

// ------------------------------------------------------------------ the sweep
// (at most 96 registers: k_feat_reduce's cap leaves 512 - 4 x 104 = 96 per SIMD free, and a sweep workgroup that needs more waits
// for a CU to drain -- with plateau_scan inlined the compiler took 97, and 16 selections beside the streaming kernel went from 2.5
// to 3.0 ms)
__global__ void __launch_bounds__(SW_TPB, 5) k_sel_sweep(BinWs ws, BinGeom g, int *__restrict__ n_picked, const void *__restrict__ score_maps,
                                                      int score_f64)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *grid = smem;                                               // pick grid, one byte per cell (+ border)
    unsigned *gridw = reinterpret_cast<unsigned *>(smem);
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem + g.grid_bytes);
    unsigned *spos = reinterpret_cast<unsigned *>(skey + SW_SURV);
    unsigned *fwin = spos + SW_SURV;                                          // FWIN + 1 offsets
    unsigned *ctl = fwin + FWIN + 1;                                          // [0] survivors, [1] state after a resolve

    __builtin_amdgcn_s_setprio(3);     // a short serial chain beside bandwidth-bound kernels: issue ahead of them
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SelHdr *hdr = ws.hdr + b;
    const unsigned nf = hdr->nf;
    const bool truncated = hdr->truncated != 0;
    const unsigned *fcnt = ws.fcur + (size_t)b * g.nfmax;
    const unsigned long long *ckey = ws.ckey + (size_t)b * g.nfmax * BIN_CAP;
    const unsigned *cpos = ws.cpos + (size_t)b * g.nfmax * BIN_CAP;
    unsigned *plist = ws.plist + (size_t)b * g.n_regions;
    const unsigned long long *okmin_inv = ws.okmin_inv + (size_t)b * g.nfmax, *okmax = ws.okmax + (size_t)b * g.nfmax;
    const double *map64 = score_f64 ? (const double *)score_maps + (size_t)b * g.H * g.W : nullptr;
    const float *map32 = score_f64 ? nullptr : (const float *)score_maps + (size_t)b * g.H * g.W;

    for (unsigned i = tid * 16; i < g.grid_bytes; i += SW_TPB * 16) *reinterpret_cast<uint4 *>(smem + i) = make_uint4(0, 0, 0, 0);
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    __syncthreads();

    int np = 0;                                  // maintained by wave 0
    int fin = 0;                                 // 0 running, 1 done, 2 bail
    int why = HALO_SWEEP_DONE;                   // reason of a hand-over (reported through ws.handover)
    const int r = g.mrad, cs = g.cs;

    // ---- the candidate stream.  Bins are cut into chunks of <= 256 candidates (one per thread); an iterator over the
    // bin offsets (staged FWIN at a time in LDS) runs THREE chunks ahead of the one being filtered, and each chunk's
    // (key, pos) loads are issued as soon as it is known -- beside a bandwidth-bound kernel a memory round trip takes
    // microseconds, a chunk only ~1 us.  The three register sets rotate by unrolling the loop body three times, so the
    // compiler's counted s_waitcnt vmcnt waits for the oldest set only.
    // A chunk = up to SW_MB consecutive non-empty bins holding <= SW_TPB candidates together (a bin has at most BIN_CAP <=
    // SW_TPB / 2 of them, so it is never split).  Nothing depends on the order inside a chunk -- the resolve step takes its
    // survivors in exact priority order -- and chunks are still monotone in the value, so merging bins changes no pick; it
    // halves the number of filter / resolve rounds (two barriers and a serial resolve each) for the same candidates.
    struct Chunk { unsigned start[SW_MB], cnt[SW_MB], total; bool last, valid, plateau; unsigned pbin; };
    unsigned fw0 = 0, fwn = 0;                   // fwin holds the candidate counts of bins fw0 .. fw0 + fwn - 1
    unsigned it_f = 0;                           // next bin to look at
    // A bin that ran out of slots (more than BIN_CAP candidates in one sub-slice of the value range: a plateau of ties) becomes a
    // chunk of its own kind, in its place in the stream: when the walk REACHES it -- everything above it has been swept -- the
    // plateau is walked through the map itself if all its candidates carry one key (plateau_scan below), and handed to the serial
    // kernel from there otherwise (round 4 handed such an image over untouched, wherever the full bin lay).
    auto next_chunk = [&]() {
        Chunk c;
#pragma unroll
        for (int i = 0; i < SW_MB; ++i) { c.start[i] = 0; c.cnt[i] = 0; }
        c.total = 0; c.last = true; c.valid = false; c.plateau = false; c.pbin = 0;
        int nb = 0;
        while (it_f < nf) {
            if (fwn == 0 || it_f >= fw0 + fwn) {             // stage the next window of bin counts (uniform: every thread gets here together)
                lds_barrier();
                fw0 = it_f;
                fwn = nf - it_f < (unsigned)FWIN ? nf - it_f : (unsigned)FWIN;
                for (unsigned i = tid; i < fwn; i += SW_TPB) fwin[i] = fcnt[fw0 + i];
                __syncthreads();
            }
            const unsigned n = fwin[it_f - fw0];
            if (n == 0) { ++it_f; continue; }
            if (n > (unsigned)BIN_CAP) {
                if (nb == 0) { c.plateau = true; c.pbin = it_f; ++it_f; nb = 1; }      // (behind an open chunk: it opens the next one)
                break;
            }
            if (nb == SW_MB || c.total + n > (unsigned)SW_TPB) break;        // this bin opens the next chunk
#pragma unroll
            for (int i = 0; i < SW_MB; ++i)
                if (i == nb) { c.start[i] = it_f * (unsigned)BIN_CAP; c.cnt[i] = n; }
            c.total += n;
            ++nb;
            ++it_f;
        }
        c.valid = nb > 0;
        return c;
    };
    struct Regs { unsigned long long key; unsigned pos; };
    auto issue = [&](const Chunk &c, Regs &d) {      // unconditional loads (clamped index): no exec-mask branch, so the waits stay counted
        unsigned t = c.valid ? ((unsigned)tid < c.total ? (unsigned)tid : c.total - 1u) : 0u, idx = 0u;
        bool found = !c.valid;
#pragma unroll
        for (int i = 0; i < SW_MB; ++i) {
            const bool here = !found && t < c.cnt[i];
            idx = here ? c.start[i] + t : idx;
            found = found || here;
            t -= found ? 0u : c.cnt[i];
        }
        d.key = ckey[idx];
        d.pos = cpos[idx];
    };

#ifdef HALO_SWEEP_STAMPS
    unsigned long long t_f = 0, t_a = 0, t_r = 0, t_b = 0, t0 = __builtin_amdgcn_s_memtime(), nb = 0, t_surv = 0, t_smax = 0;
#define STAMP(acc) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define STAMP(acc)
#endif
    // is the pixel (w << 16 | h) outside the window of every pick made so far?  (the 3 x 3 cells around it, three 8-byte LDS reads)
    auto grid_alive = [&](unsigned pos) {
        const int x = (int)(pos >> 16), y = (int)(pos & 0xffffu);
        const int cx = (int)__umulhi((unsigned)x, g.cmul), cy = (int)__umulhi((unsigned)y, g.cmul);
        const int lx = x - cx * cs, ly = y - cy * cs;
        bool alive = true;
        const int cell0 = cy * g.gstride + cx;                               // padded address of cell (cy-1, cx-1)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int addr = cell0 + a * g.gstride;
            const unsigned w0 = gridw[addr >> 2], w1 = gridw[(addr >> 2) + 1];
            const unsigned win = __builtin_amdgcn_alignbit(w1, w0, (unsigned)(addr & 3) * 8u);
            const int oy = (a - 1) * cs - ly + r;                            // ddy + r = oy + dy
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                const unsigned cc = (win >> (8 * bb)) & 0xffu;
                const unsigned q = cc - 1u;
                const int ddy = oy + (int)(q >> 4), ddx = (bb - 1) * cs - lx + r + (int)(q & 15u);
                const bool hit = cc != 0u && (unsigned)ddy <= (unsigned)(2 * r) && (unsigned)ddx <= (unsigned)(2 * r);
                alive = alive && !hit;
            }
        }
        return alive;
    };
    // survivors -> the LDS list (one wave-level atomic per wave; entries past the list's capacity are counted, not stored)
    auto push = [&](bool alive, unsigned long long key, unsigned pos) {
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&ctl[0], (unsigned)__builtin_popcountll(mask));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (alive && slot < (unsigned)SW_SURV) { skey[slot] = key; spos[slot] = pos; }
        }
    };
    // wave 0 takes the listed survivors in exact priority order (every thread calls it: two barriers); fin follows
    auto resolve_listed = [&]() {
        STAMP(t_f)
        lds_barrier();
        STAMP(t_a)
        if (wave == 0) {
            const unsigned sc = ctl[0];
#ifdef HALO_SWEEP_STAMPS
            t_surv += sc; if (sc > t_smax) t_smax = sc;
#endif
            int state = 0;
            if (sc > (unsigned)SW_SURV) state = 3;                           // more unsuppressed ties than fit: hand over
            else if (sc) {
                // at most 64 survivors (the common case): all at once; more: the sequential arg-max loop, four per lane
                if (sc <= 64u) state = resolve_bin_parallel(skey, spos, sc, lane, r, cs, g, grid, plist, np);
                else state = resolve_bin<4>(skey, spos, sc, lane, r, cs, g, grid, plist, np);
            }
            if (lane == 0) { ctl[0] = 0; ctl[1] = (unsigned)state; }
        }
        STAMP(t_r)
        lds_barrier();
        fin = (int)ctl[1];
        if (fin == 3) { fin = 2; why = HALO_SWEEP_SURVIVORS; }
        STAMP(t_b)
#ifdef HALO_SWEEP_STAMPS
        ++nb;
#endif
    };
    // ---- a plateau of exact ties, walked through the map itself.  Bin `fb` received more candidates than it has slots.  Candidates
    // that carry ONE key are ordered by their positions -- smallest w, then smallest h: the reference's tie-break (build.py:38-43)
    // -- so in priority order they are simply the pixels of the map with that key in column-major order, and the map says which
    // those are: no list of them is needed.  The bin's keys lie between the extrema of its stored and its dropped candidates, and
    // every pixel whose key lies in that range belongs to the bin (binning is monotone in the value).  So: starting with the bin's
    // largest key K, the workgroup walks the map's columns -- PL_PP positions per thread and trip (half a column of 1024 rows per trip at
    // 1024 x 2048), the next trip's loads in flight --, tests the pixels with key K against the pick grid and lists the survivors
    // (resolved when the list holds 32, or when a trip's survivors do not fit it: the trip is then redone piece by piece behind a
    // resolve, a piece being at most SW_TPB <= SW_SURV positions), and notes the largest key below K it meets in the bin's range:
    // the next K.  Listing survivors of several trips before resolving them changes no pick, exactly as merging bins does not.
    // A plateau and a few stragglers in its sub-slice of the value range take PL_KEYS passes at most; a bin with more distinct keys
    // than that (dense near-ties) hands the image over from where the walk stands.
    constexpr int PL_PP = 2, PL_KEYS = 4;      // (PL_PP = 4 pushes the kernel past its 96 registers: 76 bytes of scratch in the main path)
    auto plateau_scan = [&](unsigned fb) {
        // key range of the bin (k_sel_full_extrema: over ALL its candidates, stored or dropped)
        const unsigned long long kmax_b = okmax[fb], kmin_b = ~okmin_inv[fb];
        if (kmax_b == 0ull) { fin = 2; why = HALO_SWEEP_BIN_OVERFLOW; return; }      // (no extrema recorded: cannot happen for a full bin)
        const unsigned P = (unsigned)g.H * (unsigned)g.W, GP = SW_TPB * PL_PP;
        auto fetch = [&](unsigned g0, unsigned long long (&kk)[PL_PP], unsigned (&pp)[PL_PP]) {
#pragma unroll
            for (int i = 0; i < PL_PP; ++i) {
                const unsigned pidx = g0 + (unsigned)i * SW_TPB + (unsigned)tid;
                const unsigned pc = pidx < P ? pidx : P - 1u;
                const unsigned w_ = pc / (unsigned)g.H, h_ = pc - w_ * (unsigned)g.H;
                const size_t o = (size_t)h_ * g.W + w_;
                const double v = map64 ? map64[o] : (double)map32[o];
                kk[i] = pidx < P ? order_key(v) : 0ull;                            // 0: below every real key
                pp[i] = (w_ << 16) | h_;
            }
        };
        unsigned long long K = kmax_b;
#pragma unroll 1
        for (int pass = 0; !fin; ++pass) {
            if (pass == PL_KEYS) { fin = 2; why = HALO_SWEEP_BIN_OVERFLOW; break; }       // more distinct keys than passes: hand over
            unsigned listed = 0;                                                       // survivors in the list (uniform)
            unsigned long long knext = 0ull;                                           // largest key of the bin below K seen by this thread
            unsigned long long kc[PL_PP], kn[PL_PP];
            unsigned pc_[PL_PP], pn[PL_PP];
            fetch(0u, kc, pc_);
            for (unsigned g0 = 0; g0 < P && !fin; g0 += GP) {
                fetch(g0 + GP < P ? g0 + GP : g0, kn, pn);                             // the next trip's keys are in flight during this one
#pragma unroll
                for (int i = 0; i < PL_PP; ++i) {
                    push(kc[i] == K && grid_alive(pc_[i]), kc[i], pc_[i]);
                    knext = (kc[i] < K && kc[i] >= kmin_b && kc[i] > knext) ? kc[i] : knext;
                }
                lds_barrier();
                const unsigned total = ctl[0];
                if (total > (unsigned)SW_SURV) {
                    lds_barrier();                                                     // everybody has read the count
                    if (tid == 0) ctl[0] = listed;                                     // forget this trip's survivors ...
                    resolve_listed();                                                  // ... take the ones listed before it ...
#pragma unroll 1
                    for (int i = 0; i < PL_PP && !fin; ++i) {                          // ... and redo the trip one piece at a time
                        push(kc[i] == K && grid_alive(pc_[i]), kc[i], pc_[i]);
                        resolve_listed();
                    }
                    listed = 0;
                } else {
                    listed = total;
                    if (listed >= 32u) { resolve_listed(); listed = 0; }
                }
#pragma unroll
                for (int i = 0; i < PL_PP; ++i) { kc[i] = kn[i]; pc_[i] = pn[i]; }
            }
            if (fin) break;
            if (listed) resolve_listed();
            if (fin) break;
            // the next key of the bin: the block's maximum of what its threads met (the survivor list is empty: its first words carry it)
            knext = wave_max_u64(knext);
            lds_barrier();
            if (lane == 0) skey[wave] = knext;
            lds_barrier();
            knext = skey[0];
#pragma unroll
            for (int i = 1; i < SW_TPB / 64; ++i) knext = skey[i] > knext ? skey[i] : knext;
            lds_barrier();                                                             // read before the next pass lists survivors there
            if (knext == 0ull) break;                                                  // the bin is exhausted: on with the bins below it
            K = knext;
        }
    };
    // one chunk: filter against the pick grid, survivors -> LDS list; at the end of a bin wave 0 resolves the survivors
    auto step = [&](const Chunk &c, const Regs &d) {
        if (c.plateau) { plateau_scan(c.pbin); return; }
        push((unsigned)tid < c.total && grid_alive(d.pos), d.key, d.pos);
        if (!c.last) return;
        resolve_listed();
    };
    Chunk c0 = next_chunk(), c1, c2;
    Regs d0, d1, d2;
    issue(c0, d0);
    c1 = next_chunk(); issue(c1, d1);
    c2 = next_chunk(); issue(c2, d2);
    while (true) {
        if (!c0.valid || fin) break;
        step(c0, d0);
        c0 = next_chunk(); issue(c0, d0);
        if (!c1.valid || fin) break;
        step(c1, d1);
        c1 = next_chunk(); issue(c1, d1);
        if (!c2.valid || fin) break;
        step(c2, d2);
        c2 = next_chunk(); issue(c2, d2);
    }
#ifdef HALO_SWEEP_STAMPS
    if (tid == 0) {      // diagnostic build only: phase cycle sums of wave 0 into the header's padding words
        hdr->pad[0] = (unsigned)(t_f >> 4); hdr->pad[1] = (unsigned)(t_a >> 4); hdr->pad[2] = (unsigned)(t_r >> 4); hdr->pad[3] = (unsigned)(t_b >> 4);
        hdr->nvalid = (unsigned)nb; hdr->ncand = (unsigned)t_surv; hdr->flags = (unsigned)t_smax;
    }
#endif
    if (tid == 0) {
        if (fin == 0) {                          // candidates exhausted: final unless the threshold bin was dropped
            fin = truncated ? 2 : 1;
            if (truncated) why = hdr->t1 >= (unsigned)NB1 ? HALO_SWEEP_BAD_VALUES : HALO_SWEEP_EXHAUSTED;
        }
        if (ws.handover) { ws.handover[2 * b] = fin == 1 ? HALO_SWEEP_DONE : why; ws.handover[2 * b + 1] = np; }
        hdr->np = np;
        hdr->status = fin == 1 ? SEL_DONE : SEL_BAIL;
        if (fin == 1 && n_picked) n_picked[b] = np;
    }
}

// ------------------------------------------------------------------ windows of the picks
// build.py:45-62 for every pick at once: all four writes store constants (or ground_truth of the
// same pixel), so the order between picks does not matter.
// A few fat workgroups per image whose waves stride over the picks, not one small workgroup per four picks: beside the
// streaming feature kernel every workgroup waits to be PLACED (9 328 of them took 0.70 ms for 6 us of work in the
// round-3 bench trace); APPLY_WGS x 4 waves per image keep the same stores in flight with 18 picks per wave.
constexpr int APPLY_WGS = 32;

template <typename T>
__global__ void __launch_bounds__(256) k_sel_apply(T *__restrict__ score, unsigned char *__restrict__ active,
                                                   unsigned char *__restrict__ selected, long long *__restrict__ active_mask,
                                                   const long long *__restrict__ gt, double *__restrict__ picks, BinWs ws, BinGeom g)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int np = ws.hdr[b].np;
    const size_t hw = (size_t)g.H * g.W;
    T *sc = score + (size_t)b * hw;
    unsigned char *act = active + (size_t)b * hw, *sel = selected + (size_t)b * hw;
    long long *am = active_mask + (size_t)b * hw;
    const long long *gtb = gt + (size_t)b * hw;
    const T neg_inf = sizeof(T) == 8 ? (T)__longlong_as_double(0xfff0000000000000ll) : (T)__uint_as_float(0xff800000u);
    const unsigned *plist = ws.plist + (size_t)b * g.n_regions;
    const int nwaves = gridDim.x * 4;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < np; p += nwaves) {
        const unsigned pos = plist[p];
        const int w = (int)(pos >> 16), h = (int)(pos & 0xffffu);
        // (h, w, value) row of the pick table.  The value is read BEFORE this wave writes its window: no other pick's window
        // covers this pixel (picks are more than mask_radius apart), so it still holds the original score
        if (picks) {
            const double v = key_value(order_key((double)sc[(size_t)h * g.W + w]));      // -0 -> +0, as the serial kernel reports it
            if (lane == 0) {
                double *pk = picks + ((size_t)b * g.n_regions + p) * 3;
                pk[0] = (double)h;
                pk[1] = (double)w;
                pk[2] = v;
            }
        }
        const int my0 = h - g.mrad < 0 ? 0 : h - g.mrad, my1 = h + g.mrad >= g.H ? g.H - 1 : h + g.mrad;
        const int mx0 = w - g.mrad < 0 ? 0 : w - g.mrad, mx1 = w + g.mrad >= g.W ? g.W - 1 : w + g.mrad;
        const int ay0 = h - g.arad < 0 ? 0 : h - g.arad, ay1 = h + g.arad >= g.H ? g.H - 1 : h + g.arad;
        const int ax0 = w - g.arad < 0 ? 0 : w - g.arad, ax1 = w + g.arad >= g.W ? g.W - 1 : w + g.arad;
        const int aw = ax1 - ax0 + 1, an = aw * (ay1 - ay0 + 1);
        const int mw = mx1 - mx0 + 1, mn = mw * (my1 - my0 + 1);
        for (int e = lane; e < an; e += 64) {                       // selected[...] = True; active_mask[...] = ground_truth[...]
            const size_t o = (size_t)(ay0 + e / aw) * g.W + (ax0 + e % aw);
            sel[o] = 1;
            am[o] = gtb[o];
        }
        for (int e = lane; e < mn; e += 64) {                       // score[...] = -inf; active[...] = True
            const size_t o = (size_t)(my0 + e / mw) * g.W + (mx0 + e % mw);
            sc[o] = neg_inf;
            act[o] = 1;
        }
    }
}

}  // namespace halo

using namespace halo;

namespace halo {

static unsigned sel_target()
{
    const char *e = getenv("HALO_SEL_TARGET");       // tuning aid: expected candidates per fine bin
    const int v = e ? atoi(e) : 0;
    return v >= 8 && v <= BIN_CAP / 2 ? (unsigned)v : (unsigned)(BIN_CAP / 2);      // bins are Poisson-filled: keep 2x headroom
}

BinPlan binned_plan(int64_t B, int64_t H, int64_t W, int64_t n_regions, int64_t arad, int64_t mrad)
{
    BinPlan p;
    memset(&p, 0, sizeof(p));
    // mask radius 0 (a pick suppresses only itself) has 1-pixel cells, whose reciprocal multiplier 2^32 does not fit: serial kernel
    if (B <= 0 || H <= 0 || W <= 0 || n_regions <= 0 || H > 65535 || W > 65535 || mrad < 1 || mrad > 14) return p;
    BinGeom &g = p.g;
    g.H = (int)H; g.W = (int)W; g.n_regions = (int)n_regions; g.arad = (int)arad; g.mrad = (int)mrad;
    g.cs = (int)mrad + 1;
    g.gcy = (int)cdiv(H, g.cs);
    g.gcx = (int)cdiv(W, g.cs);
    g.gstride = (int)align_up((size_t)g.gcx + 2, 4);
    g.cmul = (unsigned)((0x100000000ull + (unsigned)g.cs - 1) / (unsigned)g.cs);
    g.grid_bytes = (unsigned)align_up((size_t)(g.gcy + 2) * g.gstride + 8, 16);
    p.lds_bytes = g.grid_bytes + (size_t)SW_SURV * 12 + (FWIN + 1) * 4 + 16;
    if (p.lds_bytes > 156 * 1024) return p;                 // pick grid does not fit the CU's LDS: serial kernel
    const unsigned long long hw = (unsigned long long)H * W, win = (unsigned long long)(2 * mrad + 1) * (2 * mrad + 1);
    const unsigned long long kneed = win * (unsigned long long)n_regions < hw ? win * (unsigned long long)n_regions : hw;
    unsigned long long cap = 2 * kneed < 65536 ? 65536 : 2 * kneed;
    if (cap > hw) cap = hw;
    g.kneed = (unsigned)kneed;
    g.captot = (unsigned)cap;
    g.target = sel_target();
    g.nfmax = (unsigned)(cap / g.target + NB1 + 1);
    size_t o = 0;
    auto take = [&](size_t per_image) { const size_t at = o; o += align_up(per_image * (size_t)B, 256); return at; };
    p.off_hdr = take(sizeof(SelHdr));
    p.off_hist1 = take((size_t)NB1 * 4);
    p.off_fcur = take((size_t)g.nfmax * 4);
    p.off_okmin = take((size_t)g.nfmax * 8);
    p.off_okmax = take((size_t)g.nfmax * 8);
    p.zero_bytes = o;                                       // everything above is cleared at the start of a call
    p.off_cbase = take((size_t)NB1 * 4);
    p.off_cm = take((size_t)NB1 * 4);
    p.off_ckey = take((size_t)g.nfmax * BIN_CAP * 8);
    p.off_cpos = take((size_t)g.nfmax * BIN_CAP * 4);
    p.off_plist = take((size_t)g.n_regions * 4);
    p.total_bytes = o + 256;
    p.ok = true;
    return p;
}

// The exact value range of score maps as the record halo_greedy_select_ranged accepts (also the scorer's fallback when it
// cannot bound the range for free): zero the records, then the same reduction the selector runs on its own.
int score_range_exact(const void *score, int dtype, int64_t B, int64_t hw, void *range_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_sel_zero, dim3(1u), dim3(256), 0, st, (uint4 *)range_out, (size_t)B * sizeof(SelHdr) / 16);
    BinWs ws;
    memset(&ws, 0, sizeof(ws));
    ws.hdr = (SelHdr *)range_out;
    ws.rng = ws.hdr;
    const unsigned gx = (unsigned)(cdiv(hw, 256) < 128 ? cdiv(hw, 256) : 128);
    if (dtype == HALO_F64) hipLaunchKernelGGL(k_sel_range<double>, dim3(gx, (unsigned)B), dim3(256), 0, st, (const double *)score, (long long)hw, ws);
    else hipLaunchKernelGGL(k_sel_range<float>, dim3(gx, (unsigned)B), dim3(256), 0, st, (const float *)score, (long long)hw, ws);
    return HALO_OK;
}

int binned_select(void *score, int dtype, int64_t B, const BinPlan &p, uint8_t *active, uint8_t *selected, int64_t *active_mask,
                  const int64_t *gt, double *picks, int32_t *n_picked, void *workspace, size_t workspace_bytes, hipStream_t st,
                  SelHdr **hdr_out, const void *score_range, int32_t *handover)
{
    if (!workspace || workspace_bytes < p.total_bytes) return fail(HALO_E_WORKSPACE, "halo_greedy_select: workspace too small");
    char *base = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if ((size_t)(base - (char *)workspace) + p.total_bytes - 256 > workspace_bytes) return fail(HALO_E_WORKSPACE, "halo_greedy_select: workspace too small");
    BinWs ws;
    ws.hdr = (SelHdr *)(base + p.off_hdr);
    ws.rng = score_range ? (SelHdr *)score_range : ws.hdr;            // the scorer already knows the range: no pass over the map
    ws.rng_hist = score_range ? (const unsigned *)((const char *)score_range + range_hist_offset(B)) : nullptr;
    ws.hist1 = (unsigned *)(base + p.off_hist1);
    ws.fcur = (unsigned *)(base + p.off_fcur);
    ws.okmin_inv = (unsigned long long *)(base + p.off_okmin);
    ws.okmax = (unsigned long long *)(base + p.off_okmax);
    ws.cbase = (unsigned *)(base + p.off_cbase);
    ws.cm = (unsigned *)(base + p.off_cm);
    ws.ckey = (unsigned long long *)(base + p.off_ckey);
    ws.cpos = (unsigned *)(base + p.off_cpos);
    ws.plist = (unsigned *)(base + p.off_plist);
    ws.handover = handover;
    const BinGeom &g = p.g;
    const long long hw = (long long)g.H * g.W;
    {   // zero_bytes is a multiple of 256 (take() aligns every array), base is 256-byte aligned
        const size_t n16 = p.zero_bytes / 16;
        const unsigned gz = (unsigned)(cdiv((int64_t)n16, 256) < 512 ? cdiv((int64_t)n16, 256) : 512);
        hipLaunchKernelGGL(k_sel_zero, dim3(gz ? gz : 1u), dim3(256), 0, st, (uint4 *)base, n16);
    }
    // 128 blocks per image for the passes with a per-block fixed cost (LDS histogram flush: 2048 atomics; 16 KiB table
    // staging): 1024 blocks measured 2x SLOWER end to end, and a wider scatter grid no faster beside the feature stream.
    // (round 3, two hardware queues, 5 s between runs: 128 blocks per image 1370-1375 images/s in the bench, 256 the same, 48 0.4 %
    // and 16 1.3 % slower -- longer-running passes cost the feature kernel beside them more)
    const unsigned gx = (unsigned)(cdiv(hw, 256) < 128 ? cdiv(hw, 256) : 128);
    const unsigned gy = (unsigned)(g.H < 128 ? g.H : 128);
    dim3 blk(256);
    if (dtype == HALO_F64) {
        if (!score_range) hipLaunchKernelGGL(k_sel_range<double>, dim3(gx, (unsigned)B), blk, 0, st, (const double *)score, hw, ws);
        hipLaunchKernelGGL(k_sel_hist1<double>, dim3(gx, (unsigned)B), blk, 0, st, (const double *)score, hw, ws);
    } else {
        if (!score_range) hipLaunchKernelGGL(k_sel_range<float>, dim3(gx, (unsigned)B), blk, 0, st, (const float *)score, hw, ws);
        hipLaunchKernelGGL(k_sel_hist1<float>, dim3(gx, (unsigned)B), blk, 0, st, (const float *)score, hw, ws);
    }
    hipLaunchKernelGGL(k_sel_scan1, dim3((unsigned)B), blk, 0, st, ws, g);
    if (dtype == HALO_F64) hipLaunchKernelGGL(k_sel_place<double>, dim3(gy, (unsigned)B), blk, 0, st, (const double *)score, ws, g);
    else hipLaunchKernelGGL(k_sel_place<float>, dim3(gy, (unsigned)B), blk, 0, st, (const float *)score, ws, g);
    {   // rare: the extrema of bins that ran out of slots (returns at once unless k_sel_place flagged the image)
        const dim3 ge((unsigned)(g.H < 64 ? g.H : 64), (unsigned)B);
        if (dtype == HALO_F64) hipLaunchKernelGGL(k_sel_full_extrema<double>, ge, blk, 0, st, (const double *)score, ws, g);
        else hipLaunchKernelGGL(k_sel_full_extrema<float>, ge, blk, 0, st, (const float *)score, ws, g);
    }
    static LdsLimitSeen seen;      // the pick grid may need more than the default 64 KiB of dynamic LDS
    if (!raise_lds_limit(seen, (const void *)k_sel_sweep, 160 * 1024))
        return fail(HALO_E_LAUNCH, "halo_greedy_select: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(k_sel_sweep, dim3((unsigned)B), dim3(SW_TPB), p.lds_bytes, st, ws, g, n_picked, (const void *)score, dtype == HALO_F64 ? 1 : 0);
    const dim3 ga((unsigned)(cdiv(g.n_regions, 4) < APPLY_WGS ? cdiv(g.n_regions, 4) : APPLY_WGS), (unsigned)B);
    if (dtype == HALO_F64)
        hipLaunchKernelGGL(k_sel_apply<double>, ga, blk, 0, st, (double *)score, active, selected, (long long *)active_mask, (const long long *)gt, picks, ws, g);
    else
        hipLaunchKernelGGL(k_sel_apply<float>, ga, blk, 0, st, (float *)score, active, selected, (long long *)active_mask, (const long long *)gt, picks, ws, g);
    *hdr_out = ws.hdr;
    return check_launch("halo_greedy_select (binned)");
}

}  // namespace halo
