// Shared pieces of the two selector implementations (halo_select.hip: serial tile-table kernel,
// halo_select_binned.hip: value-binned sweep): the total order on score values and the wave-level
// arg-max under that order.
//
// Order (core/active/build.py:38-43, the reference's two-stage torch.max): value descending, then
// smallest w, then smallest h; NaN above everything; -0 == +0.  Values are compared as ordered
// 64-bit integers, so every decision is a bit-exact function of the score map for float32 and
// float64 maps alike.
#pragma once
#include "halo_common.hpp"

namespace halo {

constexpr unsigned long long KEY_NAN = 0xffffffffffffffffull;
constexpr unsigned long long KEY_NEG_INF = 0x000fffffffffffffull;   // ~bits(-inf)
constexpr unsigned long long KEY_POS_INF = 0xfff0000000000000ull;   // bits(+inf) | sign

__device__ __forceinline__ unsigned long long order_key(double v)
{
    const bool isnan = v != v;
    v = v == 0.0 ? 0.0 : v;                                 // -0 ties with +0 in torch.max
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned long long k = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return isnan ? KEY_NAN : k;
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    if (k == KEY_NAN) return __longlong_as_double(0x7ff8000000000000ll);
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

struct Cand { unsigned long long key; unsigned pos; };     // pos = (w << 16) | h  (smaller wins a tie: min w, then min h)

__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{
    return a.key > b.key || (a.key == b.key && a.pos < b.pos);
}
// Wave-wide unsigned max via DPP row shifts / row broadcasts (no LDS crossbar traffic): 6 dependent
// VALU+DPP steps instead of 6 ds_bpermute round trips.  Result is wave-uniform (read from lane 63).
__device__ __forceinline__ unsigned wave_umax(unsigned x)
{
#define HALO_DPP_MAX(ctrl, rmask)                                                                    \
    {                                                                                                \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, true); \
        x = o > x ? o : x;                                                                           \
    }
    HALO_DPP_MAX(0x111, 0xf)   // row_shr:1
    HALO_DPP_MAX(0x112, 0xf)   // row_shr:2
    HALO_DPP_MAX(0x114, 0xf)   // row_shr:4
    HALO_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds its row's max
    HALO_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    HALO_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's max
#undef HALO_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// arg-max over a wave under `better`: high word, low word, then the smallest position among ties
__device__ __forceinline__ Cand wave_best(Cand c)
{
    const unsigned hi = (unsigned)(c.key >> 32), lo = (unsigned)c.key;
    const unsigned mh = wave_umax(hi);
    const unsigned ml = wave_umax(hi == mh ? lo : 0u);
    const bool tie = hi == mh && lo == ml;
    const unsigned mp = ~wave_umax(tie ? ~c.pos : 0u);
    Cand r;
    r.key = ((unsigned long long)mh << 32) | ml;
    r.pos = mp;
    return r;
}

// LDS-only barrier: orders LDS traffic between the waves of the workgroup without draining the
// vector-memory counter (a __syncthreads() would wait for outstanding global stores / prefetches).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-image control block of the binned selector (also read by the serial kernel when it resumes an image).
struct SelHdr {
    unsigned long long kmin_inv;   // max over finite values of ~order_key  (zero-initialised => "no value yet")
    unsigned long long kmax;       // max over finite values of order_key
    unsigned nvalid;               // finite values (> -inf): what can ever be picked
    unsigned flags;                // SEL_F_*
    unsigned ncand;                // candidates appended to the staging list
    unsigned nf;                   // fine bins in use
    unsigned t1;                   // first coarse bin that contributes candidates
    unsigned truncated;            // 1: the threshold bin was dropped (staging capacity) -> exhaustion is not final
    int status;                    // SEL_RUN / SEL_DONE / SEL_BAIL
    int np;                        // picks made so far
    unsigned pad[4];
};
static_assert(sizeof(SelHdr) == 64, "SelHdr is 64 bytes");
enum { SEL_RUN = 0, SEL_DONE = 1, SEL_BAIL = 2 };
enum { SEL_F_BAD = 1,              // NaN or +inf present: the value range cannot be binned
       SEL_F_OVERFLOW = 2,         // a fine bin received more candidates than it has slots (a plateau of ties)
       SEL_F_HIST = 4 };           // (range records of the scorer) the coarse histogram behind the records is valid for this image

constexpr int NB1 = 2048;          // coarse bins of the value-binned selector

// A range record buffer (halo_score_range_bytes): B records, then -- 256-byte aligned -- B coarse histograms of NB1 counters.  The
// scorer's fused tail fills a histogram while it writes a normalised score map (k_combine_box3) and sets SEL_F_HIST; the selector
// then skips its own pass over the map (k_sel_hist1) and CONSUMES the flag (k_sel_scan1 clears it: the counts describe the map as
// it was when it was scored).
__host__ __device__ inline size_t range_hist_offset(long long B) { return ((size_t)B * sizeof(SelHdr) + 255) / 256 * 256; }

struct ValRange { double lo, scale; bool ok; };

__device__ __forceinline__ ValRange sel_range(const SelHdr &h)
{
    ValRange r;
    const double lo = key_value(~h.kmin_inv), hi = key_value(h.kmax);
    r.lo = lo;
    r.scale = (double)NB1 / (hi - lo);
    r.ok = !(h.flags & SEL_F_BAD) && h.nvalid > 0 && hi > lo && r.scale > 0.0 && r.scale < 1.0e300;
    return r;
}

// coarse bin of a finite value v; t = position in bin units (monotone, non-decreasing in v).  The range record only has to bound
// the values for the bins to be well filled, NOT for correctness: a value below `lo` is clamped into the lowest bin explicitly
// (t >= 0: nothing relies on how a negative double converts), one above `hi` into the highest, and the sub-bin of such a value
// is clamped by its caller -- binning stays monotone, so the picks do not depend on the record.
__device__ __forceinline__ int coarse_bin(double v, const ValRange &r, double &t)
{
    t = (v - r.lo) * r.scale;
    t = t > 0.0 ? t : 0.0;
    const int j = t < (double)(NB1 - 1) ? (int)t : NB1 - 1;
    return j;
}

}  // namespace halo
