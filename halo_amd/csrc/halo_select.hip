// select_pixels_to_label (core/active/build.py:27-64) as one persistent workgroup per image.
//
// The reference runs, per region, two full-map torch.max reductions plus >= 3 device->host
// .item() syncs and four slice writes: 2331 dependent iterations per 1024x2048 image.  Here the
// whole loop lives on the device: the score map stays in HBM/L2, the workgroup keeps a
// tile-maximum table (one {ordered key, position} entry per TH x TW tile) in LDS, and every
// step is  workgroup argmax over per-thread cached tile bests -> suppress the (2*mask+1)^2
// window -> re-reduce only the <= 4 tiles the window touches.  No host round trip, no
// re-scan of the map.  Images are independent, so a batch of B images runs as B workgroups
// that overlap with the bandwidth-bound scoring kernels of the next batch.
//
// Exactness: candidates are ordered by (value descending, w ascending, h ascending) with
// NaN above everything and -0 == +0 -- the order the reference's two-stage
// `torch.max(score, dim=0)` / `torch.max(values, dim=0)` produces (first occurrence wins).
// Values are compared as 64-bit ordered integers, so the picks are bit-exact functions of
// the score map for both float32 and float64 maps.
#include "halo_common.hpp"

namespace halo {

constexpr int SEL_TPB = 512;
constexpr int SEL_WAVES = SEL_TPB / 64;
constexpr unsigned long long KEY_NAN = 0xffffffffffffffffull;
constexpr unsigned long long KEY_NEG_INF = 0x000fffffffffffffull;   // ~bits(-inf)

__device__ __forceinline__ unsigned long long order_key(double v)
{
    if (v != v) return KEY_NAN;
    if (v == 0.0) v = 0.0;                                  // -0 ties with +0 in torch.max
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    if (k == KEY_NAN) return __longlong_as_double(0x7ff8000000000000ll);
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

struct Cand { unsigned long long key; unsigned pos; };     // pos = w*H + h  (smaller wins a tie)

__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{
    return a.key > b.key || (a.key == b.key && a.pos < b.pos);
}
__device__ __forceinline__ Cand wave_best(Cand c)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.key = __shfl_xor(c.key, off);
        o.pos = __shfl_xor(c.pos, off);
        if (better(o, c)) c = o;
    }
    return c;
}

struct SelGeom { int H, W, th_shift, tw_shift, nty, ntx, nt; };

// Reduce one tile (cooperatively, one wave) -> lane-uniform Cand.  Pixels inside the window
// [wy0,wy1] x [wx0,wx1] count as -inf (the suppression being applied in this same step).
// The tile shape is a compile-time constant so that all of a lane's loads are issued
// back to back (one memory round trip per tile instead of one per pixel row).
template <typename T, int TSH, int TSW>
__device__ __forceinline__ Cand tile_reduce(const T *__restrict__ sc, const SelGeom &g, int tile, int lane,
                                            int wy0, int wy1, int wx0, int wx1)
{
    constexpr int TW = 1 << TSW, NPL = (1 << (TSH + TSW)) / 64;
    static_assert(NPL >= 1, "tile smaller than a wave");
    const int ty = tile / g.ntx, tx = tile % g.ntx;
    const int y0 = ty << TSH, x0 = tx << TSW;
    double v[NPL];
    bool inb[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        const int y = y0 + (e >> TSW), x = x0 + (e & (TW - 1));
        inb[i] = y < g.H && x < g.W;
        v[i] = inb[i] ? (double)sc[(size_t)y * g.W + x] : 0.0;
    }
    Cand best;
    best.key = 0ull;            // below every real key (real keys are >= KEY_NEG_INF > 0)
    best.pos = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        const int y = y0 + (e >> TSW), x = x0 + (e & (TW - 1));
        Cand c;
        c.key = (y >= wy0 && y <= wy1 && x >= wx0 && x <= wx1) ? KEY_NEG_INF : order_key(v[i]);
        c.pos = (unsigned)x * (unsigned)g.H + (unsigned)y;
        if (inb[i] && better(c, best)) best = c;
    }
    return wave_best(best);
}

template <typename T, int TSH, int TSW>
__global__ void __launch_bounds__(SEL_TPB) k_greedy_select(T *__restrict__ score, SelGeom g, int n_regions, int arad,
                                                           int mrad, unsigned char *__restrict__ active,
                                                           unsigned char *__restrict__ selected,
                                                           long long *__restrict__ active_mask,
                                                           const long long *__restrict__ gt, double *__restrict__ picks,
                                                           int *__restrict__ n_picked)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *tkey = reinterpret_cast<unsigned long long *>(smem);
    unsigned *tpos = reinterpret_cast<unsigned *>(smem + (size_t)g.nt * 8);
    // wave exchange buffers, double-buffered by step parity
    unsigned long long *wkey = reinterpret_cast<unsigned long long *>(smem + (size_t)g.nt * 12 + ((16 - ((size_t)g.nt * 12) % 16) % 16));
    unsigned *wpos = reinterpret_cast<unsigned *>(wkey + 2 * SEL_WAVES);

    // Latency-bound serial loop sharing CUs with bandwidth-bound streaming kernels: take issue
    // priority over them (they only wait on memory anyway).
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t hw = (size_t)g.H * g.W;
    T *sc = score + (size_t)b * hw;
    unsigned char *act = active + (size_t)b * hw;
    unsigned char *sel = selected + (size_t)b * hw;
    long long *am = active_mask + (size_t)b * hw;
    const long long *gtb = gt + (size_t)b * hw;

    // ---- build the tile table
    for (int t = wave; t < g.nt; t += SEL_WAVES) {
        const Cand c = tile_reduce<T, TSH, TSW>(sc, g, t, lane, 1, 0, 1, 0);
        if (lane == 0) { tkey[t] = c.key; tpos[t] = c.pos; }
    }
    __syncthreads();

    // thread-local best over the tiles this thread owns (t = tid, tid+SEL_TPB, ...)
    Cand mine;
    auto rescan = [&]() {
        mine.key = 0ull;
        mine.pos = 0xffffffffu;
        for (int t = tid; t < g.nt; t += SEL_TPB) {
            Cand c;
            c.key = tkey[t];
            c.pos = tpos[t];
            if (better(c, mine)) mine = c;
        }
    };
    rescan();

    int np = 0;
    for (int it = 0; it < n_regions; ++it) {
        // ---- workgroup argmax
        const Cand wb = wave_best(mine);
        const int par = (it & 1) * SEL_WAVES;
        if (lane == 0) { wkey[par + wave] = wb.key; wpos[par + wave] = wb.pos; }
        __syncthreads();
        Cand top;
        top.key = wkey[par];
        top.pos = wpos[par];
#pragma unroll
        for (int i = 1; i < SEL_WAVES; ++i) {
            Cand c;
            c.key = wkey[par + i];
            c.pos = wpos[par + i];
            if (better(c, top)) top = c;
        }
        if (top.key == KEY_NEG_INF || top.key == 0ull) break;            // build.py:40-41
        const int w = (int)(top.pos / (unsigned)g.H), h = (int)(top.pos % (unsigned)g.H);

        // ---- windows (build.py:45-53): low side clipped at 0, high side by the slice
        const int my0 = h - mrad < 0 ? 0 : h - mrad, my1 = h + mrad >= g.H ? g.H - 1 : h + mrad;
        const int mx0 = w - mrad < 0 ? 0 : w - mrad, mx1 = w + mrad >= g.W ? g.W - 1 : w + mrad;
        const int ay0 = h - arad < 0 ? 0 : h - arad, ay1 = h + arad >= g.H ? g.H - 1 : h + arad;
        const int ax0 = w - arad < 0 ? 0 : w - arad, ax1 = w + arad >= g.W ? g.W - 1 : w + arad;
        if (tid == 0 && picks) {
            double *pk = picks + ((size_t)b * n_regions + np) * 3;
            pk[0] = (double)h;
            pk[1] = (double)w;
            pk[2] = key_value(top.key);
        }
        ++np;
        {   // score[...] = -inf ; active[...] = True   (build.py:56-57)
            const int mw = mx1 - mx0 + 1, mn = mw * (my1 - my0 + 1);
            for (int e = tid; e < mn; e += SEL_TPB) {
                const size_t o = (size_t)(my0 + e / mw) * g.W + (mx0 + e % mw);
                if constexpr (sizeof(T) == 8) sc[o] = (T)__longlong_as_double(0xfff0000000000000ll);
                else sc[o] = (T)__uint_as_float(0xff800000u);
                act[o] = 1;
            }
            // selected[...] = True ; active_mask[...] = ground_truth[...]   (build.py:58-62)
            const int aw = ax1 - ax0 + 1, an = aw * (ay1 - ay0 + 1);
            for (int e = tid; e < an; e += SEL_TPB) {
                const size_t o = (size_t)(ay0 + e / aw) * g.W + (ax0 + e % aw);
                sel[o] = 1;
                am[o] = gtb[o];
            }
        }
        // ---- re-reduce the tiles the mask window touches (one wave per tile)
        const int ty0 = my0 >> TSH, ty1 = my1 >> TSH, tx0 = mx0 >> TSW, tx1 = mx1 >> TSW;
        const int ntx_w = tx1 - tx0 + 1, ntouch = ntx_w * (ty1 - ty0 + 1);
        for (int q = wave; q < ntouch; q += SEL_WAVES) {
            const int t = (ty0 + q / ntx_w) * g.ntx + (tx0 + q % ntx_w);
            const Cand c = tile_reduce<T, TSH, TSW>(sc, g, t, lane, my0, my1, mx0, mx1);
            if (lane == 0) { tkey[t] = c.key; tpos[t] = c.pos; }
        }
        __syncthreads();
        // owners of touched tiles refresh their cached best
        bool own = false;
        for (int q = 0; q < ntouch; ++q) {
            const int t = (ty0 + q / ntx_w) * g.ntx + (tx0 + q % ntx_w);
            own |= (t % SEL_TPB) == tid;
        }
        if (own) rescan();
    }
    if (tid == 0 && n_picked) n_picked[b] = np;
}

}  // namespace halo

using namespace halo;

static SelGeom make_geom(int64_t H, int64_t W)
{
    SelGeom g;
    g.H = (int)H;
    g.W = (int)W;
    // 16 x 32 tiles: 4096 tiles (48 KiB of LDS) at 1024 x 2048; larger maps use 32 x 64 / 64 x 128
    static const int shapes[3][2] = {{4, 5}, {5, 6}, {6, 7}};
    for (int i = 0; i < 3; ++i) {
        g.th_shift = shapes[i][0];
        g.tw_shift = shapes[i][1];
        g.nty = (int)cdiv(H, 1 << g.th_shift);
        g.ntx = (int)cdiv(W, 1 << g.tw_shift);
        g.nt = g.nty * g.ntx;
        if ((size_t)g.nt * 12 <= 96 * 1024) break;
    }
    return g;
}

extern "C" size_t halo_select_workspace_bytes(int64_t B, int64_t H, int64_t W)
{
    (void)B; (void)H; (void)W;
    return 256;   // the tile table lives in LDS; nothing is needed in HBM today
}

extern "C" int halo_greedy_select(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                                  int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                                  int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes;
    hipStream_t st = (hipStream_t)stream;
    if (!score || !active || !selected || !active_mask || !gt || B <= 0 || H <= 0 || W <= 0)
        return fail(HALO_E_ARG, "halo_greedy_select: null/empty argument");
    if (dtype != HALO_F32 && dtype != HALO_F64) return fail(HALO_E_ARG, "halo_greedy_select: bad dtype");
    if (n_regions < 0 || active_radius < 0 || mask_radius < 0) return fail(HALO_E_ARG, "halo_greedy_select: negative parameter");
    if ((uint64_t)H * (uint64_t)W >= 0xffffffffull) return fail(HALO_E_UNSUPPORTED, "halo_greedy_select: image too large");
    if (n_regions > 0x7fffffff) n_regions = 0x7fffffff;
    const SelGeom g = make_geom(H, W);
    if ((size_t)g.nt * 12 > 96 * 1024) return fail(HALO_E_UNSUPPORTED, "halo_greedy_select: image too large for the tile table");
    const size_t lds = align_up((size_t)g.nt * 12, 16) + 2 * SEL_WAVES * 12 + 64;
    dim3 grid((unsigned)B), block(SEL_TPB);
#define HALO_SEL_LAUNCH(T, A, B_)                                                                                          \
    hipLaunchKernelGGL((k_greedy_select<T, A, B_>), grid, block, lds, st, (T *)score, g, (int)n_regions, (int)active_radius, \
                       (int)mask_radius, active, selected, (long long *)active_mask, (const long long *)gt, picks, n_picked)
    if (dtype == HALO_F64) {
        if (g.th_shift == 4) HALO_SEL_LAUNCH(double, 4, 5);
        else if (g.th_shift == 5) HALO_SEL_LAUNCH(double, 5, 6);
        else HALO_SEL_LAUNCH(double, 6, 7);
    } else {
        if (g.th_shift == 4) HALO_SEL_LAUNCH(float, 4, 5);
        else if (g.th_shift == 5) HALO_SEL_LAUNCH(float, 5, 6);
        else HALO_SEL_LAUNCH(float, 6, 7);
    }
#undef HALO_SEL_LAUNCH
    return check_launch("halo_greedy_select");
}
