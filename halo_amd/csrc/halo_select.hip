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
#include "halo_select_common.hpp"

namespace halo {

constexpr int SEL_TPB_MAIN = 512;        // the selector proper: one workgroup per image
constexpr int SEL_TPB_RESUME = 256;      // behind the binned sweep: see k_greedy_resume

struct SelGeom { int H, W, th_shift, tw_shift, nty, ntx, nt; };

// Reduce one tile (cooperatively, one wave) -> lane-uniform Cand.  Pixels inside the window
// [wy0,wy1] x [wx0,wx1] count as -inf (the suppression being applied in this same step).
// The tile shape is a compile-time constant so that all of a lane's loads are issued
// back to back (one memory round trip per tile instead of one per pixel row).
template <typename T, int TSH, int TSW>
__device__ __forceinline__ Cand tile_reduce(const T *__restrict__ sc, const SelGeom &g, int ty, int tx, int lane,
                                            int wy0, int wy1, int wx0, int wx1, int py0 = 1, int py1 = 0, int px0 = 1,
                                            int px1 = 0)
{
    constexpr int TW = 1 << TSW, NPL = (1 << (TSH + TSW)) / 64;
    static_assert(NPL >= 1, "tile smaller than a wave");
    // element e = lane + 64*i of the tile sits at (row e >> TSW, column e & (TW-1)); for TW <= 64 a lane
    // keeps ONE column and walks rows, so the column tests are hoisted out of the loop by the compiler
    const int x0 = tx << TSW, y0 = ty << TSH;
    double v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        const int y = y0 + (e >> TSW), x = x0 + (e & (TW - 1));
        const int yc = y < g.H ? y : g.H - 1, xc = x < g.W ? x : g.W - 1;   // clamped: the load is unconditional
        v[i] = (double)sc[(size_t)yc * g.W + xc];                         // (no exec-mask juggling per element)
    }
    Cand best;
    best.key = 0ull;            // below every real key (real keys are >= KEY_NEG_INF > 0)
    best.pos = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        const int y = y0 + (e >> TSW), x = x0 + (e & (TW - 1));
        const bool masked = (x >= wx0 && x <= wx1 && y >= wy0 && y <= wy1) || (x >= px0 && x <= px1 && y >= py0 && y <= py1);
        const bool inb = x < g.W && y < g.H;
        Cand c;
        c.key = inb ? (masked ? KEY_NEG_INF : order_key(v[i])) : 0ull;    // out-of-image: below every real key
        c.pos = ((unsigned)x << 16) | (unsigned)y;
        const bool take = better(c, best);
        best.key = take ? c.key : best.key;
        best.pos = take ? c.pos : best.pos;
    }
    return wave_best(best);
}

// The selector's body for a workgroup of TPB threads (WAVES - 1 reducer waves + 1 writer wave).
template <typename T, int TSH, int TSW, int EPT, int SEL_TPB>
__device__ __forceinline__ void greedy_select_body(T *__restrict__ score, const SelGeom &g, int n_regions, int arad, int mrad,
                                                   unsigned char *__restrict__ active, unsigned char *__restrict__ selected,
                                                   long long *__restrict__ active_mask, const long long *__restrict__ gt,
                                                   double *__restrict__ picks, int *__restrict__ n_picked,
                                                   const SelHdr *__restrict__ resume, int n_images)
{
    constexpr int SEL_WAVES = SEL_TPB / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *tkey = reinterpret_cast<unsigned long long *>(smem);
    unsigned *tpos = reinterpret_cast<unsigned *>(smem + (size_t)g.nt * 8);
    // wave exchange buffers, double-buffered by step parity
    unsigned long long *wkey = reinterpret_cast<unsigned long long *>(smem + (size_t)g.nt * 12 + ((16 - ((size_t)g.nt * 12) % 16) % 16));
    unsigned *wpos = reinterpret_cast<unsigned *>(wkey + 2 * SEL_WAVES);

    // Latency-bound serial loop sharing CUs with bandwidth-bound streaming kernels: take issue
    // priority over them (they only wait on memory anyway).
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t hw = (size_t)g.H * g.W;
    // A workgroup serves images blockIdx.x, blockIdx.x + gridDim.x, ...: one image each when this kernel is the selector;
    // behind the binned selector (halo_select_binned.hip) the launch has only a couple of workgroups, which skip the
    // images the sweep finished (normally all of them) and continue the ones it handed over from pick `np`, on the map
    // as the reference would have it after those picks.  (A 512-thread workgroup with a 48+ KiB table is placed
    // beside a saturating streaming kernel only when a CU drains: 16 of them took 6-7 ms to place and exit.)
    for (int b = blockIdx.x; b < n_images; b += gridDim.x) {
    int np0 = 0;
    if (resume) {
        if (resume[b].status == SEL_DONE) continue;
        np0 = resume[b].np;
    }
    T *sc = score + (size_t)b * hw;
    unsigned char *act = active + (size_t)b * hw;
    unsigned char *sel = selected + (size_t)b * hw;
    long long *am = active_mask + (size_t)b * hw;
    const long long *gtb = gt + (size_t)b * hw;

    // ---- build the tile table
    for (int ty = 0; ty < g.nty; ++ty)
        for (int tx = wave; tx < g.ntx; tx += SEL_WAVES) {
            const Cand c = tile_reduce<T, TSH, TSW>(sc, g, ty, tx, lane, 1, 0, 1, 0);
            if (lane == 0) { tkey[ty * g.ntx + tx] = c.key; tpos[ty * g.ntx + tx] = c.pos; }
        }
    __syncthreads();

    // thread-local best over the tiles this thread owns (t = tid, tid+SEL_TPB, ...)
    Cand mine;
    auto rescan = [&]() {      // EPT entries per thread; clamped unconditional LDS reads issued back to back
        Cand e[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int t = tid + j * SEL_TPB;
            const bool ok = t < g.nt;
            const int tc = ok ? t : g.nt - 1;
            const unsigned long long k = tkey[tc];
            const unsigned q = tpos[tc];
            e[j].key = ok ? k : 0ull;
            e[j].pos = ok ? q : 0xffffffffu;
        }
        mine = e[0];
#pragma unroll
        for (int j = 1; j < EPT; ++j) {
            const bool take = better(e[j], mine);
            mine.key = take ? e[j].key : mine.key;
            mine.pos = take ? e[j].pos : mine.pos;
        }
    };
    rescan();

    // Per step: [A] workgroup arg-max -> [B] wave 7 writes the windows while waves 0..6 re-reduce the
    // touched tiles -> [C] owners refresh.  Window stores are NOT waited for inside the step: the
    // writer wave drains them at the start of the NEXT step's phase B (they had a whole step to
    // land), and tile re-reductions mask the current AND the previous window analytically, so a
    // tile load never depends on a store younger than two steps (memory-model argument in DESIGN.md).
    int np = np0;
    int py0 = 1, py1 = 0, px0 = 1, px1 = 0;                              // previous window (empty)
    const T neg_inf = sizeof(T) == 8 ? (T)__longlong_as_double(0xfff0000000000000ll) : (T)__uint_as_float(0xff800000u);
    for (int it = np0; it < n_regions; ++it) {
        // ---- [A] workgroup argmax
        const Cand wb = wave_best(mine);
        const int par = (it & 1) * SEL_WAVES;
        if (lane == 0) { wkey[par + wave] = wb.key; wpos[par + wave] = wb.pos; }
        lds_barrier();
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
        const int w = (int)(top.pos >> 16), h = (int)(top.pos & 0xffffu);

        // windows (build.py:45-53): low side clipped at 0, high side by the slice
        const int my0 = h - mrad < 0 ? 0 : h - mrad, my1 = h + mrad >= g.H ? g.H - 1 : h + mrad;
        const int mx0 = w - mrad < 0 ? 0 : w - mrad, mx1 = w + mrad >= g.W ? g.W - 1 : w + mrad;
        const int ty0 = my0 >> TSH, ty1 = my1 >> TSH, tx0 = mx0 >> TSW, tx1 = mx1 >> TSW;

        // ---- [B]
        if (wave == SEL_WAVES - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the previous step's stores have landed
            const int ay0 = h - arad < 0 ? 0 : h - arad, ay1 = h + arad >= g.H ? g.H - 1 : h + arad;
            const int ax0 = w - arad < 0 ? 0 : w - arad, ax1 = w + arad >= g.W ? g.W - 1 : w + arad;
            const int aw = ax1 - ax0 + 1, an = aw * (ay1 - ay0 + 1);
            const int mw = mx1 - mx0 + 1, mn = mw * (my1 - my0 + 1);
            // selected[...] = True ; active_mask[...] = ground_truth[...]   (build.py:58-62): loads first
            for (int e0 = 0; e0 < an; e0 += 64) {
                const int e = e0 + lane;
                if (e < an) {
                    const size_t o = (size_t)(ay0 + e / aw) * g.W + (ax0 + e % aw);
                    const long long gv = gtb[o];
                    sel[o] = 1;
                    am[o] = gv;
                }
            }
            // score[...] = -inf ; active[...] = True   (build.py:56-57)
            for (int e = lane; e < mn; e += 64) {
                const size_t o = (size_t)(my0 + e / mw) * g.W + (mx0 + e % mw);
                sc[o] = neg_inf;
                act[o] = 1;
            }
            if (lane == 0 && picks) {
                double *pk = picks + ((size_t)b * n_regions + np) * 3;
                pk[0] = (double)h;
                pk[1] = (double)w;
                pk[2] = key_value(top.key);
            }
        } else {
            int q = 0;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx, ++q) {
                    if (q % (SEL_WAVES - 1) != wave) continue;
                    const Cand c = tile_reduce<T, TSH, TSW>(sc, g, ty, tx, lane, my0, my1, mx0, mx1, py0, py1, px0, px1);
                    if (lane == 0) { tkey[ty * g.ntx + tx] = c.key; tpos[ty * g.ntx + tx] = c.pos; }
                }
        }
        ++np;
        lds_barrier();
        // ---- [C] owners of touched tiles refresh their cached best
        bool own = false;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) own |= ((ty * g.ntx + tx) & (SEL_TPB - 1)) == tid;
        if (own) rescan();
        py0 = my0; py1 = my1; px0 = mx0; px1 = mx1;
    }
    if (tid == 0 && n_picked) n_picked[b] = np;
    __syncthreads();                                                     // the tables are rebuilt for the next image
    }
}

template <typename T, int TSH, int TSW, int EPT>
__global__ void __launch_bounds__(SEL_TPB_MAIN) k_greedy_select(T *__restrict__ score, SelGeom g, int n_regions, int arad, int mrad,
                                                                unsigned char *__restrict__ active, unsigned char *__restrict__ selected,
                                                                long long *__restrict__ active_mask, const long long *__restrict__ gt,
                                                                double *__restrict__ picks, int *__restrict__ n_picked,
                                                                const SelHdr *__restrict__ resume, int n_images)
{
    greedy_select_body<T, TSH, TSW, EPT, SEL_TPB_MAIN>(score, g, n_regions, arad, mrad, active, selected, active_mask, gt, picks, n_picked,
                                                        resume, n_images);
}

// The same selector as the stage behind the binned sweep, where it normally finds every image finished: 256 threads --
// one wave per SIMD -- held to 96 VGPRs (amdgpu_waves_per_eu: the compiler spills what does not fit; this path is rare
// and slow anyway), which is what k_feat_reduce's cap leaves free on every SIMD, so the workgroup is placed at once
// instead of waiting milliseconds for a CU to drain.
template <typename T, int TSH, int TSW, int EPT>
__global__ void __launch_bounds__(SEL_TPB_RESUME) __attribute__((amdgpu_waves_per_eu(5)))
k_greedy_resume(T *__restrict__ score, SelGeom g, int n_regions, int arad, int mrad, unsigned char *__restrict__ active,
                unsigned char *__restrict__ selected, long long *__restrict__ active_mask, const long long *__restrict__ gt,
                double *__restrict__ picks, int *__restrict__ n_picked, const SelHdr *__restrict__ resume, int n_images)
{
    greedy_select_body<T, TSH, TSW, EPT, SEL_TPB_RESUME>(score, g, n_regions, arad, mrad, active, selected, active_mask, gt, picks,
                                                          n_picked, resume, n_images);
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

#include "halo_select_plan.hpp"

static bool serial_supported(int64_t H, int64_t W)
{
    if (H > 65535 || W > 65535) return false;
    const SelGeom g = make_geom(H, W);
    return (size_t)g.nt * 12 <= 96 * 1024;
}

extern "C" size_t halo_select_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t n_regions, int64_t mask_radius)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (n_regions > H * W) n_regions = H * W;
    const BinPlan p = binned_plan(B, H, W, n_regions, 0, mask_radius);
    return p.ok ? p.total_bytes + 256 : 256;
}

template <typename T, int A, int B_, int EPT>
static int launch_serial(const SelGeom &g, size_t lds, dim3 grid, hipStream_t st, void *score, int n_regions, int arad, int mrad,
                         uint8_t *active, uint8_t *selected, int64_t *active_mask, const int64_t *gt, double *picks,
                         int32_t *n_picked, const SelHdr *resume, int n_images)
{
    static LdsLimitSeen seen;           // per instantiation: tile tables above 64 KiB need the dynamic-LDS limit raised
    if (lds > 64 * 1024 && !raise_lds_limit(seen, (const void *)k_greedy_select<T, A, B_, EPT>, 128 * 1024))
        return fail(HALO_E_LAUNCH, "halo_greedy_select: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((k_greedy_select<T, A, B_, EPT>), grid, dim3(SEL_TPB_MAIN), lds, st, (T *)score, g, n_regions, arad, mrad, active,
                       selected, (long long *)active_mask, (const long long *)gt, picks, n_picked, resume, n_images);
    return HALO_OK;
}

template <typename T, int A, int B_, int EPT>
static int launch_resume(const SelGeom &g, size_t lds, dim3 grid, hipStream_t st, void *score, int n_regions, int arad, int mrad,
                         uint8_t *active, uint8_t *selected, int64_t *active_mask, const int64_t *gt, double *picks,
                         int32_t *n_picked, const SelHdr *resume, int n_images)
{
    static LdsLimitSeen seen;
    if (lds > 64 * 1024 && !raise_lds_limit(seen, (const void *)k_greedy_resume<T, A, B_, EPT>, 128 * 1024))
        return fail(HALO_E_LAUNCH, "halo_greedy_select: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((k_greedy_resume<T, A, B_, EPT>), grid, dim3(SEL_TPB_RESUME), lds, st, (T *)score, g, n_regions, arad, mrad, active,
                       selected, (long long *)active_mask, (const long long *)gt, picks, n_picked, resume, n_images);
    return HALO_OK;
}

extern "C" int halo_greedy_select(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                                  int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                                  int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                                  void *workspace, size_t workspace_bytes, int method, void *stream)
{
    return halo_greedy_select_ranged(score, dtype, B, H, W, n_regions, active_radius, mask_radius, active, selected, active_mask, gt,
                                     picks, n_picked, workspace, workspace_bytes, method, nullptr, stream);
}

extern "C" size_t halo_score_range_bytes(int64_t B) { return B > 0 ? range_hist_offset(B) + (size_t)B * NB1 * sizeof(unsigned) : 0; }

extern "C" int halo_score_range(const void *score, int dtype, int64_t B, int64_t H, int64_t W, void *score_range, void *stream)
{
    if (!score || !score_range || B <= 0 || H <= 0 || W <= 0) return fail(HALO_E_ARG, "halo_score_range: null/empty argument");
    if (dtype != HALO_F32 && dtype != HALO_F64) return fail(HALO_E_ARG, "halo_score_range: bad dtype");
    const int rc = score_range_exact(score, dtype, B, H * W, score_range, (hipStream_t)stream);
    return rc != HALO_OK ? rc : check_launch("halo_score_range");
}

extern "C" int halo_greedy_select_ranged(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                                         int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                                         int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                                         void *workspace, size_t workspace_bytes, int method, const void *score_range, void *stream)
{
    return halo_greedy_select_ex(score, dtype, B, H, W, n_regions, active_radius, mask_radius, active, selected, active_mask, gt, picks,
                                 n_picked, workspace, workspace_bytes, method, score_range, nullptr, stream);
}

namespace halo {
// the sweep did not run for these images (serial method, or a geometry it does not serve): {HALO_SWEEP_NOT_RUN, 0} rows
__global__ void __launch_bounds__(256) k_sel_not_run(int *__restrict__ handover, int n_images)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < n_images) { handover[2 * b] = HALO_SWEEP_NOT_RUN; handover[2 * b + 1] = 0; }
}
}  // namespace halo

extern "C" int halo_greedy_select_ex(void *score, int dtype, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                                     int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                                     int64_t *active_mask, const int64_t *gt, double *picks, int32_t *n_picked,
                                     void *workspace, size_t workspace_bytes, int method, const void *score_range,
                                     int32_t *handover, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!score || !active || !selected || !active_mask || !gt || B <= 0 || H <= 0 || W <= 0)
        return fail(HALO_E_ARG, "halo_greedy_select: null/empty argument");
    if (dtype != HALO_F32 && dtype != HALO_F64) return fail(HALO_E_ARG, "halo_greedy_select: bad dtype");
    if (n_regions < 0 || active_radius < 0 || mask_radius < 0) return fail(HALO_E_ARG, "halo_greedy_select: negative parameter");
    if (method < HALO_SELECT_AUTO || method > HALO_SELECT_BINNED) return fail(HALO_E_ARG, "halo_greedy_select: bad method");
    if (H > 65535 || W > 65535) return fail(HALO_E_UNSUPPORTED, "halo_greedy_select: image side above 65535");
    if (n_regions > H * W) n_regions = H * W;            // there are no more pixels than that to pick
    if (n_regions == 0) {
        if (n_picked && hipMemsetAsync(n_picked, 0, (size_t)B * 4, st) != hipSuccess) return fail(HALO_E_LAUNCH, "halo_greedy_select: memset failed");
        if (handover && hipMemsetAsync(handover, 0, (size_t)B * 8, st) != hipSuccess) return fail(HALO_E_LAUNCH, "halo_greedy_select: memset failed");
        return HALO_OK;
    }
    if (!serial_supported(H, W)) return fail(HALO_E_UNSUPPORTED, "halo_greedy_select: image too large for the tile table");
    const SelGeom g = make_geom(H, W);

    // ---- the binned sweep first (unless the serial kernel was asked for or the pick grid does not fit LDS)
    const SelHdr *resume = nullptr;
    if (method != HALO_SELECT_SERIAL) {
        const BinPlan p = binned_plan(B, H, W, n_regions, active_radius, mask_radius);
        if (p.ok) {
            SelHdr *hdr = nullptr;
            const int rc = binned_select(score, dtype, B, p, active, selected, active_mask, gt, picks, n_picked, workspace,
                                         workspace_bytes, st, &hdr, score_range, handover);
            if (rc != HALO_OK) return rc;
            resume = hdr;
        } else if (method == HALO_SELECT_BINNED)
            return fail(HALO_E_UNSUPPORTED, "halo_greedy_select: the binned selector does not serve this geometry "
                                            "(mask radius above 14 or pick grid larger than LDS)");
    }

    if (handover && !resume)
        hipLaunchKernelGGL(k_sel_not_run, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, handover, (int)B);

    // ---- serial tile-table kernel: the whole job, or only the images the sweep handed over
    const size_t lds = align_up((size_t)g.nt * 12, 16) + 2 * (SEL_TPB_MAIN / 64) * 12 + 64;
    // behind the sweep: a few workgroups walk the images (see the kernel), skipping the finished ones; HALO_SEL_RESUME_WGS
    // overrides (tuning aid).  The 256-thread, 96-VGPR resume kernel fits beside the streaming kernel on any CU, so its workgroups
    // are placed at once, and a degenerate round in which the sweep hands over EVERY image (NaN / +inf / constant maps,
    // plateaus of ties) is continued for all images at once.
    unsigned nwg = (unsigned)B;
    if (resume) {
        static const int wgs_env = [] { const char *e = getenv("HALO_SEL_RESUME_WGS"); return e ? atoi(e) : 0; }();
        // (round 5: one workgroup per image up to 64 -- they return at once for finished images, and a batch in which the sweep hands
        // over EVERY image, bench.py --data plateau, took two rounds of eight: 39 ms per 16-image step)
        const unsigned want = wgs_env > 0 ? (unsigned)wgs_env : 64u;
        nwg = want < (unsigned)B ? want : (unsigned)B;
    }
    dim3 grid(nwg);
    int rc = HALO_OK;
#define HALO_SEL_LAUNCH(T, A, B_)                                                                                                \
    if (resume && A < 6) /* 64 x 128 tiles (maps above 4096^2) need 256 VGPRs: they resume on the main kernel */                 \
        rc = g.nt <= 16 * SEL_TPB_RESUME                                                                                         \
                 ? launch_resume<T, A, B_, 16>(g, lds, grid, st, score, (int)n_regions, (int)active_radius, (int)mask_radius,    \
                                               active, selected, active_mask, gt, picks, n_picked, resume, (int)B)               \
                 : launch_resume<T, A, B_, 32>(g, lds, grid, st, score, (int)n_regions, (int)active_radius, (int)mask_radius,    \
                                               active, selected, active_mask, gt, picks, n_picked, resume, (int)B);              \
    else                                                                                                                         \
    rc = g.nt <= 8 * SEL_TPB_MAIN                                                                                                \
             ? launch_serial<T, A, B_, 8>(g, lds, grid, st, score, (int)n_regions, (int)active_radius, (int)mask_radius, active,    \
                                          selected, active_mask, gt, picks, n_picked, resume, (int)B)                          \
             : launch_serial<T, A, B_, 16>(g, lds, grid, st, score, (int)n_regions, (int)active_radius, (int)mask_radius, active,   \
                                           selected, active_mask, gt, picks, n_picked, resume, (int)B);
    if (dtype == HALO_F64) {
        if (g.th_shift == 4) { HALO_SEL_LAUNCH(double, 4, 5) }
        else if (g.th_shift == 5) { HALO_SEL_LAUNCH(double, 5, 6) }
        else { HALO_SEL_LAUNCH(double, 6, 7) }
    } else {
        if (g.th_shift == 4) { HALO_SEL_LAUNCH(float, 4, 5) }
        else if (g.th_shift == 5) { HALO_SEL_LAUNCH(float, 5, 6) }
        else { HALO_SEL_LAUNCH(float, 6, 7) }
    }
#undef HALO_SEL_LAUNCH
    if (rc != HALO_OK) return rc;
    return check_launch("halo_greedy_select");
}
