// Pool-side helpers of the acquisition round (core/datasets/cityscapes.py:245-251, core/active/build.py:52-62):
//   halo_pack_pick_tables   per-image pick tables -> the fixed-size int32 wire block of the round's ONE all-gather
//   halo_reset_round_state  the loader's round-1 state (active = selected = False, active_mask = 255) in one pass of
//                           16-byte stores (three torch fills ran at 0.65-1.9 TB/s)
//   halo_undo_picks         the same state restored from a pick table: only the windows the selection wrote are rewritten
//   halo_device_identity    PCI bus id + UUID of a device (ranks of one node must hold distinct devices)
//   halo_pool_alloc / _free device memory in ONE physically contiguous range (hipDeviceMallocContiguous): torch's pluggable-
//                           allocator signature, for the pools the scoring pass streams (halo_amd.pool.contiguous_memory)
//   halo_hbm_read_probe     measurement aid: a flat non-temporal streaming read of a buffer, the box's own ceiling for the
//                           bytes the feature kernel streams (bench.py reports it beside the roofline; nothing depends on it)
#include "halo_common.hpp"
#include <atomic>

namespace halo {

// wire row of one image (halo_amd/pool.py): per pick (h << 16) | w and the float64 score's two words, then the count
__global__ void __launch_bounds__(256) k_pack_tables(const double *__restrict__ picks, const int *__restrict__ n_picked, int n,
                                                     int *__restrict__ wire, long long row_stride)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int *row = wire + (size_t)b * row_stride;
    if (i < n) {
        const double *p = picks + ((size_t)b * n + i) * 3;
        const long long h = (long long)p[0], w = (long long)p[1];
        const long long s = __double_as_longlong(p[2]);
        row[3 * i + 0] = (int)((h << 16) | w);
        row[3 * i + 1] = (int)(s & 0xffffffffll);
        row[3 * i + 2] = (int)(s >> 32);
    }
    if (i == 0) row[3 * n] = n_picked[b];
}

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// a wave owns 1024 consecutive pixels: 1 KiB of `active`, 1 KiB of `selected`, 8 KiB of `active_mask`, every store
// instruction 16 bytes per lane and 1 KiB contiguous per wave
__global__ void __launch_bounds__(256) k_reset_state(unsigned char *__restrict__ active, unsigned char *__restrict__ selected,
                                                     long long *__restrict__ amask, long long n)
{
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024;      // this wave's first pixel
    if (w0 >= n) return;
    if (w0 + 1024 <= n) {
        const u4_t z = {0u, 0u, 0u, 0u}, m = {255u, 0u, 255u, 0u};
        __builtin_nontemporal_store(z, reinterpret_cast<u4_t *>(active + w0) + lane);
        __builtin_nontemporal_store(z, reinterpret_cast<u4_t *>(selected + w0) + lane);
        u4_t *am = reinterpret_cast<u4_t *>(amask + w0) + lane;       // every store instruction covers 1 KiB contiguous
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(m, am + 64 * k);
    } else {
        for (long long i = w0 + lane; i < n; i += 64) { active[i] = 0; selected[i] = 0; amask[i] = 255; }
    }
}

__global__ void __launch_bounds__(256) k_reset_state_bytes(unsigned char *__restrict__ active, unsigned char *__restrict__ selected,
                                                           long long *__restrict__ amask, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { active[i] = 0; selected[i] = 0; amask[i] = 255; }
}

// a wave per pick (a few fat workgroups per image, their waves striding over the picks: small workgroups wait to be
// placed beside the streaming kernel): the windows select_pixels_to_label wrote (build.py:52-62) go back to the round-1 state
constexpr int UNDO_WGS = 32;
__global__ void __launch_bounds__(256) k_undo_picks(const double *__restrict__ picks, const int *__restrict__ n_picked, int n, int H, int W,
                                                    int arad, int mrad, unsigned char *__restrict__ active,
                                                    unsigned char *__restrict__ selected, long long *__restrict__ amask)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int np = n_picked[b] < n ? n_picked[b] : n;
    const size_t base = (size_t)b * H * W;
    const int nwaves = gridDim.x * 4;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < np; i += nwaves) {
        const double *p = picks + ((size_t)b * n + i) * 3;
        const int h = (int)p[0], w = (int)p[1];
        const int my0 = h - mrad < 0 ? 0 : h - mrad, my1 = h + mrad >= H ? H - 1 : h + mrad;
        const int mx0 = w - mrad < 0 ? 0 : w - mrad, mx1 = w + mrad >= W ? W - 1 : w + mrad;
        const int mw = mx1 - mx0 + 1, mn = mw * (my1 - my0 + 1);
        for (int e = lane; e < mn; e += 64) active[base + (size_t)(my0 + e / mw) * W + mx0 + e % mw] = 0;
        const int ay0 = h - arad < 0 ? 0 : h - arad, ay1 = h + arad >= H ? H - 1 : h + arad;
        const int ax0 = w - arad < 0 ? 0 : w - arad, ax1 = w + arad >= W ? W - 1 : w + arad;
        const int aw = ax1 - ax0 + 1, an = aw * (ay1 - ay0 + 1);
        for (int e = lane; e < an; e += 64) {
            const size_t o = base + (size_t)(ay0 + e / aw) * W + ax0 + e % aw;
            selected[o] = 0;
            amask[o] = 255;
        }
    }
}

}  // namespace halo

using namespace halo;

extern "C" int halo_pack_pick_tables(const double *picks, const int32_t *n_picked, int64_t B, int64_t n_regions, int32_t *wire,
                                     int64_t wire_row_stride, void *stream)
{
    if (B == 0) return HALO_OK;
    if (!picks || !n_picked || !wire || B < 0 || n_regions < 0 || wire_row_stride < 3 * n_regions + 1 || B > 65535)
        return fail(HALO_E_ARG, "halo_pack_pick_tables: bad argument");
    hipLaunchKernelGGL(k_pack_tables, dim3((unsigned)(cdiv(n_regions, 256) > 0 ? cdiv(n_regions, 256) : 1), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, picks, (const int *)n_picked, (int)n_regions, (int *)wire, (long long)wire_row_stride);
    return check_launch("halo_pack_pick_tables");
}

extern "C" int halo_reset_round_state(uint8_t *active, uint8_t *selected, int64_t *active_mask, int64_t n_pixels, void *stream)
{
    if (n_pixels == 0) return HALO_OK;
    if (!active || !selected || !active_mask || n_pixels < 0) return fail(HALO_E_ARG, "halo_reset_round_state: bad argument");
    const bool wide = (((uintptr_t)active | (uintptr_t)selected | (uintptr_t)active_mask) & 15) == 0;
    if (wide)
        hipLaunchKernelGGL(k_reset_state, dim3((unsigned)cdiv(n_pixels, 4 * 1024)), dim3(256), 0, (hipStream_t)stream, active, selected,
                           (long long *)active_mask, (long long)n_pixels);
    else
        hipLaunchKernelGGL(k_reset_state_bytes, dim3((unsigned)cdiv(n_pixels, 256)), dim3(256), 0, (hipStream_t)stream, active, selected,
                           (long long *)active_mask, (long long)n_pixels);
    return check_launch("halo_reset_round_state");
}

extern "C" int halo_undo_picks(const double *picks, const int32_t *n_picked, int64_t B, int64_t H, int64_t W, int64_t n_regions,
                               int64_t active_radius, int64_t mask_radius, uint8_t *active, uint8_t *selected,
                               int64_t *active_mask, void *stream)
{
    if (B == 0 || n_regions == 0) return HALO_OK;
    if (!picks || !n_picked || !active || !selected || !active_mask || B < 0 || B > 65535 || H <= 0 || W <= 0 || n_regions < 0 ||
        active_radius < 0 || mask_radius < 0)
        return fail(HALO_E_ARG, "halo_undo_picks: bad argument");
    hipLaunchKernelGGL(k_undo_picks, dim3((unsigned)(cdiv(n_regions, 4) < UNDO_WGS ? cdiv(n_regions, 4) : UNDO_WGS), (unsigned)B), dim3(256), 0, (hipStream_t)stream, picks,
                       (const int *)n_picked, (int)n_regions, (int)H, (int)W, (int)active_radius, (int)mask_radius, active, selected,
                       (long long *)active_mask);
    return check_launch("halo_undo_picks");
}

// every lane keeps eight 16-byte loads in flight; a workgroup walks the buffer in steps of the whole grid
__global__ void __launch_bounds__(256) k_read_probe(const u4_t *__restrict__ x, size_t n16, unsigned *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u4_t v = __builtin_nontemporal_load(x + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u && sink) atomicXor(sink, acc);      // keeps the loads alive; practically never taken
}

// the scoring pass's access pattern with its arithmetic and nothing else: `planes` planes of plane_bytes bytes per group read as
// float64, a 128-thread workgroup owns 2 KiB of every plane (16 bytes per lane, eight planes in flight, one fma per element),
// XCD-contiguous chunk map as k_feat_reduce, 16 bytes per lane written to `out`.  (A variant that only XORs what it loads and
// writes nothing does NOT see the slow stretches the real kernel sees.)
typedef double pd2_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_walk_probe(const pd2_t *__restrict__ x, size_t plane16, int planes, unsigned xcd_g, pd2_t *__restrict__ out)
{
    unsigned bx = blockIdx.x;
    const unsigned xj = blockIdx.x >> 3;
    if (xcd_g != 0 && blockIdx.x < (gridDim.x / (8 * xcd_g)) * (8 * xcd_g)) bx = (xj / xcd_g) * 8 * xcd_g + (blockIdx.x & 7) * xcd_g + xj % xcd_g;
    const pd2_t *p = x + (size_t)blockIdx.y * planes * plane16 + (size_t)bx * 128 + threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    int c = 0;
    for (; c + 8 <= planes; c += 8) {
        pd2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u, p += plane16) v[u] = __builtin_nontemporal_load(p);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    for (; c < planes; ++c, p += plane16) { const pd2_t v = __builtin_nontemporal_load(p); a0 = __builtin_fma(v.x, v.x, a0); a1 = __builtin_fma(v.y, v.y, a1); }
    pd2_t r; r.x = a0; r.y = a1;
    out[(size_t)blockIdx.y * plane16 + (size_t)bx * 128 + threadIdx.x] = r;
}

extern "C" int halo_hbm_walk_probe(const void *buf, size_t bytes, size_t plane_bytes, int planes, void *out, void *stream)
{
    if (!buf || !out || ((uintptr_t)buf & 15) || ((uintptr_t)out & 15) || planes <= 0 || plane_bytes == 0 || (plane_bytes & 2047))
        return fail(HALO_E_ARG, "halo_hbm_walk_probe: 16-byte aligned buffers, planes > 0 and plane_bytes a multiple of 2048 required");
    const size_t group = plane_bytes * (size_t)planes;
    if (bytes < group || bytes % group) return fail(HALO_E_ARG, "halo_hbm_walk_probe: bytes must be a whole number of groups of planes * plane_bytes");
    const size_t groups = bytes / group, chunks = plane_bytes / 2048;
    if (groups > 65535 || chunks > 0x7fffffffull) return fail(HALO_E_ARG, "halo_hbm_walk_probe: too many groups / chunks");
    unsigned xcd_g = 256;
    while (xcd_g > 1 && 8 * xcd_g > chunks) xcd_g >>= 1;
    if (8 * xcd_g > chunks) xcd_g = 0;
    hipLaunchKernelGGL(k_walk_probe, dim3((unsigned)chunks, (unsigned)groups), dim3(128), 0, (hipStream_t)stream, (const pd2_t *)buf,
                       plane_bytes / 16, planes, xcd_g, (pd2_t *)out);
    return check_launch("halo_hbm_walk_probe");
}

extern "C" int halo_hbm_read_probe(const void *buf, size_t bytes, void *sink, int blocks, void *stream)
{
    if (!buf || (bytes & 15) || ((uintptr_t)buf & 15)) return fail(HALO_E_ARG, "halo_hbm_read_probe: 16-byte aligned buffer and size required");
    if (bytes == 0) return HALO_OK;
    if (blocks <= 0) blocks = 256 * 16;
    hipLaunchKernelGGL(k_read_probe, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u4_t *)buf, bytes / 16, (unsigned *)sink);
    return check_launch("halo_hbm_read_probe");
}

// ---- physically contiguous pool memory.  The scoring pass walks C planes 16 MiB apart; how fast that goes depends on the
// physical pages behind the tensor: the same kernel lands on a 6.3 or a 6.6 TB/s plateau per hipMalloc'ed allocation (NOTES.md),
// and always on the upper one when the range is physically contiguous (tools/alloc_microbench2.hip: 6.5-6.7 vs 6.24-6.59 TB/s).
// The two functions have the signatures torch.cuda.memory.CUDAPluggableAllocator binds; when no contiguous range of the
// size is free the allocation falls back to hipMalloc and is counted (halo_pool_alloc_stats).
static std::atomic<unsigned long long> g_pool_stats[4];     // contiguous bytes, fallback bytes, live allocations, failures

extern "C" void *halo_pool_alloc(size_t size, int device, void *stream)
{
    (void)stream;
    if (size == 0) return nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && device != prev) (void)hipSetDevice(device);
    void *p = nullptr;
    static const bool plain = [] { const char *e = getenv("HALO_POOL_PLAIN"); return e && atoi(e) != 0; }();
    hipError_t e = plain ? hipErrorOutOfMemory : hipExtMallocWithFlags(&p, size, hipDeviceMallocContiguous);
    if (e == hipSuccess && p) {
        g_pool_stats[0] += size;
    } else {
        (void)hipGetLastError();
        p = nullptr;
        e = hipMalloc(&p, size);
        if (e == hipSuccess && p) g_pool_stats[1] += size;
        else { (void)hipGetLastError(); p = nullptr; g_pool_stats[3] += 1; }
    }
    if (p) g_pool_stats[2] += 1;
    if (device >= 0 && device != prev && prev >= 0) (void)hipSetDevice(prev);
    return p;
}

extern "C" void halo_pool_free(void *ptr, size_t size, int device, void *stream)
{
    (void)size; (void)stream;
    if (!ptr) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && device != prev) (void)hipSetDevice(device);
    (void)hipFree(ptr);
    g_pool_stats[2] -= 1;
    if (device >= 0 && device != prev && prev >= 0) (void)hipSetDevice(prev);
}

extern "C" int halo_pool_alloc_stats(uint64_t out[4])
{
    if (!out) return fail(HALO_E_ARG, "halo_pool_alloc_stats: out required");
    for (int i = 0; i < 4; ++i) out[i] = g_pool_stats[i].load();
    return HALO_OK;
}

extern "C" int halo_device_identity(int device, char *buf, size_t len)
{
    if (!buf || len < 64) return fail(HALO_E_ARG, "halo_device_identity: buffer of at least 64 bytes required");
    char pci[32] = {0};
    hipError_t e = hipDeviceGetPCIBusId(pci, (int)sizeof(pci), device);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(HALO_E_LAUNCH, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e)); }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(HALO_E_LAUNCH, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e)); }
    int off = snprintf(buf, len, "pci=%s uuid=", pci);
    for (int i = 0; i < 16 && off + 3 < (int)len; ++i) off += snprintf(buf + off, len - off, "%02x", (unsigned char)prop.uuid.bytes[i]);
    return HALO_OK;
}
