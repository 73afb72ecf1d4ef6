// tools/libhalo_probe.so -- measurement aids, NOT part of the product ABI (include/halo_hip.h): a flat non-temporal streaming read
// (bench.py's `roofline.flat_read`: this box's own ceiling for the bytes k_feat_reduce streams), the scorer's plane walk with its
// arithmetic and nothing else, and an allocator for physically contiguous HBM ranges (round 3's placement study, NOTES.md).
// Built by tools/halo_probe.py (hipcc --offload-arch=gfx950 -shared); nothing in halo_amd/ loads it.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace {
enum { PROBE_OK = 0, PROBE_E_ARG = -1, PROBE_E_LAUNCH = -3 };
thread_local char g_err[256];
int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e)); return PROBE_E_LAUNCH; }
    return PROBE_OK;
}
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

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

}  // namespace

extern "C" const char *halo_probe_last_error(void) { return g_err; }

extern "C" int halo_hbm_walk_probe(const void *buf, size_t bytes, size_t plane_bytes, int planes, void *out, void *stream)
{
    if (!buf || !out || ((uintptr_t)buf & 15) || ((uintptr_t)out & 15) || planes <= 0 || plane_bytes == 0 || (plane_bytes & 2047))
        return fail(PROBE_E_ARG, "halo_hbm_walk_probe: 16-byte aligned buffers, planes > 0 and plane_bytes a multiple of 2048 required");
    const size_t group = plane_bytes * (size_t)planes;
    if (bytes < group || bytes % group) return fail(PROBE_E_ARG, "halo_hbm_walk_probe: bytes must be a whole number of groups of planes * plane_bytes");
    const size_t groups = bytes / group, chunks = plane_bytes / 2048;
    if (groups > 65535 || chunks > 0x7fffffffull) return fail(PROBE_E_ARG, "halo_hbm_walk_probe: too many groups / chunks");
    unsigned xcd_g = 256;
    while (xcd_g > 1 && 8 * xcd_g > chunks) xcd_g >>= 1;
    if (8 * xcd_g > chunks) xcd_g = 0;
    hipLaunchKernelGGL(k_walk_probe, dim3((unsigned)chunks, (unsigned)groups), dim3(128), 0, (hipStream_t)stream, (const pd2_t *)buf,
                       plane_bytes / 16, planes, xcd_g, (pd2_t *)out);
    return check_launch("halo_hbm_walk_probe");
}

extern "C" int halo_hbm_read_probe(const void *buf, size_t bytes, void *sink, int blocks, void *stream)
{
    if (!buf || (bytes & 15) || ((uintptr_t)buf & 15)) return fail(PROBE_E_ARG, "halo_hbm_read_probe: 16-byte aligned buffer and size required");
    if (bytes == 0) return PROBE_OK;
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
    if (!out) return fail(PROBE_E_ARG, "halo_pool_alloc_stats: out required");
    for (int i = 0; i < 4; ++i) out[i] = g_pool_stats[i].load();
    return PROBE_OK;
}

