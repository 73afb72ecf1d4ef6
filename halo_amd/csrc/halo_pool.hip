// Pool-side helpers of the acquisition round (core/datasets/cityscapes.py:245-251, core/active/build.py:52-62):
//   halo_pack_pick_tables   per-image pick tables -> the fixed-size int32 wire block of the round's ONE all-gather
//   halo_reset_round_state  the loader's round-1 state (active = selected = False, active_mask = 255) in one pass of
//                           16-byte stores (three torch fills ran at 0.65-1.9 TB/s)
//   halo_undo_picks         the same state restored from a pick table: only the windows the selection wrote are rewritten
//   halo_device_identity    PCI bus id + UUID of a device (ranks of one node must hold distinct devices)
// (The HBM read / walk probes and the contiguous-range allocator of round 3 are measurement aids: tools/halo_probe.hip.)
#include "halo_common.hpp"

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
