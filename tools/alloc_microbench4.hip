// Tuning aid (not part of the library): does the ALIGNMENT of the virtual range decide the plane walk's plateau?  (TLB theory: the
// driver's PTE fragment size is bounded by the common alignment of virtual and physical address.)  A 128 GiB pool through
// hipMalloc, hipExtMallocWithFlags(contiguous) and through the virtual-memory API with a reservation aligned to 2 MiB / 1 GiB /
// 64 GiB; plane walk (fma + store) and flat read per 16 GiB window.
//   hipcc -O3 --offload-arch=gfx950 tools/alloc_microbench4.hip -o tools/alloc_mb4 && tools/alloc_mb4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); fflush(stdout); return 1; } } while (0)

__global__ void k_fill(double *x, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        x[i] = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.2;
    }
}
__global__ void __launch_bounds__(256) k_flat(const d2_t *__restrict__ x, size_t n, double *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double a = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u].x + v[u].y;
    }
    if (a == 123.456) out[0] = a;
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_planes(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out, unsigned G)
{
    const int b = blockIdx.y;
    unsigned bx = blockIdx.x;
    if (G) { const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3; bx = (j / G) * 8 * G + xcd * G + j % G; }
    const long long i0 = ((long long)bx * 128 + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + 8 <= C; c += 8) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}
template <typename F> static float avg_ms(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int r = 0; r < 4; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / 4;
}
int main(int argc, char **argv)
{
    const int C = 256; const long long hw = 1024ll * 2048;
    const size_t img = (size_t)C * hw;
    const int gib = argc > 1 ? atoi(argv[1]) : 128;
    const size_t bytes = (size_t)gib << 30;
    double *out; CK(hipMalloc(&out, (size_t)4 * hw * 8));
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("recommended granularity %zu bytes\n", gran);
    const char *names[] = {"hipMalloc", "hipExtMalloc contiguous", "VMM reservation aligned 2 MiB", "VMM reservation aligned 1 GiB", "VMM reservation aligned 64 GiB"};
    for (int rep = 0; rep < 2; ++rep)
    for (int how = 0; how < 5; ++how) {
        double *feat = nullptr; hipMemGenericAllocationHandle_t h = 0; bool vmm = how >= 2;
        if (how == 0) CK(hipMalloc(&feat, bytes));
        else if (how == 1) { if (hipExtMallocWithFlags((void **)&feat, bytes, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); printf("contiguous: no\n"); continue; } }
        else {
            const size_t align = how == 2 ? (size_t)2 << 20 : how == 3 ? (size_t)1 << 30 : (size_t)64 << 30;
            hipError_t e = hipMemAddressReserve((void **)&feat, bytes, align, nullptr, 0);
            if (e != hipSuccess) { printf("%s: reserve failed: %s\n", names[how], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            e = hipMemCreate(&h, bytes, &prop, 0);
            if (e != hipSuccess) { printf("%s: create failed: %s\n", names[how], hipGetErrorString(e)); (void)hipGetLastError(); (void)hipMemAddressFree(feat, bytes); continue; }
            CK(hipMemMap(feat, bytes, 0, h, 0));
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(feat, bytes, &acc, 1));
        }
        hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, feat, bytes / 8); CK(hipDeviceSynchronize());
        printf("rep %d %-34s %p:", rep, names[how], (void *)feat);
        const int nimg = gib / 4;
        for (int i0 = 0; i0 + 4 <= nimg; i0 += 4) {
            const double *w = feat + (size_t)i0 * img;
            const double gb = 4.0 * img * 8 / 1e9;
            const float p1 = avg_ms([&] { hipLaunchKernelGGL(k_planes, dim3((unsigned)(hw / 256), 4), dim3(128), 0, 0, w, (long long)img, C, hw, out, 256u); });
            printf(" %5.0f", gb / p1 * 1e3);
        }
        const float f = avg_ms([&] { hipLaunchKernelGGL(k_flat, dim3(16384), dim3(256), 0, 0, (const d2_t *)feat, (size_t)16 * img / 2, out); });
        printf("   | flat (first 64 GiB) %5.0f GB/s\n", 16.0 * img * 8 / 1e9 / f * 1e3);
        fflush(stdout);
        if (vmm) { CK(hipMemUnmap(feat, bytes)); CK(hipMemRelease(h)); CK(hipMemAddressFree(feat, bytes)); }
        else CK(hipFree(feat));
    }
    return 0;
}
