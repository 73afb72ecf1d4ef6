// Tuning aid (not part of the library): the k_feat_reduce access pattern (C planes of H*W f64, a lane owns 16 B, a 128-thread
// block 2 KiB per plane) on non-constant data, 16 images per launch as in bench.py, average of 6 launches -- by cache-policy
// bits of the load (inline asm), by block -> chunk mapping, against a flat read of the same bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/feat_microbench2.hip -o tools/feat_mb2 && tools/feat_mb2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ void k_fill(double *x, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        x[i] = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.2;
    }
}

#define LOADS8(MOD)                                                                                        \
    asm volatile("global_load_dwordx4 %0, %8, off " MOD "\n\tglobal_load_dwordx4 %1, %9, off " MOD "\n\t"  \
                 "global_load_dwordx4 %2, %10, off " MOD "\n\tglobal_load_dwordx4 %3, %11, off " MOD "\n\t" \
                 "global_load_dwordx4 %4, %12, off " MOD "\n\tglobal_load_dwordx4 %5, %13, off " MOD "\n\t" \
                 "global_load_dwordx4 %6, %14, off " MOD "\n\tglobal_load_dwordx4 %7, %15, off " MOD        \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
                 : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]) : "memory")

// MODE: 0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt;  MAP: 0 linear, 1 XCD-contiguous
template <int MODE, int MAP, int TPB_, int WPE>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, WPE)))
k_planes_asm(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out)
{
    const int b = blockIdx.y;
    unsigned bx = blockIdx.x;
    if (MAP == 1) { const unsigned per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const long long i0 = ((long long)bx * TPB_ + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + 8 <= C; c += 8) {
        d2_t v[8];
        const double *q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = p + (size_t)(c + u) * hw;
        if (MODE == 0) LOADS8("");
        if (MODE == 1) LOADS8("nt");
        if (MODE == 2) LOADS8("sc0");
        if (MODE == 3) LOADS8("sc1");
        if (MODE == 4) LOADS8("sc0 sc1");
        if (MODE == 5) LOADS8("sc0 nt");
        if (MODE == 6) LOADS8("sc1 nt");
        if (MODE == 7) LOADS8("sc0 sc1 nt");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}

// the compiler's own schedule (builtin nt loads, unroll U), block -> chunk map as above
template <int U, int MAP, int TPB_, int WPE>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, WPE)))
k_planes(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out)
{
    const int b = blockIdx.y;
    unsigned bx = blockIdx.x;
    if (MAP == 1) { const unsigned per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    if (MAP == 2) { bx = gridDim.x - 1 - bx; }
    const long long i0 = ((long long)bx * TPB_ + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + U <= C; c += U) {
        d2_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
        for (int u = 0; u < U; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}

// XCD-contiguous in granules of G chunks: workgroup id -> XCD (id & 7, the hardware's round-robin), the XCD's j-th workgroup
// (j = id >> 3) takes chunk  (j / G) * 8G + xcd * G + j % G
template <int U, int TPB_, int WPE>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, WPE)))
k_planes_g(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out, unsigned G)
{
    const int b = blockIdx.y;
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const unsigned bx = (j / G) * 8 * G + xcd * G + j % G;
    const long long i0 = ((long long)bx * TPB_ + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + U <= C; c += U) {
        d2_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
        for (int u = 0; u < U; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}

// one image per XCD at a time: flat id -> xcd = id & 7, j = id >> 3; image = 8 * (j / chunks) + xcd, chunk = j % chunks
template <int U, int TPB_>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_planes_imgxcd(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out, unsigned chunks)
{
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = 8 * (j / chunks) + xcd;
    const long long i0 = ((long long)(j % chunks) * TPB_ + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + U <= C; c += U) {
        d2_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
        for (int u = 0; u < U; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}

// flat read, XCD-contiguous: each XCD streams its own eighth of the buffer
__global__ void __launch_bounds__(256) k_flat_xcd(const d2_t *__restrict__ x, size_t n, double *__restrict__ out)
{
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const size_t n8 = n / 8;
    const d2_t *base = x + xcd * n8;
    size_t i = (size_t)j * 256 + threadIdx.x;
    const size_t stride = (size_t)per * 256;
    double a = 0;
    for (; i + 7 * stride < n8; i += 8 * stride) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u].x + v[u].y;
    }
    if (a == 123.456) out[0] = a;
}

// a block owns S consecutive 2 KiB chunks; per group of 8 planes it walks its S chunks one after the other (8 loads in flight),
// so the 8 pages of the group are used S times in a row and every plane is read in runs of S * 2 KiB; S accumulator pairs per lane
template <int S, int WPE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, WPE)))
k_planes_sub(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out, unsigned G)
{
    const int b = blockIdx.y;
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const unsigned bx = G ? (j / G) * 8 * G + xcd * G + j % G : blockIdx.x;
    const long long i0 = ((long long)bx * S * 128 + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0[S], a1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) a0[s] = a1[s] = 0;
    for (int c = 0; c + 8 <= C; c += 8) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            d2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw + s * 256));
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0[s] = __builtin_fma(v[u].x, v[u].x, a0[s]); a1[s] = __builtin_fma(v[u].y, v[u].y, a1[s]); }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) { d2_t r; r.x = a0[s]; r.y = a1[s]; *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0 + s * 256) = r; }
}

// images along x: consecutive blocks work on the same chunk of different images (grid (B, chunks))
template <int U, int TPB_>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_planes_bfast(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out)
{
    const int b = blockIdx.x;
    const long long i0 = ((long long)blockIdx.y * TPB_ + threadIdx.x) * 2;
    const double *p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + U <= C; c += U) {
        d2_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
        for (int u = 0; u < U; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
}

// persistent: `blocks` workgroups walk the chunks of the launch in order (chunk = blockIdx + k * gridDim)
template <int U, int TPB_>
__global__ void __launch_bounds__(TPB_) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_planes_persist(const double *__restrict__ feat, long long bstride, int C, long long hw, double *__restrict__ out, long long nchunks_img, int B)
{
    const long long total = nchunks_img * B;
    for (long long ch = blockIdx.x; ch < total; ch += gridDim.x) {
        const int b = (int)(ch / nchunks_img);
        const long long i0 = ((ch % nchunks_img) * TPB_ + threadIdx.x) * 2;
        const double *p = feat + (size_t)b * bstride + i0;
        double a0 = 0, a1 = 0;
        for (int c = 0; c + U <= C; c += U) {
            d2_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p + (size_t)(c + u) * hw));
#pragma unroll
            for (int u = 0; u < U; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
        }
        d2_t r; r.x = a0; r.y = a1;
        *reinterpret_cast<d2_t *>(out + (size_t)b * hw + i0) = r;
    }
}

template <int NT>
__global__ void __launch_bounds__(256) k_flat(const d2_t *__restrict__ x, size_t n, double *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double a = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u].x + v[u].y;
    }
    if (a == 123.456) out[0] = a;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double g_gb;
template <typename F> void run(const char *name, F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    double sum = 0; float best = 1e30f;
    for (int r = 0; r < 6; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; best = ms < best ? ms : best; }
    printf("%-64s avg %7.3f ms  %6.0f GB/s   (min %7.3f ms %6.0f GB/s)\n", name, sum / 6, g_gb / (sum / 6) * 1e3, best, g_gb / best * 1e3);
    fflush(stdout);
}

int main()
{
    const int B = 16, C = 256; const long long hw = 1024ll * 2048;
    const size_t n = (size_t)B * C * hw;
    double *feat, *out;
    CK(hipMalloc(&feat, n * 8)); CK(hipMalloc(&out, (size_t)B * hw * 8));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, feat, n); CK(hipDeviceSynchronize());
    g_gb = (double)n * 8 / 1e9;
    printf("bytes per launch %.2f GB (non-constant data)\n", g_gb);
    const long long bs = (long long)C * hw;
#define G(T) dim3((unsigned)(hw / 2 / T), B)
#define ASM(MODE, MAP, T, W, label) run(label, [&] { hipLaunchKernelGGL((k_planes_asm<MODE, MAP, T, W>), G(T), dim3(T), 0, 0, feat, bs, C, hw, out); });
#define PL(U, MAP, T, W, label) run(label, [&] { hipLaunchKernelGGL((k_planes<U, MAP, T, W>), G(T), dim3(T), 0, 0, feat, bs, C, hw, out); });
    PL(8, 0, 128, 4, "planes builtin nt, unroll 8, 128 thr, <=4 waves/SIMD (library)")
    PL(8, 1, 128, 4, "planes builtin nt, XCD-contiguous chunks (G = chunks/8)")
    for (unsigned Gk : {32u, 128u, 256u, 512u, 1024u}) {
        char nm[96]; snprintf(nm, sizeof nm, "planes XCD granule %u chunks (%u KiB), 128 thr", Gk, Gk * 2);
        run(nm, [&] { hipLaunchKernelGGL((k_planes_g<8, 128, 4>), G(128), dim3(128), 0, 0, feat, bs, C, hw, out, Gk); });
    }
    for (unsigned Gk : {512u, 1024u, 2048u}) {
        char nm[96]; snprintf(nm, sizeof nm, "planes XCD granule %u chunks (%u KiB), 64 thr", Gk, Gk);
        run(nm, [&] { hipLaunchKernelGGL((k_planes_g<8, 64, 4>), G(64), dim3(64), 0, 0, feat, bs, C, hw, out, Gk); });
    }
    for (unsigned Gk : {512u, 1024u}) {
        char nm[96]; snprintf(nm, sizeof nm, "planes XCD granule %u chunks, 128 thr, <=8 waves", Gk);
        run(nm, [&] { hipLaunchKernelGGL((k_planes_g<8, 128, 8>), G(128), dim3(128), 0, 0, feat, bs, C, hw, out, Gk); });
        snprintf(nm, sizeof nm, "planes XCD granule %u chunks, 128 thr, unroll 16", Gk);
        run(nm, [&] { hipLaunchKernelGGL((k_planes_g<16, 128, 4>), G(128), dim3(128), 0, 0, feat, bs, C, hw, out, Gk); });
    }
#define SUB(S_, W_, Gk) { char nm[96]; snprintf(nm, sizeof nm, "planes, block owns %d chunks (%d KiB runs), <=%d waves, granule %u", S_, 2 * S_, W_, Gk); \
    run(nm, [&] { hipLaunchKernelGGL((k_planes_sub<S_, W_>), dim3((unsigned)(hw / 2 / 128 / S_), B), dim3(128), 0, 0, feat, bs, C, hw, out, Gk); }); }
    SUB(1, 4, 256u) SUB(2, 4, 128u) SUB(4, 4, 64u) SUB(8, 4, 32u) SUB(4, 4, 0u) SUB(4, 8, 64u) SUB(8, 8, 32u) SUB(16, 4, 16u) SUB(2, 4, 256u) SUB(4, 4, 256u)
    run("planes, one image per XCD at a time", [&] { hipLaunchKernelGGL((k_planes_imgxcd<8, 128>), dim3((unsigned)(hw / 2 / 128) * B), dim3(128), 0, 0, feat, bs, C, hw, out, (unsigned)(hw / 2 / 128)); });
    for (int blocks : {4096, 16384}) {
        char nm[96]; snprintf(nm, sizeof nm, "flat nt read, %d workgroups of 256", blocks);
        run(nm, [&] { hipLaunchKernelGGL((k_flat<1>), dim3(blocks), dim3(256), 0, 0, (const d2_t *)feat, n / 2, out); });
        snprintf(nm, sizeof nm, "flat nt read XCD-contiguous, %d workgroups of 256", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k_flat_xcd, dim3(blocks), dim3(256), 0, 0, (const d2_t *)feat, n / 2, out); });
    }
    PL(8, 0, 128, 4, "planes builtin nt, unroll 8, 128 thr, <=4 (again, drift check)")
    return 0;
}
