// Tuning aid (not part of the library): what does the MI355X sustain for the k_feat_reduce access
// pattern (C channel planes of H*W f64, each lane owning 16 B of consecutive pixels) under
// different unroll depths / cache policies / block sizes, and for a flat streaming read?
//   hipcc -O3 --offload-arch=gfx950 tools/feat_microbench.hip -o /tmp/feat_mb && /tmp/feat_mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int UNROLL, int NT, int TPB_>
__global__ void __launch_bounds__(TPB_) k_planes(const double* __restrict__ feat, long long bstride, int C, long long hw, double* __restrict__ out)
{
    const int b = blockIdx.y;
    const long long i0 = ((long long)blockIdx.x * TPB_ + threadIdx.x) * 2;
    if (i0 >= hw) return;
    const double* p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + UNROLL <= C; c += UNROLL) {
        d2_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const d2_t* q = reinterpret_cast<const d2_t*>(p + (size_t)(c + u) * hw);
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t*>(out + (size_t)b * hw + i0) = r;
}

// explicit plane stride (in doubles)
template <int UNROLL, int TPB_>
__global__ void __launch_bounds__(TPB_) k_planes_s(const double* __restrict__ feat, long long bstride, int C, long long hw, long long pstride, double* __restrict__ out)
{
    const int b = blockIdx.y;
    const long long i0 = ((long long)blockIdx.x * TPB_ + threadIdx.x) * 2;
    if (i0 >= hw) return;
    const double* p = feat + (size_t)b * bstride + i0;
    double a0 = 0, a1 = 0;
    for (int c = 0; c + UNROLL <= C; c += UNROLL) {
        d2_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p + (size_t)(c + u) * pstride));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); }
    }
    d2_t r; r.x = a0; r.y = a1;
    *reinterpret_cast<d2_t*>(out + (size_t)b * hw + i0) = r;
}

// 4 doubles (32 B) per lane: a wave reads 2 KiB contiguous per plane
template <int UNROLL, int TPB_>
__global__ void __launch_bounds__(TPB_) k_planes4(const double* __restrict__ feat, long long bstride, int C, long long hw, double* __restrict__ out)
{
    const int b = blockIdx.y;
    const long long i0 = ((long long)blockIdx.x * TPB_ + threadIdx.x) * 2 + (long long)(threadIdx.x >> 6) * 0;
    // lane l of wave w: first 16 B at wave_base + l*16, second at wave_base + 1024 + l*16 (both coalesced)
    const long long wave_base = ((long long)blockIdx.x * (TPB_ / 64) + (threadIdx.x >> 6)) * 256;
    const long long j0 = wave_base + (threadIdx.x & 63) * 2, j1 = j0 + 128;
    if (j1 + 1 >= hw) return;
    (void)i0;
    const double* p = feat + (size_t)b * bstride;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int c = 0; c + UNROLL <= C; c += UNROLL) {
        d2_t v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p + (size_t)(c + u) * hw + j0));
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p + (size_t)(c + u) * hw + j1));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { a0 = __builtin_fma(v[u].x, v[u].x, a0); a1 = __builtin_fma(v[u].y, v[u].y, a1); a2 = __builtin_fma(w[u].x, w[u].x, a2); a3 = __builtin_fma(w[u].y, w[u].y, a3); }
    }
    d2_t r; r.x = a0; r.y = a1; d2_t q; q.x = a2; q.y = a3;
    *reinterpret_cast<d2_t*>(out + (size_t)b * hw + j0) = r;
    *reinterpret_cast<d2_t*>(out + (size_t)b * hw + j1) = q;
}

// flat grid-stride streaming read (ceiling for pure reads)
template <int NT>
__global__ void __launch_bounds__(256) k_flat(const d2_t* __restrict__ x, size_t n, double* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double a = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        d2_t v0 = NT ? __builtin_nontemporal_load(x + i) : x[i];
        d2_t v1 = NT ? __builtin_nontemporal_load(x + i + stride) : x[i + stride];
        d2_t v2 = NT ? __builtin_nontemporal_load(x + i + 2 * stride) : x[i + 2 * stride];
        d2_t v3 = NT ? __builtin_nontemporal_load(x + i + 3 * stride) : x[i + 3 * stride];
        a += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    if (a == 123.456) out[0] = a;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    float best = t[0]; double sum = 0; for (float v : t) { best = v < best ? v : best; sum += v; }
    printf("   min %.3f ms  avg %.3f ms", best, sum / t.size());
    return best;
}

int main()
{
    const int B = 8, C = 256; const long long hw = 1024ll * 2048;
    const size_t n = (size_t)B * C * (hw + 70000);
    double *feat, *out;
    CK(hipMalloc(&feat, n * 8)); CK(hipMalloc(&out, (size_t)B * hw * 8));
    CK(hipMemset(feat, 0x3c, n * 8));
    const double gb = (double)B * C * hw * 8 / 1e9;
    printf("bytes per launch %.2f GB\n", gb);
#define RUN(U, NT, T) { printf("planes unroll %2d nt %d tpb %4d:", U, NT, T); dim3 g((unsigned)((hw / 2 + T - 1) / T), B); \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_planes<U, NT, T>), g, dim3(T), 0, 0, feat, (long long)C * hw, C, hw, out); }, 8); printf("  -> %.0f GB/s\n", gb / ms * 1e3); }
    RUN(4, 1, 256) RUN(8, 1, 256) RUN(16, 1, 256) RUN(32, 1, 256)
    RUN(8, 0, 256) RUN(16, 0, 256)
    RUN(8, 1, 128) RUN(8, 1, 512) RUN(8, 1, 1024) RUN(16, 1, 512) RUN(4, 1, 1024)
    RUN(8, 1, 64) RUN(16, 1, 64) RUN(16, 1, 128) RUN(4, 1, 128) RUN(32, 1, 128)
#define RUN4(U, T) { printf("planes4 unroll %2d tpb %4d:", U, T); dim3 g((unsigned)((hw / 4 + T - 1) / T), B); \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_planes4<U, T>), g, dim3(T), 0, 0, feat, (long long)C * hw, C, hw, out); }, 8); printf("  -> %.0f GB/s\n", gb / ms * 1e3); }
    RUN4(4, 256) RUN4(8, 256) RUN4(4, 128) RUN4(8, 128) RUN4(8, 64)
    // plane stride: 2^24 bytes (the contiguous (C,H,W) tensor) vs padded strides -- does power-of-two aliasing matter?
    {
        const long long pads[] = {0, 64, 512, 4096, 65536 + 512};      // extra doubles per plane
        for (long long pad : pads) {
            const long long stride = hw + pad;
            if ((size_t)B * C * stride > n) continue;
            printf("planes unroll  8 nt 1 tpb  128 plane stride hw+%-6lld:", pad);
            dim3 g((unsigned)((hw / 2 + 127) / 128), B);
            float ms = time_ms([&] { hipLaunchKernelGGL((k_planes_s<8, 128>), g, dim3(128), 0, 0, feat, (long long)C * stride, C, hw, stride, out); }, 8);
            printf("  -> %.0f GB/s\n", gb / ms * 1e3);
        }
    }
    for (int nt = 1; nt < 2; ++nt) for (int blocks : {2048, 4096, 8192, 16384}) {
        printf("flat nt %d blocks %5d:", nt, blocks);
        float ms = nt ? time_ms([&] { hipLaunchKernelGGL((k_flat<1>), dim3(blocks), dim3(256), 0, 0, (const d2_t*)feat, (size_t)B * C * hw / 2, out); }, 8)
                      : time_ms([&] { hipLaunchKernelGGL((k_flat<0>), dim3(blocks), dim3(256), 0, 0, (const d2_t*)feat, (size_t)B * C * hw / 2, out); }, 8);
        printf("  -> %.0f GB/s\n", gb / ms * 1e3);
    }
    return 0;
}
