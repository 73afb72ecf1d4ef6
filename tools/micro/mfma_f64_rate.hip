// Issue-rate probe for the f64 matrix instructions of gfx950: v_mfma_f64_16x16x4_f64 (1024 MACs) against v_mfma_f64_4x4x4_4b_f64
// (4 blocks x 64 MACs) and v_fma_f64 (64 MACs), 8 independent accumulator chains per wave, one / two waves per SIMD, no memory.
// Prints cycles per instruction and SIMD at the clock the run held (s_memrealtime is 100 MHz: wall clock; the shader clock comes
// from s_memtime) and MACs per cycle and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a, double b, unsigned long long *clk)
{
    v4d acc[8];
    double s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (v4d){0, 0, 0, 0}; s[i] = threadIdx.x * 1e-3 + i; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if constexpr (OP == 1) s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[i], 0, 0, 0);
            if constexpr (OP == 2) s[i] = __builtin_fma(s[i], b, a);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int OP>
static int run(const char *name, int macs, int waves_per_simd)
{
    const int iters = 20000, blocks = 256 * waves_per_simd;               // 256 CUs x 4 SIMDs: one 256-thread block per CU and wave slot
    double *out; unsigned long long *clk, hclk = 0;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHK(hipMalloc(&clk, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0000001, 0.9999999, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999, clk);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
    const double n_inst = (double)iters * 8 * waves_per_simd;             // per SIMD
    // s_memtime-style counter of __builtin_readcyclecounter: constant 100 MHz on gfx9 -> use wall clock and report both
    const double ns_per_inst = ms * 1e6 / n_inst;
    printf("%-28s %d wave(s)/SIMD: %7.2f ns per instruction and SIMD  = %6.1f cycles at 2.4 GHz (%.1f at 1.9)   %6.2f MACs per ns and SIMD   [%.3f ms]\n", name,
           waves_per_simd, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 1.9, macs / ns_per_inst, ms);
    CHK(hipFree(out)); CHK(hipFree(clk));
    return 0;
}
int main()
{
    for (int w = 1; w <= 2; ++w) {
        if (run<0>("v_mfma_f64_16x16x4_f64", 1024, w)) return 1;
        if (run<1>("v_mfma_f64_4x4x4_4b_f64", 256, w)) return 1;
        if (run<2>("v_fma_f64", 64, w)) return 1;
    }
    return 0;
}
