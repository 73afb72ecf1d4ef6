// Issue-rate probe: scalar v_fma_f32 vs packed v_pk_fma_f32 vs v_fma_f64 (independent chains, no memory), lane-FMAs per second.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0)
{
    const float a = a0, b = 0.999f;
    if constexpr (MODE == 0) {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], b, a);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 1) {
        f2 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], (f2)(b), (f2)(a));
        }
        f2 s = (f2)(0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], (double)b, (double)a);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;
    }
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 8192;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double lane_fma = (double)blocks * 256 * iters * (mode == 2 ? 8 : 16);
            const double instr = (double)blocks * 256 * iters * (mode == 0 ? 16 : 8);
            printf("%s: %.3f ms  %.2f T lane-FMA/s  %.2f T lane-instr/s\n", mode == 0 ? "v_fma_f32   " : mode == 1 ? "v_pk_fma_f32" : "v_fma_f64   ", ms, lane_fma / ms / 1e9, instr / ms / 1e9);
        }
    }
    return 0;
}
