// Do the FP64 pipe and the FP32 lanes of a gfx950 SIMD overlap ACROSS waves?  Even workgroups run a chain of v_fma_f64, odd workgroups a
// chain of v_fma_f32 (8 resident workgroups of 4 waves per CU: every SIMD holds 4 waves of each kind).  Three launches: the float64 half
// alone, the float32 half alone, both together.  together ~ max(a, b): the two instruction kinds issue side by side (a float32-heavy
// pass could hide behind a float64-bound one in ONE kernel); together ~ a + b: they share the issue slot.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KIND_EVEN, int KIND_ODD>      // 0: nothing, 1: v_fma_f64, 2: v_fma_f32, 3: v_cvt / v_cndmask style 4-cycle op (v_rndne_f32), 4: ds_read (LDS)
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    __shared__ float lds[1024];
    const int kind = (blockIdx.x & 1) ? KIND_ODD : KIND_EVEN;
    float x[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i + 1.0f; d[i] = x[i]; }
    const double da = a, db = b;
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b; lds[threadIdx.x + 512] = a; lds[threadIdx.x + 768] = b;
    __syncthreads();
    if (kind == 0) return;
    for (int it = 0; it < iters; ++it) {
#define S_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(da));
#define S_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
#define S_RND(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
        if (kind == 1) { REP8(S_F64) }
        else if (kind == 2) { REP8(S_F32) REP8(S_F32) }
        else if (kind == 3) { REP8(S_RND) }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] += lds[(threadIdx.x * 4 + i * 64 + it) & 1023];
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef void (*kern_t)(float *, int, float, float);
static float run(kern_t fn, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 8192;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main()
{
    float *out; CHK(hipMalloc(&out, 256 * 8192 * 4));
    const float f64 = run(k<1, 0>, out), f32 = run(k<0, 2>, out), both = run(k<1, 2>, out);
    printf("v_fma_f64 half alone %.3f ms | v_fma_f32 half alone (2x the instructions) %.3f ms | together %.3f ms  (sum %.3f, max %.3f)\n", f64, f32, both, f64 + f32, f64 > f32 ? f64 : f32);
    const float rnd = run(k<0, 3>, out), b2 = run(k<1, 3>, out);
    printf("v_fma_f64 half alone %.3f ms | v_rndne_f32 half alone %.3f ms | together %.3f ms  (sum %.3f, max %.3f)\n", f64, rnd, b2, f64 + rnd, f64 > rnd ? f64 : rnd);
    const float b3 = run(k<2, 3>, out), f32e = run(k<2, 0>, out);
    printf("v_fma_f32 half alone %.3f ms | v_rndne_f32 half alone %.3f ms | together %.3f ms  (sum %.3f, max %.3f)\n", f32e, rnd, b3, f32e + rnd, f32e > rnd ? f32e : rnd);
    const float l = run(k<0, 4>, out), b4 = run(k<1, 4>, out);
    printf("v_fma_f64 half alone %.3f ms | LDS reads half alone %.3f ms | together %.3f ms  (sum %.3f, max %.3f)\n", f64, l, b4, f64 + l, f64 > l ? f64 : l);
    const float full64 = run(k<1, 1>, out), full32 = run(k<2, 2>, out);
    printf("all workgroups v_fma_f64 %.3f ms | all workgroups v_fma_f32 %.3f ms\n", full64, full32);
    return 0;
}
