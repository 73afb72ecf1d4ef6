// Issue-rate probe for gfx950: one instruction type per kernel, 16 independent chains per lane, 8 waves per SIMD, no memory.
// Reports wave64 instructions per second and cycles per wave instruction and SIMD at the clock the run held (s_memtime / wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b, int n)
{
    float x[16];
    double d[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i + 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 1e-3 + i + 1.0; p[i] = (f2){x[i], x[i + 8]}; }
    const double da = a, db = b;
    for (int it = 0; it < iters; ++it) {
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
#define S_PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((f2){b, b}), "v"((f2){a, a}));
#define S_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(da));
#define S_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
#define S_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
#define S_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
#define S_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
#define S_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(n));
#define S_FREXP(i) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(x[i]));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define S_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
#define S_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
#define S_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define S_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define S_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(n));
#define S_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(n));
#define S_CMPCND(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
#define S_CMPCND2(i) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cmp_gt_f32 s[20:21], %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, s[20:21]" : "+v"(x[i]), "+v"(x[i + 8]) : "v"(a) : "vcc", "s20", "s21");
#define S_FMAMIX(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_rndne_f32 %1, %1" : "+v"(x[i]), "+v"(x[i + 8]) : "v"(b), "v"(a));
        if constexpr (OP == 0) { REP16(S_FMA) }
        if constexpr (OP == 1) { REP8(S_PK) }
        if constexpr (OP == 2) { REP8(S_F64) }
        if constexpr (OP == 3) { REP8(S_MUL64) }
        if constexpr (OP == 4) { REP8(S_ADD64) }
        if constexpr (OP == 5) { REP16(S_RNDNE) }
        if constexpr (OP == 6) { REP16(S_CVT) }
        if constexpr (OP == 7) { REP16(S_LDEXP) }
        if constexpr (OP == 8) { REP16(S_FREXP) }
        if constexpr (OP == 9) { REP16(S_RCP) }
        if constexpr (OP == 10) { REP16(S_CND) }
        if constexpr (OP == 11) { REP16(S_CMP) }
        if constexpr (OP == 12) { REP16(S_MAX) }
        if constexpr (OP == 13) { REP16(S_MUL) }
        if constexpr (OP == 14) { REP16(S_ADD) }
        if constexpr (OP == 15) { REP16(S_EXP) }
        if constexpr (OP == 16) { REP16(S_ADDU) }
        if constexpr (OP == 17) { REP16(S_LSHLADD) }
        if constexpr (OP == 18) { REP16(S_CMPCND) }
        if constexpr (OP == 19) { REP8(S_CMPCND2) }
        if constexpr (OP == 20) { REP8(S_FMAMIX) }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef void (*kern_t)(float *, int, float, float, int);
struct Op { const char *name; kern_t fn; int per_iter; };
int main(int argc, char **argv)
{
    const double target_ms = argc > 1 ? atof(argv[1]) : 150.0;
    float *out; CHK(hipMalloc(&out, 256 * 8192 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2048, blocks = 8192;          // 8192 blocks x 4 waves = 32 waves per SIMD in total, 8 resident
    Op ops[] = {{"v_fma_f32", k<0>, 16}, {"v_pk_fma_f32", k<1>, 8}, {"v_fma_f64", k<2>, 8}, {"v_mul_f64", k<3>, 8}, {"v_add_f64", k<4>, 8},
                {"v_rndne_f32", k<5>, 16}, {"v_cvt_i32_f32", k<6>, 16}, {"v_ldexp_f32", k<7>, 16}, {"v_frexp_mant_f32", k<8>, 16},
                {"v_rcp_f32", k<9>, 16}, {"v_cndmask_b32 (vcc)", k<10>, 16}, {"v_cmp_gt_f32 -> vcc", k<11>, 16}, {"v_max_f32", k<12>, 16},
                {"v_mul_f32", k<13>, 16}, {"v_add_f32", k<14>, 16}, {"v_exp_f32", k<15>, 16}, {"v_add_u32", k<16>, 16},
                {"v_lshl_add_u32", k<17>, 16}, {"v_cmp + s_nop 1 + v_cndmask (2 VALU)", k<18>, 32}, {"2 x v_cmp then 2 x v_cndmask (4 VALU)", k<19>, 32},
                {"v_fma_f32 + v_rndne_f32 interleaved (2 VALU)", k<20>, 16}};
    for (auto &op : ops) {
        // warm-up launch, then as many launches as fill target_ms
        hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f, 0.999f, 0);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f, 0.999f, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms1; CHK(hipEventElapsedTime(&ms1, e0, e1));
        int n = (int)(target_ms / ms1) + 1;
        CHK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f, 0.999f, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)blocks * 4 * iters * op.per_iter * n;          // wave64 instructions
        const double per_simd_s = winstr / 1024.0 / (ms * 1e-3);                     // per SIMD and second
        printf("%-46s first %.3f ms | %4d launches %.1f ms: %.2f T lane-instr/s, %.2f ns per wave instruction and SIMD (= %.2f cycles at 2.4 GHz)\n",
               op.name, ms1, n, ms, winstr * 64 / ms / 1e9, 1e9 / per_simd_s, 2.4e9 / per_simd_s);
    }
    return 0;
}
