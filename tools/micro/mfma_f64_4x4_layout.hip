// Layout probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: for every (la, lb) a wave runs the instruction with A = one-hot at lane la,
// B = one-hot at lane lb, C = 0 and reports which lane of D holds the 1 (or none).  From the table the (block, row, k) / (block, k, col)
// / (block, row, col) roles of the lanes follow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out)
{
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, l = threadIdx.x;
    const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[blockIdx.x] = l + 1;
}
int main()
{
    int *dev; hipMalloc(&dev, 4096 * 4); hipMemset(dev, 0, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, dev);
    std::vector<int> h(4096); hipMemcpy(h.data(), dev, 4096 * 4, hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it pairs with and the D lanes hit
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes -> D lane:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) printf(" %d->%d", lb, h[la * 64 + lb] - 1);
        printf("\n");
    }
    return 0;
}
