/* Calls the Sleef functions linked into libtorch_cpu.so (Sleef_expf8_u10: what ATen's Vectorized<float>::exp() evaluates on an
 * AVX2 host; the AVX-512 build calls Sleef_expf16_u10, the same algorithm 16 lanes wide) so that tests/test_aten_exact.py can
 * hold the oracle's restatement against the real thing.    gcc -O2 -mavx2 -shared -fPIC sleef_shim.c -ldl */
#include <dlfcn.h>
#include <immintrin.h>
typedef __m256 (*f8_t)(__m256);
static void *handle;
int shim_open(const char *path)
{
    handle = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
    if (!handle) handle = dlopen(path, RTLD_NOW);
    return handle != 0;
}
int shim_call_f8(const char *name, const float *x, float *y, long n)          /* n a multiple of 8 */
{
    f8_t f = (f8_t)dlsym(handle, name);
    if (!f) return -1;
    for (long i = 0; i + 8 <= n; i += 8) _mm256_storeu_ps(y + i, f(_mm256_loadu_ps(x + i)));
    return 0;
}
