// Shared host-side helpers for the C-ABI translation units of libhalo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/halo_hip.h"

namespace halo {

char *err_buf();   // thread-local message buffer (halo_api.hip)
constexpr int ERR_LEN = 512;

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), ERR_LEN, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HALO_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return HALO_OK;
}

// Kernels that need more than the default 64 KiB of dynamic LDS must have the limit raised once per (kernel, device):
// the attribute lives in the device's copy of the code object, so a flag per kernel alone would leave the second GPU of
// a process at the default.  `seen` is the caller's static bit set (256 devices), one per kernel instantiation.
struct LdsLimitSeen { unsigned long long bits[4] = {0, 0, 0, 0}; };
inline bool raise_lds_limit(LdsLimitSeen &seen, const void *kernel, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool tracked = dev >= 0 && dev < 256;
    if (tracked && ((seen.bits[dev >> 6] >> (dev & 63)) & 1ull)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (tracked) seen.bits[dev >> 6] |= 1ull << (dev & 63);      // a lost update between threads only repeats the call
    return true;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char *base;
    size_t cap, off;
    Arena(void *p, size_t n) : base((char *)p), cap(n), off(0) {}
    template <typename T> T *take(size_t count)
    {
        size_t start = align_up(off, 256);
        size_t end = start + count * sizeof(T);
        if (end > cap) { off = cap + 1; return nullptr; }
        off = end;
        return (T *)(base + start);
    }
    bool ok() const { return off <= cap; }
};

}  // namespace halo
