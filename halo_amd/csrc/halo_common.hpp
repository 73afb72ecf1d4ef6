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
