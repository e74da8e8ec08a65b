// host_common.h -- error plumbing and small RAII helpers for the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/nnpops_hip.h"

namespace nnpops {

std::string& last_error_slot();   // thread-local, defined in capi_common.hip

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

#define NNPOPS_HIP_TRY(expr)                                                                            \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return ::nnpops::fail(NNPOPS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                                                  \
    } while (0)

#define NNPOPS_REQUIRE(cond, ...)                                                  \
    do {                                                                           \
        if (!(cond)) return ::nnpops::fail(NNPOPS_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

// Scoped hipSetDevice: the caller's current device is restored on exit (the reference's binding
// calls cudaSetDevice and leaves it changed -- SymmetryFunctions.cpp:129; we do not leak that).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename T>
inline int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    NNPOPS_HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return NNPOPS_OK;
}

template <typename T>
inline void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace nnpops
