// Shared host-side helpers of libppn (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ppn.h"

namespace ppn {

char* error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define PPN_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ::ppn::fail(PPN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                              \
    } while (0)

#define PPN_LAUNCH_CHECK() PPN_HIP_CHECK(hipGetLastError())

// Raise a kernel's dynamic-LDS limit only when it grows (the attribute is sticky; a hipFuncSetAttribute per
// launch costs host time on the launch path).  One process drives one GPU in this design.
#define PPN_LDS_ONCE(var, fn, attr, bytes)                              \
    do {                                                                \
        if ((bytes) > (var)) {                                          \
            PPN_HIP_CHECK(hipFuncSetAttribute(fn, attr, bytes));        \
            (var) = (bytes);                                            \
        }                                                               \
    } while (0)

}  // namespace ppn

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits until every
// global store (and atomic) this wave has issued is acknowledged -- in a conv epilogue or at the top of a persistent
// tile loop that puts one full store round trip on the critical path of every chunk / tile.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
