// Shared host-side helpers of libsdn_hip.so (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/sdn_hip.h"

#define SDN_API extern "C" __attribute__((visibility("default")))

namespace sdn {

char* error_slot();  // thread-local buffer, 512 bytes

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_slot(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SDN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SDN_OK;
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace sdn
