// Shared host/device helpers for libstabnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#define STABNET_OK 0
#define STABNET_ERR_BAD_ARG (-1)
#define STABNET_ERR_LAUNCH (-2)
#define STABNET_ERR_WORKSPACE (-3)

void stabnet_set_error(const char* fmt, ...);

#define SN_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            stabnet_set_error(__VA_ARGS__);      \
            return STABNET_ERR_BAD_ARG;          \
        }                                        \
    } while (0)

#define SN_LAUNCH_CHECK(what)                                                    \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            stabnet_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return STABNET_ERR_LAUNCH;                                           \
        }                                                                        \
    } while (0)

// 0 when `p` is memory of the CURRENT device (the one kernels of this thread launch on); an error otherwise: a pointer of
// another GPU passed with the wrong device current would fault inside a kernel instead.
int sn_check_device(const void* p, const char* what, hipStream_t st = nullptr);   // (skipped while `st` is being captured)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
