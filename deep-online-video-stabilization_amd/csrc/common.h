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

#ifdef __HIPCC__
// Workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2.  A kernel whose neighbouring workgroups re-read each
// other's input (pool windows, sampler taps) works as id sn_xcd_band(blockIdx.x, gridDim.x) instead: the ids of one XCD then
// cover ONE contiguous eighth of the range (balanced when the count is not a multiple of 8), and the shared input is fetched
// into one L2 instead of several.  A bijection on [0, nb).
__device__ __forceinline__ unsigned sn_xcd_band(unsigned b, unsigned nb) {
    if (nb < 8u) return b;
    const unsigned xcd = b & 7u, per = nb >> 3, rem = nb & 7u;
    return xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
}
#endif
