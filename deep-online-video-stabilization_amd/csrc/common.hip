// Error reporting + library identity for libstabnet_hip.so.
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void stabnet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sn_check_device(const void* p, const char* what, hipStream_t st) {
    hipPointerAttribute_t attr;
    int cur = -1;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return STABNET_OK;
    (void)hipGetLastError();
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return STABNET_OK; }   // no runtime state to compare with
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        stabnet_set_error("%s is not a device pointer known to the HIP runtime", what);
        return STABNET_ERR_BAD_ARG;
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) {
        stabnet_set_error("%s is host memory (type %d), not device memory: the HIP path has no host fallback", what, (int)attr.type);
        return STABNET_ERR_BAD_ARG;
    }
    if (attr.type == hipMemoryTypeDevice && attr.device != cur) {
        stabnet_set_error("%s lives on device %d but device %d is current: call under torch.cuda.device(%d) / hipSetDevice",
                          what, attr.device, cur, attr.device);
        return STABNET_ERR_BAD_ARG;
    }
    return STABNET_OK;
}

extern "C" {
const char* stabnet_last_error(void) { return g_err; }
int stabnet_abi_version(void) { return 2; }
}
