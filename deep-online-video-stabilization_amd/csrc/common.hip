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

extern "C" {
const char* stabnet_last_error(void) { return g_err; }
int stabnet_abi_version(void) { return 1; }
}
