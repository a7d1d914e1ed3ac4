// Error reporting + ABI version of librnr_hip.so.
#include "rnr_internal.h"
#include <string.h>

namespace rnr {

static thread_local char g_err[512] = {0};

char* err_buf() { return g_err; }

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

}  // namespace rnr

extern "C" int rnr_abi_version(void) { return RNR_ABI_VERSION; }
extern "C" const char* rnr_last_error(void) { return rnr::g_err; }
