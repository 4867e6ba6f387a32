// runtime.hip — error reporting and library identification for libhcp_mi355x.so.
#include "hcp_common.h"

static thread_local char g_err[512] = "";

extern "C" int hcp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

// Thread-local message of the last failing call on this thread ("" if none).
HCP_API const char* hcp_last_error(void) { return g_err; }

// 1 when this object was built by tests/emu (CPU interpreter), 0 for the gfx950 product library.
HCP_API int hcp_is_emulated(void) { return HCP_IS_EMULATED; }

HCP_API int hcp_abi_version(void) { return 1; }
