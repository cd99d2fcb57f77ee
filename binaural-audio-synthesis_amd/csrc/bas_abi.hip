// ABI bookkeeping of libbas_hip.so: version, thread-local error text, launch checks.
#include "bas_internal.h"

static thread_local char g_err[512] = "";

void bas_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int bas_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int bas_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    return bas_fail((int)e, "%s: %s", what, hipGetErrorString(e));
}

extern "C" int bas_version(void) { return BAS_ABI_VERSION; }

extern "C" const char *bas_last_error(void) { return g_err; }
