// libxmh core: version, thread-local error string, device queries.
#include "xmh_common.h"

#include <string.h>

namespace xmh {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int device_cu_count() {
    static thread_local int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cached = n;
    return n;
}

}  // namespace xmh

extern "C" int xmh_version(void) { return 100; }

extern "C" const char* xmh_last_error(void) { return xmh::g_err; }
