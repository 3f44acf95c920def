// libxmh core: version, thread-local error string, device queries.
#include "xmh_common.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

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

int raise_dynamic_lds(const void* kern, size_t bytes, const char* who) {
    if (bytes <= 64 * 1024) return XMH_OK;
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> raised[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(XMH_EHIP, "%s: no current device", who);
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < kMaxDev) {
        auto it = raised[dev].find(kern);
        if (it != raised[dev].end() && bytes <= it->second) return XMH_OK;
    }
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(XMH_EHIP, "%s: cannot raise dynamic LDS to %zu: %s", who, bytes, hipGetErrorString(e));
    if (dev >= 0 && dev < kMaxDev) raised[dev][kern] = bytes;
    return XMH_OK;
}

// ---- per-kernel event timing (bench only) ----------------------------------------------------------
constexpr int kProfSlots = 32;
struct ProfSlot {
    char name[48];
    hipEvent_t a, b;
    bool made, pending;
    double total_ms;
    long count;
};
static ProfSlot g_prof[kProfSlots];
static int g_prof_n = 0;
static bool g_prof_on = false;

static int prof_slot(const char* name) {
    for (int i = 0; i < g_prof_n; ++i)
        if (!strcmp(g_prof[i].name, name)) return i;
    if (g_prof_n == kProfSlots) return -1;
    ProfSlot& s = g_prof[g_prof_n];
    snprintf(s.name, sizeof(s.name), "%s", name);
    s.made = s.pending = false;
    s.total_ms = 0.0;
    s.count = 0;
    return g_prof_n++;
}

static void prof_collect(ProfSlot& s) {
    if (!s.pending) return;
    float ms = 0.f;
    if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
        s.total_ms += ms;
        s.count += 1;
    }
    s.pending = false;
}

// ---- roctx ranges (rocprofv3 --marker-trace) -------------------------------------------------------
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
static bool g_range_on = false;
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;

static bool roctx_lookup() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* so : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(so, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
            g_roctx_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
            if (g_roctx_push && g_roctx_pop) return;
            g_roctx_push = nullptr;
            g_roctx_pop = nullptr;
        }
    });
    return g_roctx_push != nullptr;
}

RangeScope::RangeScope(const char* name) : on(g_range_on) {
    if (on) g_roctx_push(name);
}

RangeScope::~RangeScope() { end(); }

void RangeScope::end() {
    if (on) g_roctx_pop();
    on = false;
}

ProfScope::ProfScope(const char* name, hipStream_t stream) : slot(-1), st(stream) {
    if (!g_prof_on) return;
    slot = prof_slot(name);
    if (slot < 0) return;
    ProfSlot& s = g_prof[slot];
    if (!s.made) {
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) {
            slot = -1;
            return;
        }
        s.made = true;
    }
    prof_collect(s);                 // fold the previous launch of this kernel before re-recording
    (void)hipEventRecord(s.a, st);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    (void)hipEventRecord(g_prof[slot].b, st);
    g_prof[slot].pending = true;
}

}  // namespace xmh

extern "C" int xmh_prof_enable(int on) {
    if ((on & 2) && !xmh::roctx_lookup()) return xmh::fail(XMH_ENOTSUP, "xmh_prof_enable: no roctx library (librocprofiler-sdk-roctx.so / libroctx64.so) on the loader path");
    xmh::g_range_on = (on & 2) != 0;
    xmh::g_prof_on = (on & 1) != 0;
    for (int i = 0; i < xmh::g_prof_n; ++i) {
        xmh::prof_collect(xmh::g_prof[i]);
        xmh::g_prof[i].total_ms = 0.0;
        xmh::g_prof[i].count = 0;
    }
    return XMH_OK;
}

extern "C" int xmh_prof_read(const char* name, double* avg_ms, int64_t* launches) {
    if (!name || !avg_ms || !launches) return xmh::fail(XMH_EINVAL, "xmh_prof_read: null argument");
    for (int i = 0; i < xmh::g_prof_n; ++i) {
        if (!strcmp(xmh::g_prof[i].name, name)) {
            xmh::prof_collect(xmh::g_prof[i]);
            *launches = xmh::g_prof[i].count;
            *avg_ms = xmh::g_prof[i].count ? xmh::g_prof[i].total_ms / (double)xmh::g_prof[i].count : 0.0;
            return XMH_OK;
        }
    }
    *launches = 0;
    *avg_ms = 0.0;
    return XMH_OK;
}

extern "C" int xmh_range_push(const char* name) {
    if (!name) return xmh::fail(XMH_EINVAL, "xmh_range_push: null name");
    if (xmh::g_range_on) xmh::g_roctx_push(name);
    return XMH_OK;
}

extern "C" int xmh_range_pop(void) {
    if (xmh::g_range_on) xmh::g_roctx_pop();
    return XMH_OK;
}

extern "C" int xmh_version(void) { return 100; }

#ifndef XMH_BUILD_ID
#define XMH_BUILD_ID "unknown"
#endif
extern "C" const char* xmh_build_id(void) { return XMH_BUILD_ID; }

extern "C" const char* xmh_last_error(void) { return xmh::g_err; }
