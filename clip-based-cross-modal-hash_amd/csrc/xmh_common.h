// Shared host-side helpers for libxmh.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/xmh.h"

namespace xmh {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// Launch check: HIP reports bad configs at launch; asynchronous faults surface at the caller's sync.
#define XMH_LAUNCH_CHECK(what)                                                                        \
    do {                                                                                              \
        hipError_t e__ = hipGetLastError();                                                           \
        if (e__ != hipSuccess) return ::xmh::fail(XMH_EHIP, "%s: %s", what, hipGetErrorString(e__));  \
    } while (0)

#define XMH_HIP(call)                                                                                          \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess) return ::xmh::fail(XMH_EHIP, "%s: %s", #call, hipGetErrorString(e__));          \
    } while (0)

inline hipStream_t as_stream(xmh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Pair cache of the ranking scan, entries of codes above 64 bits and of ternary codes (distance << 1 | relevant, at most 1025): round 6
// stores them 12 bits each -- the 8 entries of a lane and batch are THREE dwords (entry k = bits [12 k, 12 k + 12) of the 96), a record
// row of 64 lanes 768 bytes -- instead of 16 bits each in four.  Pass 2 of such codes is bound by the stream of this cache (configs[4]'s
// shard: 12.5 GB per pass at 5 TB/s); a quarter of the bytes is a quarter of that time.  Entries 2 and 5 straddle a dword.
constexpr int kCache12Dwords = 3;
// entry k (compile-time) of a record -> (the dword to extract from, bit offset in it); x = w1:w0 >> 24, y = w2:w1 >> 28 hold the straddlers
#define XMH_CACHE12_WORD(k, w0, w1, w2, x, y) ((k) < 2 ? (w0) : (k) == 2 ? (x) : (k) < 5 ? (w1) : (k) == 5 ? (y) : (w2))
#define XMH_CACHE12_BIT(k) ((k) == 0 ? 0 : (k) == 1 ? 12 : (k) == 2 ? 0 : (k) == 3 ? 4 : (k) == 4 ? 16 : (k) == 5 ? 0 : (k) == 6 ? 8 : 20)

int device_cu_count();

// Opt `kern` in to `bytes` (> 64 KB) of dynamic LDS on the CURRENT device.  hipFuncSetAttribute applies to the current device only
// and costs a few microseconds, so the (device, kernel) -> size pairs already raised are remembered (mutex-guarded: host threads and
// devices may interleave) and repeated calls stay off the launch path.  Returns XMH_OK or XMH_EHIP with the error text set.
int raise_dynamic_lds(const void* kern, size_t bytes, const char* who);

// Optional per-kernel timing for bench.py (xmh_prof_enable): HIP events recorded on the launch stream right
// around ONE kernel launch, keyed by a short name.  Disabled by default: no events, no overhead.
struct ProfScope {
    ProfScope(const char* name, hipStream_t st);
    ~ProfScope();
    int slot;
    hipStream_t st;
};

// roctx range around a phase of the path (tower forward, head, pack, pass 1, pass 2, top-k phases, ...): shows up in a
// `rocprofv3 --marker-trace` timeline.  Off unless xmh_prof_enable(2 or 3): then librocprofiler-sdk-roctx.so (or libroctx64.so) is
// looked up with dlopen -- libxmh.so itself does not link it -- and every scope is one push / pop pair; off = one branch.
struct RangeScope {
    explicit RangeScope(const char* name);
    ~RangeScope();
    void end();                                            // close the range before the scope does
    bool on;
};
#define XMH_RANGE(name) ::xmh::RangeScope xmh_range_scope_(name)

}  // namespace xmh

// Tuning switches (tile rules, query-group shapes, A/B toggles of tools/) exist only in a library built with -DXMH_EXPERIMENTS; the shipped
// build reads none of them (the library keeps no process-wide state beyond the eight documented XMH_* switches).
#ifdef XMH_EXPERIMENTS
#include <stdlib.h>
inline const char* xmh_experiment_env(const char* name) { return getenv(name); }
#else
inline const char* xmh_experiment_env(const char*) { return nullptr; }
#endif

