// common.hpp -- shared state and helpers of libsynthhip.so (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <string>
#include "../../include/synthhip.h"

struct sh_buf {
    void*  ptr;
    size_t bytes;
    bool   owner;      // false for views created by sh_buf_view (they alias a parent allocation)
    size_t cap = 0;    // bytes actually allocated (size class of the buffer pool); 0 for views
};

namespace sh {

// A bank render with several voice groups leaves float64 partial buses that still have to be folded into the caller's
// bus.  In a stream of renders that fold is done by the NEXT render kernel's first workgroups (no kernel of its own, no
// launch boundary); any other API call folds it first (SH_REQUIRE_INIT -> flush_pending).
struct PendingCombine {
    bool     active = false;
    const void* parts = nullptr;     // double2[groups][nframes]
    uint32_t groups = 0, nframes = 0, tile_frames = 0;
    void*    o32 = nullptr;          // float2[nframes] or NULL
    void*    o64 = nullptr;          // double2[nframes] or NULL
};

struct State {
    bool        initialized = false;
    int         device = -1;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;     // collectives (created by sh_dist_init): the reduce of block s overlaps the render of block s+1
    hipEvent_t  ev_start = nullptr, ev_stop = nullptr;
    // grow-only scratch used by kernels that need a second pass (partial buses, scan sums, flags)
    void*       scratch = nullptr;
    size_t      scratch_bytes = 0;
    int*        flag = nullptr;        // device int: overflow flag for quantise
    int*        flag_host = nullptr;   // pinned host mirror
    void*       trig = nullptr;        // device: 512 x (sin, cos) of k*2pi/512, float64 (devmath.hpp sincos_tab)
    // partial buses of bank renders, double-buffered: one may still await its fold while the next launch fills the other
    void*       parts_buf[2] = {nullptr, nullptr};
    size_t      parts_bytes[2] = {0, 0};
    int         parts_cur = 0;
    PendingCombine pending;
};

State& state();
// Every extern "C" entry point holds this for its whole body: one stream, one scratch buffer, one pool and one pending
// fold are shared by all callers, and the real-time mixer is driven from two threads (one adds samples, one pulls
// chunks).  Recursive: entry points call each other.
std::recursive_mutex& api_mutex();
#define SH_API_LOCK() std::lock_guard<std::recursive_mutex> sh_api_lock__(sh::api_mutex())
int  set_error(int code, const char* fmt, ...);
int  hip_error(hipError_t e, const char* what);
int  ensure_scratch(size_t bytes);
int  pool_alloc(size_t bytes, void** ptr, size_t* cap);   // device buffer pool (runtime.hip)
void pool_free(void* ptr, size_t cap);
void pool_trim();
int  flush_pending();                  // fold a pending bank-render combine now (osc.hip)
int  bus_finalize_on(hipStream_t st, const double* in, size_t nvalues, float* out);   // float64 bus -> float32 (osc.hip)

#define SH_REQUIRE_INIT_KEEP_PENDING()                                                 \
    SH_API_LOCK();                                                                     \
    do {                                                                               \
        if (!sh::state().initialized)                                                  \
            return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");     \
    } while (0)

#define SH_REQUIRE_INIT()                                                              \
    SH_API_LOCK();                                                                     \
    do {                                                                               \
        if (!sh::state().initialized)                                                  \
            return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");     \
        if (sh::state().pending.active) {                                              \
            int rc_pending__ = sh::flush_pending();                                    \
            if (rc_pending__) return rc_pending__;                                     \
        }                                                                              \
    } while (0)

#define SH_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) return sh::hip_error(e__, #call);                       \
    } while (0)

#define SH_CHECK_LAUNCH(name)                                                          \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) return sh::hip_error(e__, name);                        \
    } while (0)

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace sh
