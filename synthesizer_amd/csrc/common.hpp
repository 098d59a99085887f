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
// bus.  In a stream of renders that fold is done by the first workgroups of the render kernel TWO launches later (no
// kernel of its own, no launch boundary); any other API call folds what is outstanding first (SH_REQUIRE_INIT ->
// flush_pending).  The run, the folds owed and the ring of partial-bus buffers belong to the BANK (osc.hip, struct sh_bank): banks
// rendering turn by turn each keep their pipeline; only a render that breaks its own bank's run folds that bank's leftovers.  Two launches, not one: consecutive renders alternate between two HIP streams so that the tail of one
// launch (its last, partly empty round of workgroups, the launch gap) is filled by the next -- which therefore must not
// depend on its predecessor: launch n reads the launch records that launch n-2 resolved and folds launch n-2's partial
// buses (same stream), and shares nothing with launch n-1.
struct PendingCombine {
    const void* parts = nullptr;     // double2[groups][nframes]
    uint32_t groups = 0, nframes = 0;
    void*    o32 = nullptr;          // float2[nframes] or NULL
    void*    o64 = nullptr;          // double2[nframes] or NULL
    void*    o16 = nullptr;          // int16 stereo PCM (one dword per frame) or NULL
    double   scale = 0.0;            // ... quantised as trunc(scale * float32(bus)), saturating
    const void* gen_valid = nullptr; // split launch: uint32[groups], which of the general kernel's parts[groups + g] exist
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
    int         last_mixdown_fused = 0;   // stretches of the last sh_bank_mixdown_i16 call that were folded where the samples are made (sh_get_option)
    int         quantise_round = 0;    // sh_set_option(SH_OPT_QUANTISE_ROUND): 1 = round half to even instead of truncation
    int*        flag = nullptr;        // device int: overflow flag for quantise
    int*        flag_host = nullptr;   // pinned host mirror
    void*       trig = nullptr;        // device: 512 x (sin, cos) of k*2pi/512, float64 (devmath.hpp sincos_tab)
    // Bank renders that still owe the fold of their partial buses (over all banks: each bank keeps its own run of pipelined
    // launches, its pending folds and its ring of partial-bus buffers -- osc.hip, struct sh_bank)
    int         pending_total = 0;
    int         open_runs = 0;         // banks whose run of pipelined renders is open (single-group banks owe no fold, but their
                                       // launches alternate streams all the same: other entry points must join them first)
    // second render stream and the events that tie it to `stream`
    hipStream_t stream2 = nullptr;
    hipEvent_t  ev_join = nullptr;      // on `stream` when a run of renders starts: stream2's first launch of the run waits for it
    hipEvent_t  ev_aux = nullptr;       // on stream2 after every launch there: `stream` waits for it when the run ends
    hipEvent_t  ev_prep = nullptr;      // on `stream` after a prepare kernel that a stream2 launch needs
    hipEvent_t  ev_sync = nullptr;      // join_streams: on `stream`, waited for by stream2
    bool        aux_busy = false;       // stream2 holds work `stream` has not waited for
};

State& state();

// Measurement knobs, read from the environment ONCE (sh_init) -- include/synthhip.h lists them.  None of them changes a result;
// they select between code paths that produce the same buses so that A/B timings can be taken with one library.
struct Knobs {
    bool no_speculation = false;   // SYNTHHIP_NO_SPECULATION=1: launch records by a prepare kernel in front of every render
    bool no_overlap = false;       // SYNTHHIP_NO_OVERLAP=1: consecutive renders stay on one stream
    bool no_split = false;         // SYNTHHIP_NO_SPLIT=1: lean and general code in one kernel
    bool no_seg = false;           // SYNTHHIP_NO_SEG=1: transition launches / row heads are not cut into segments
    bool no_tiles = false;         // SYNTHHIP_NO_TILES=1: banks whose notes do not move in lock-step are not classified tile by tile
    int  self = 0;                 // SYNTHHIP_SELF: a render that stands alone resolves its records (1, 3) / folds its partial buses (1, 2) inside its one kernel.  Off: measured slower than the prepare kernel + k_bus_combine (profiles/r06_run_lengths.txt)
    bool no_ladder = false;        // SYNTHHIP_NO_LADDER=1: a launch that stands alone keeps its wavefronts at one priority (LaunchArgs::alone; profiles/r06_prio_ladder.txt)
    bool no_period = false;        // SYNTHHIP_NO_PERIOD=1: 16-bit mono resampling between rates with a short period goes through k_resample_small (rounds 1-5), not k_resample_period_i16
    int  period_chunks = 0;        // SYNTHHIP_PERIOD_CHUNKS=n: consecutive chunks per workgroup of k_resample_period_i16 (0: by the rates' ratio)
    int  rt_cus = 0;               // SYNTHHIP_RT_CUS=n: the last n compute units are kept for real-time lanes (the library's streams leave them out)
    bool no_small_pipeline = false;// SYNTHHIP_NO_SMALL_PIPELINE=1: single-group banks render on one stream (round-2 behaviour)
    int  variant = 0;              // SYNTHHIP_VARIANT=WFM: waves, frames per lane, min waves per SIMD of the render kernel (484, 444, 844, 821, 421, 211)
    int  groups = 0;               // SYNTHHIP_GROUPS: voice groups of a render launch
    int  pool_fill = -1;           // SYNTHHIP_POOL_FILL=0..255: blocks that grow are filled with this byte first (diagnostics)
};
const Knobs& knobs();
void load_knobs();                 // sh_init

// What the library has done behind the caller's back since sh_init (sh_debug_counters): a streaming caller's steady state
// must leave all of them unchanged.
struct Counters {
    uint64_t device_allocs = 0;    // hipMalloc calls (pool misses included)
    uint64_t device_frees = 0;     // hipFree calls
    uint64_t stream_syncs = 0;     // host-side waits the library inserted on its own (growing a buffer; NOT the caller's sh_sync / downloads)
    uint64_t pool_hits = 0;        // buffers handed out again without touching the driver
    uint64_t segmented_launches = 0, tiled_launches = 0, tiled_predicted = 0;     // which shape the bank renders took
};
Counters& counters();
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
// A pool-backed device block that only grows (partial buses, record sets of segmented launches): growing it makes both render
// streams wait for each other (events, no host synchronisation), gives the old block back to the pool and takes a larger one --
// the contents are NOT kept.
struct Pooled {
    void*  ptr = nullptr;
    size_t cap = 0;
};
int  join_streams();                   // stream waits for stream2 and stream2 for stream (osc_render.hip)
int  grow_pooled(Pooled& p, size_t bytes);
void release_pooled(Pooled& p);
int  flush_pending();                  // end every bank's run of renders: join stream2, fold all outstanding partial buses now (osc.hip)
void free_render_buffers();            // the banks' partial-bus rings (sh_shutdown)
bool fold_owed_into(const void* p, size_t bytes);   // does any bank still owe a fold into this memory? (osc_render.hip)
inline bool has_pending() { return state().pending_total != 0 || state().aux_busy || state().open_runs != 0; }
int  bus_finalize_on(hipStream_t st, const double* in, size_t nvalues, float* out);   // float64 bus -> float32 (osc.hip)

// HIP's current device is a property of the HOST THREAD: a thread other than the one that called sh_init (the reference drives its
// mixer from two) starts on device 0 -- on rank 3 of a multi-GPU job its allocations and launches would land on the wrong GPU.  Every
// entry point binds its thread to the library's device once (round 4; sh_init(device != 0) has yet to run on hardware: DESIGN section 5).
int bind_thread_to_device();           // runtime.hip
#define SH_REQUIRE_INIT_KEEP_PENDING()                                                 \
    SH_API_LOCK();                                                                     \
    do {                                                                               \
        if (!sh::state().initialized)                                                  \
            return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");     \
        { int rc_bind__ = sh::bind_thread_to_device(); if (rc_bind__) return rc_bind__; } \
    } while (0)

#define SH_REQUIRE_INIT()                                                              \
    SH_API_LOCK();                                                                     \
    do {                                                                               \
        if (!sh::state().initialized)                                                  \
            return sh::set_error(SH_ERR_NOTINIT, "sh_init() has not been called");     \
        { int rc_bind__ = sh::bind_thread_to_device(); if (rc_bind__) return rc_bind__; } \
        if (sh::has_pending()) {                                                       \
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

// A dispatch counts the WORK-ITEMS of each grid dimension in 32 bits: a launch with grid.x * block.x >= 2^32 wraps silently
// (a kernel over 2^32 samples runs a handful of blocks and reports success -- tests/test_gpu_huge.py).  One-dimensional
// launches of unbounded length therefore fold their blocks into two dimensions (the kernels number them with block_id() and
// already guard against blocks beyond the data), and launches whose x dimension is a tile count are cut into several by
// their entry points.
constexpr unsigned GRID_X_MAX = 1u << 21;            // x at most 1024 threads: 2^31 work-items
inline dim3 grid1d(size_t n, size_t per_block) {
    size_t nb = (n + per_block - 1) / per_block;
    if (nb == 0) nb = 1;
    if (nb <= GRID_X_MAX) return dim3((unsigned)nb);
    return dim3(GRID_X_MAX, (unsigned)((nb + GRID_X_MAX - 1) / GRID_X_MAX));
}
__device__ __forceinline__ size_t block_id() { return (size_t)blockIdx.y * gridDim.x + blockIdx.x; }

// Rows that are read exactly once: streaming (non-temporal) loads are worth +10..13 % once the data no longer fits the 256 MB
// Infinity Cache (mixers: 6.4 -> 7.2 TB/s on 2 GB) and cost 11 % when it does (36 MB: 8.9 -> 7.9 TB/s) -- the entry points
// choose by the bytes a call reads (CHANGELOG.md item 28).
constexpr size_t STREAM_BYTES = (size_t)128 << 20;
template <bool NT, typename V>
__device__ __forceinline__ V load_vec(const void* p) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p)) : *reinterpret_cast<const V*>(p);
}

// ---- 24-bit PCM (audioop width 3) -----------------------------------------------------------------------------------
// audioop reads a 3-byte sample as GETINT24 (sign-extended) and, wherever it works through GETSAMPLE32 (ratecv, lin2lin) or
// scales linearly (mul, tomono, tostereo, add, bias), the result for width 3 equals the width-4 operation on value << 8 taken
// >> 8 (DESIGN.md section 4b spells the argument out per call).  The library therefore unpacks 3-byte samples into int32
// temporaries (pcm_ops.hip), runs the 32-bit kernels, and packs the result: two extra streaming passes, no new arithmetic.
int unpack24(const void* in, size_t nsamples, int shift, int32_t* out);    // out[i] = int24(in[3i..]) << shift
int pack24(const int32_t* in, size_t nsamples, int shift, void* out);      // 3 low bytes of in[i] >> shift
struct Temp {                       // a pool-backed device temporary wrapped as a (non-owning) sh_buf
    sh_buf buf{nullptr, 0, false, 0};
    size_t cap = 0;
    int  alloc(size_t bytes) {
        if (!bytes) bytes = 4;
        int rc = pool_alloc(bytes, &buf.ptr, &cap);
        buf.bytes = bytes;
        return rc;
    }
    ~Temp() { if (buf.ptr) pool_free(buf.ptr, cap); }       // stream-ordered reuse: see the pool's comment in runtime.hip
};

}  // namespace sh
