// runtime.hip -- lifecycle, device buffers, timing and error plumbing of libsynthhip.so.
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <unordered_map>

namespace sh {

static thread_local char g_err[512] = "";

State& state() {
    static State s;
    return s;
}

static Knobs g_knobs;
const Knobs& knobs() { return g_knobs; }
Counters& counters() {
    static Counters c;
    return c;
}

void load_knobs() {
    auto flag = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };
    auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
    Knobs k;
    k.no_speculation = flag("SYNTHHIP_NO_SPECULATION");
    k.no_overlap = flag("SYNTHHIP_NO_OVERLAP");
    k.no_split = flag("SYNTHHIP_NO_SPLIT");
    k.no_seg = flag("SYNTHHIP_NO_SEG");
    k.no_tiles = flag("SYNTHHIP_NO_TILES");
    k.no_small_pipeline = flag("SYNTHHIP_NO_SMALL_PIPELINE");
    k.self = (int)num("SYNTHHIP_SELF", 0);
    k.rt_cus = (int)num("SYNTHHIP_RT_CUS", 0);
    k.no_ladder = flag("SYNTHHIP_NO_LADDER");
    k.no_period = flag("SYNTHHIP_NO_PERIOD");
    k.period_chunks = (int)num("SYNTHHIP_PERIOD_CHUNKS", 0);
    k.variant = (int)num("SYNTHHIP_VARIANT", 0);
    k.groups = (int)num("SYNTHHIP_GROUPS", 0);
    k.pool_fill = (int)num("SYNTHHIP_POOL_FILL", -1);
    g_knobs = k;
}

int bind_thread_to_device() {
    // (asked of the runtime every time, not remembered: another HIP user on this host thread -- a library, the caller's own
    //  hipSetDevice -- may have moved the thread's current device between two calls of ours; hipGetDevice is a thread-local read)
    const int want = state().device;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != want) SH_HIP(hipSetDevice(want));
    return SH_OK;
}

std::recursive_mutex& api_mutex() {
    static std::recursive_mutex m;
    return m;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_error(hipError_t e, const char* what) {
    int code = (e == hipErrorOutOfMemory) ? SH_ERR_NOMEM : SH_ERR_HIP;
    return set_error(code, "%s: %s", what, hipGetErrorString(e));
}

// ---- device buffer pool ------------------------------------------------------------
// hipFree synchronises the device and hipMalloc takes tens of microseconds: a chain of Sample operations (each
// producing a new buffer and dropping the old one) would stall the stream at every step.  Freed buffers are kept,
// by size class (eight classes per power of two: at most 12.5 % slack), and handed out again.  Reuse is safe without
// synchronisation because everything the library enqueues goes through one stream, in order: work that still reads a
// freed buffer precedes whatever the next owner enqueues.  (The second render stream is joined into that order before a
// buffer changes hands: sh_buf_free ends the run of renders first.  The communication stream only touches buffers that
// DistVoiceBank keeps for its lifetime.)
namespace {
struct Pool {
    std::unordered_map<size_t, std::vector<void*>> free_;
    size_t cached = 0;
    static constexpr size_t LIMIT = size_t(16) << 30;      // bytes kept at most
};
Pool g_pool;

size_t size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    size_t p = 256;
    while (p < bytes) p <<= 1;                              // 2^k >= bytes > 2^(k-1)
    const size_t step = p >> 4;                             // eight classes in (2^(k-1), 2^k]
    return (bytes + step - 1) / step * step;
}
}  // namespace

int pool_alloc(size_t bytes, void** ptr, size_t* cap) {
    const size_t c = size_class(bytes);
    auto it = g_pool.free_.find(c);
    if (it != g_pool.free_.end() && !it->second.empty()) {
        *ptr = it->second.back();
        it->second.pop_back();
        g_pool.cached -= c;
        *cap = c;
        counters().pool_hits += 1;
        return SH_OK;
    }
    counters().device_allocs += 1;
    hipError_t e = hipMalloc(ptr, c);
    if (e == hipErrorOutOfMemory && g_pool.cached) {        // give the cache back and try once more
        pool_trim();
        e = hipMalloc(ptr, c);
    }
    if (e != hipSuccess) return hip_error(e, "hipMalloc");
    *cap = c;
    return SH_OK;
}

void pool_free(void* ptr, size_t cap) {
    if (g_pool.cached + cap > Pool::LIMIT) {
        counters().stream_syncs += 1;
        counters().device_frees += 1;
        (void)hipStreamSynchronize(state().stream);
        (void)hipFree(ptr);
        return;
    }
    g_pool.free_[cap].push_back(ptr);
    g_pool.cached += cap;
}

void pool_trim() {
    if (state().stream) (void)hipStreamSynchronize(state().stream);
    for (auto& kv : g_pool.free_)
        for (void* p : kv.second) { counters().device_frees += 1; (void)hipFree(p); }
    g_pool.free_.clear();
    g_pool.cached = 0;
}

int ensure_scratch(size_t bytes) {
    // (entry points that use the scratch hold no run of renders: SH_REQUIRE_INIT joined the streams; the pool's reuse is
    // ordered by the main stream)
    State& s = state();
    if (s.scratch_bytes >= bytes) return SH_OK;
    if (s.scratch) {
        pool_free(s.scratch, s.scratch_bytes);
        s.scratch = nullptr;
        s.scratch_bytes = 0;
    }
    size_t want = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
    return pool_alloc(want, &s.scratch, &s.scratch_bytes);
}

int grow_pooled(Pooled& p, size_t bytes) {
    if (p.cap >= bytes && p.ptr) return SH_OK;
    if (p.ptr) {
        // launches on either stream may still use the old block, and the block that comes back from the pool may still be in use
        // by work on the main stream: order the two streams both ways, then hand over
        int rc = join_streams();
        if (rc) return rc;
        pool_free(p.ptr, p.cap);
        p.ptr = nullptr;
        p.cap = 0;
    }
    int rc = pool_alloc(bytes, &p.ptr, &p.cap);
    if (!rc && knobs().pool_fill >= 0) {                     // (diagnostics: poison / clear what comes from the pool)
        SH_HIP(hipMemsetAsync(p.ptr, knobs().pool_fill, p.cap, state().stream));
        rc = join_streams();
    }
    return rc;
}

void release_pooled(Pooled& p) {
    if (p.ptr) pool_free(p.ptr, p.cap);
    p.ptr = nullptr;
    p.cap = 0;
}

}  // namespace sh

using sh::state;

extern "C" {

const char* sh_last_error(void) { return sh::g_err; }

#ifndef SH_SOURCE_HASH
#define SH_SOURCE_HASH "unknown"
#endif
// "src:<hash>" = SHA-256 (first 16 hex digits) of csrc/ + include/synthhip.h + the compiler flags, embedded by
// synthesizer_amd/build.py: a library whose hash differs from the tree's is stale and gets rebuilt.
const char* sh_version(void) { return "synthhip 0.2 (gfx950) src:" SH_SOURCE_HASH; }

int sh_abi(uint32_t* out, int n) {
    const uint32_t v[7] = {SH_ABI_VERSION, (uint32_t)sizeof(sh_segment), (uint32_t)sizeof(sh_partial), (uint32_t)sizeof(sh_envelope),
                           (uint32_t)sizeof(sh_voice), (uint32_t)sizeof(sh_devinfo), (uint32_t)sizeof(sh_counters)};
    for (int k = 0; k < n && k < 7 && out; ++k) out[k] = v[k];
    return 7;
}

int sh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sh_is_initialized(void) { return state().initialized ? 1 : 0; }

int sh_init(int device) {
    SH_API_LOCK();
    sh::State& s = state();
    if (s.initialized) {
        if (s.device == device) return SH_OK;
        return sh::set_error(SH_ERR_INVALID, "already initialised on device %d", s.device);
    }
    sh::load_knobs();
    sh::counters() = sh::Counters();
    int n = sh_device_count();
    if (n <= 0) return sh::set_error(SH_ERR_NOTINIT, "no HIP device visible");
    if (device < 0 || device >= n) return sh::set_error(SH_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    SH_HIP(hipSetDevice(device));
    // SYNTHHIP_RT_CUS=n (0: off): the last n compute units are kept for real-time lanes (sh_rt_*) -- the library's streams are created with
    // a CU mask that leaves them out, a lane's stream with a mask of them alone: a mixer turn then never waits for wavefronts of a render
    // launch to retire, and every render launch pays n of the chip's CUs (profiles/r06_rt_lane.txt).
    const int rt_cus = sh::knobs().rt_cus;
    if (rt_cus > 0) {
        hipDeviceProp_t prop;
        SH_HIP(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int c = 0; c < ncu - rt_cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
        SH_HIP(hipExtStreamCreateWithCUMask(&s.stream, (uint32_t)mask.size(), mask.data()));
        SH_HIP(hipExtStreamCreateWithCUMask(&s.stream2, (uint32_t)mask.size(), mask.data()));
    } else {
        SH_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        SH_HIP(hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
    }
    SH_HIP(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&s.ev_aux, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&s.ev_prep, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&s.ev_sync, hipEventDisableTiming));
    SH_HIP(hipEventCreate(&s.ev_start));
    SH_HIP(hipEventCreate(&s.ev_stop));
    SH_HIP(hipMalloc((void**)&s.flag, sizeof(int) * 16));
    SH_HIP(hipHostMalloc((void**)&s.flag_host, sizeof(int) * 16, hipHostMallocDefault));
    SH_HIP(hipMemsetAsync(s.flag, 0, sizeof(int) * 16, s.stream));
    {   // sin/cos table for the oscillator kernels: correctly rounded doubles of sin/cos(k*2pi/512)
        std::vector<double> tab(2 * 512);
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int k = 0; k < 512; ++k) {
            long double a = two_pi * (long double)k / 512.0L;
            tab[2 * k] = (double)sinl(a);
            tab[2 * k + 1] = (double)cosl(a);
        }
        SH_HIP(hipMalloc(&s.trig, tab.size() * sizeof(double)));
        SH_HIP(hipMemcpy(s.trig, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    SH_HIP(hipStreamSynchronize(s.stream));
    s.device = device;
    s.initialized = true;
    return SH_OK;
}

int sh_shutdown(void) {
    SH_API_LOCK();
    sh::State& s = state();
    if (!s.initialized) return SH_OK;
    (void)sh_dist_shutdown();          // communicator, slot events and the communication stream go first (they belong to this device context)
    (void)hipStreamSynchronize(s.stream);
    if (sh::has_pending()) sh::flush_pending();
    (void)hipStreamSynchronize(s.stream);
    (void)hipStreamSynchronize(s.stream2);
    sh::free_render_buffers();
    if (s.scratch) sh::pool_free(s.scratch, s.scratch_bytes);
    sh::pool_trim();
    (void)hipEventDestroy(s.ev_join);
    (void)hipEventDestroy(s.ev_aux);
    (void)hipEventDestroy(s.ev_prep);
    (void)hipEventDestroy(s.ev_sync);
    (void)hipStreamDestroy(s.stream2);
    if (s.flag) (void)hipFree(s.flag);
    if (s.trig) (void)hipFree(s.trig);
    if (s.flag_host) (void)hipHostFree(s.flag_host);
    (void)hipEventDestroy(s.ev_start);
    (void)hipEventDestroy(s.ev_stop);
    (void)hipStreamDestroy(s.stream);
    s = sh::State();
    return SH_OK;
}

int sh_device_pci(char* out, int n) {
    SH_REQUIRE_INIT_KEEP_PENDING();
    if (!out || n < 16) return sh::set_error(SH_ERR_INVALID, "sh_device_pci: buffer of at least 16 bytes");
    SH_HIP(hipDeviceGetPCIBusId(out, n, sh::state().device));
    return SH_OK;
}

int sh_device_info(sh_devinfo* out) {
    SH_REQUIRE_INIT();
    if (!out) return sh::set_error(SH_ERR_INVALID, "out is NULL");
    hipDeviceProp_t p;
    SH_HIP(hipGetDeviceProperties(&p, state().device));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->clock_mhz = p.clockRate / 1000;
    out->hbm_bytes = p.totalGlobalMem;
    out->wavefront = p.warpSize;
    out->device = state().device;
    return SH_OK;
}

int sh_debug_counters(sh_counters* out) {
    SH_API_LOCK();
    if (!out) return sh::set_error(SH_ERR_INVALID, "sh_debug_counters: out is NULL");
    const sh::Counters& c = sh::counters();
    out->device_allocs = c.device_allocs;
    out->device_frees = c.device_frees;
    out->stream_syncs = c.stream_syncs;
    out->pool_hits = c.pool_hits;
    out->segmented_launches = c.segmented_launches;
    out->tiled_launches = c.tiled_launches;
    out->tiled_predicted = c.tiled_predicted;
    return SH_OK;
}

// Options are plain host state (no device work): callable before sh_init, without a GPU.
int sh_set_option(int option, int value) {
    SH_API_LOCK();
    if (option == SH_OPT_QUANTISE_ROUND) { state().quantise_round = value ? 1 : 0; return SH_OK; }
    return sh::set_error(SH_ERR_INVALID, "sh_set_option: unknown option %d", option);
}

int sh_get_option(int option) {
    SH_API_LOCK();
    if (option == SH_OPT_QUANTISE_ROUND) return state().quantise_round;
    if (option == SH_INFO_LAST_MIXDOWN_FUSED) return state().last_mixdown_fused;
    return sh::set_error(SH_ERR_INVALID, "sh_get_option: unknown option %d", option);
}

int sh_sync(void) {
    SH_REQUIRE_INIT();
    SH_HIP(hipStreamSynchronize(state().stream));
    if (state().comm_stream) SH_HIP(hipStreamSynchronize(state().comm_stream));
    return SH_OK;
}

// ---- buffers -----------------------------------------------------------------------

int sh_buf_alloc(size_t bytes, sh_buf** out) {
    SH_REQUIRE_INIT_KEEP_PENDING();          // touches no stream: a run of renders goes on (the memory's last owner flushed when freeing it)
    if (!out) return sh::set_error(SH_ERR_INVALID, "out is NULL");
    sh_buf* b = new (std::nothrow) sh_buf;
    if (!b) return sh::set_error(SH_ERR_NOMEM, "host allocation failed");
    b->ptr = nullptr;
    b->bytes = bytes;
    b->owner = true;
    if (bytes) {
        int rc = sh::pool_alloc(bytes, &b->ptr, &b->cap);
        if (rc) {
            delete b;
            return rc;
        }
    }
    *out = b;
    return SH_OK;
}

int sh_buf_view(sh_buf* parent, size_t offset, size_t bytes, sh_buf** out) {
    SH_REQUIRE_INIT_KEEP_PENDING();
    if (!parent || !out) return sh::set_error(SH_ERR_INVALID, "sh_buf_view: NULL argument");
    if (offset > parent->bytes || bytes > parent->bytes - offset) return sh::set_error(SH_ERR_INVALID, "sh_buf_view: range outside the parent buffer");
    sh_buf* b = new (std::nothrow) sh_buf;
    if (!b) return sh::set_error(SH_ERR_NOMEM, "host allocation failed");
    b->ptr = (char*)parent->ptr + offset;
    b->bytes = bytes;
    b->owner = false;
    *out = b;
    return SH_OK;
}

int sh_buf_free(sh_buf* b) {
    if (!b) return SH_OK;
    SH_API_LOCK();
    if (!b->owner) {
        delete b;
        return SH_OK;
    }
    if (b->ptr && state().initialized) {
        // a render may still owe this buffer the fold of its partial buses: enqueue it before the memory changes hands
        if (sh::has_pending()) sh::flush_pending();
        sh::pool_free(b->ptr, b->cap);
    }
    delete b;
    return SH_OK;
}

size_t sh_buf_size(const sh_buf* b) { return b ? b->bytes : 0; }
void*  sh_buf_devptr(sh_buf* b) { return b ? b->ptr : nullptr; }

static int check_range(const sh_buf* b, size_t off, size_t bytes, const char* what) {
    if (!b) return sh::set_error(SH_ERR_INVALID, "%s: buffer is NULL", what);
    if (off > b->bytes || bytes > b->bytes - off)
        return sh::set_error(SH_ERR_INVALID, "%s: range [%zu, +%zu) outside buffer of %zu bytes", what, off, bytes, b->bytes);
    return SH_OK;
}

int sh_buf_upload(sh_buf* b, size_t offset, const void* host, size_t bytes) {
    SH_REQUIRE_INIT();
    int rc = check_range(b, offset, bytes, "sh_buf_upload");
    if (rc) return rc;
    if (!bytes) return SH_OK;
    if (!host) return sh::set_error(SH_ERR_INVALID, "host pointer is NULL");
    SH_HIP(hipMemcpyAsync((char*)b->ptr + offset, host, bytes, hipMemcpyHostToDevice, state().stream));
    SH_HIP(hipStreamSynchronize(state().stream));
    return SH_OK;
}

int sh_buf_download(const sh_buf* b, size_t offset, void* host, size_t bytes) {
    SH_REQUIRE_INIT();
    int rc = check_range(b, offset, bytes, "sh_buf_download");
    if (rc) return rc;
    if (!bytes) return SH_OK;
    if (!host) return sh::set_error(SH_ERR_INVALID, "host pointer is NULL");
    SH_HIP(hipMemcpyAsync(host, (const char*)b->ptr + offset, bytes, hipMemcpyDeviceToHost, state().stream));
    SH_HIP(hipStreamSynchronize(state().stream));
    return SH_OK;
}

int sh_buf_fill_zero(sh_buf* b, size_t offset, size_t bytes) {
    SH_REQUIRE_INIT();
    int rc = check_range(b, offset, bytes, "sh_buf_fill_zero");
    if (rc) return rc;
    if (!bytes) return SH_OK;
    SH_HIP(hipMemsetAsync((char*)b->ptr + offset, 0, bytes, state().stream));
    return SH_OK;
}

int sh_buf_copy(sh_buf* dst, size_t dst_off, const sh_buf* src, size_t src_off, size_t bytes) {
    SH_REQUIRE_INIT();
    int rc = check_range(dst, dst_off, bytes, "sh_buf_copy(dst)");
    if (rc) return rc;
    rc = check_range(src, src_off, bytes, "sh_buf_copy(src)");
    if (rc) return rc;
    if (!bytes) return SH_OK;
    SH_HIP(hipMemcpyAsync((char*)dst->ptr + dst_off, (const char*)src->ptr + src_off, bytes,
                          hipMemcpyDeviceToDevice, state().stream));
    return SH_OK;
}

// ---- timing ------------------------------------------------------------------------

int sh_timer_start(void) {
    SH_REQUIRE_INIT();
    SH_HIP(hipEventRecord(state().ev_start, state().stream));
    return SH_OK;
}

int sh_timer_stop(float* elapsed_ms) {
    SH_REQUIRE_INIT();
    SH_HIP(hipEventRecord(state().ev_stop, state().stream));
    SH_HIP(hipEventSynchronize(state().ev_stop));
    float ms = 0.f;
    SH_HIP(hipEventElapsedTime(&ms, state().ev_start, state().ev_stop));
    if (elapsed_ms) *elapsed_ms = ms;
    return SH_OK;
}

}  // extern "C"
