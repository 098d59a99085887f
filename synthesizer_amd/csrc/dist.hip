// dist.hip -- multi-GPU voice sharding: one process per GPU, each renders the partial stereo bus
// of its shard of the voice table; the partial buses (float64, frames x 2) are summed by RCCL
// over xGMI.  The reference is single-process; this exchange is the only collective on the path.
//
// The message is tiny (48 000 frames x 16 B = 768 KB per second of audio), so the collective is
// latency-bound: callers batch >= 1 s per call.  librccl.so is dlopen()ed on first use so that
// single-GPU users never pay for loading it.
#include "common.hpp"
#include <stdlib.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // (optional: sh_dist_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = -1, world = 0;
    double* token = nullptr;     // 1-element device buffer for the barrier
    // pipelined reduce: the collective of block s runs on comm_stream while the main stream renders s+1
    static constexpr int SLOTS = 4;
    hipEvent_t ev_rendered[SLOTS] = {};
    hipEvent_t ev_rendered2[SLOTS] = {};      // (the second render stream's, for sh_dist_reduce_bus_lagged)
    bool       marked[SLOTS] = {};            // sh_dist_mark_slot has recorded the two events for the slot's current contents
    hipEvent_t ev_done[SLOTS] = {};
};

Rccl& R() {
    static Rccl r;
    return r;
}

int load_rccl() {
    Rccl& r = R();
    if (r.lib) return SH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return sh::set_error(SH_ERR_RCCL, "cannot load librccl.so: %s", dlerror());
#define SH_SYM(field, sym)                                                              \
    *(void**)(&r.field) = dlsym(r.lib, sym);                                            \
    if (!r.field) return sh::set_error(SH_ERR_RCCL, "librccl.so lacks %s", sym);
    SH_SYM(GetUniqueId, "ncclGetUniqueId")
    SH_SYM(CommInitRank, "ncclCommInitRank")
    SH_SYM(CommDestroy, "ncclCommDestroy")
    SH_SYM(Reduce, "ncclReduce")
    SH_SYM(AllReduce, "ncclAllReduce")
    SH_SYM(GetErrorString, "ncclGetErrorString")
#undef SH_SYM
    *(void**)(&r.CommCount) = dlsym(r.lib, "ncclCommCount");
    *(void**)(&r.CommUserRank) = dlsym(r.lib, "ncclCommUserRank");
    *(void**)(&r.GetVersion) = dlsym(r.lib, "ncclGetVersion");
    return SH_OK;
}

int rccl_error(ncclResult_t e, const char* what) {
    return sh::set_error(SH_ERR_RCCL, "%s: %s", what, R().GetErrorString ? R().GetErrorString(e) : "rccl error");
}

#define SH_RCCL(call)                                                                   \
    do {                                                                                \
        ncclResult_t e__ = (call);                                                      \
        if (e__ != ncclSuccess) return rccl_error(e__, #call);                          \
    } while (0)

}  // namespace

extern "C" {

int sh_dist_unique_id(void* id128) {
    if (!id128) return sh::set_error(SH_ERR_INVALID, "sh_dist_unique_id: NULL");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    SH_RCCL(R().GetUniqueId(&id));
    static_assert(sizeof(id) == SH_DIST_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return SH_OK;
}

int sh_dist_init(int rank, int world, const void* id128) {
    SH_REQUIRE_INIT();
    if (!id128 || world < 1 || rank < 0 || rank >= world) return sh::set_error(SH_ERR_INVALID, "sh_dist_init: bad rank/world");
    int rc = load_rccl();
    if (rc) return rc;
    Rccl& r = R();
    if (r.comm) return sh::set_error(SH_ERR_INVALID, "sh_dist_init: already initialised");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    SH_RCCL(r.CommInitRank(&r.comm, world, id, rank));
    fflush(stdout);      // RCCL prints a version banner through stdio: push it out now, not at exit after the caller's output
    r.rank = rank;
    r.world = world;
    SH_HIP(hipMalloc((void**)&r.token, sizeof(double)));
    SH_HIP(hipMemsetAsync(r.token, 0, sizeof(double), sh::state().stream));
    // The communication stream is created like the render streams.  (Measured in round 3: a stream of another PRIORITY gets a hardware
    // queue of its own, but with two priority levels active the render launches themselves slow down -- 59 and 60 us per block against
    // 46; what keeps the exchange off the renders' critical path is WHEN it is enqueued: sh_dist_mark_slot.)
    SH_HIP(hipStreamCreateWithFlags(&sh::state().comm_stream, hipStreamNonBlocking));
    for (int k = 0; k < Rccl::SLOTS; ++k) {
        SH_HIP(hipEventCreateWithFlags(&r.ev_rendered[k], hipEventDisableTiming));
        SH_HIP(hipEventCreateWithFlags(&r.ev_rendered2[k], hipEventDisableTiming));
        SH_HIP(hipEventCreateWithFlags(&r.ev_done[k], hipEventDisableTiming));
    }
    return SH_OK;
}

int sh_dist_shutdown(void) {
    SH_API_LOCK();
    Rccl& r = R();
    if (r.comm) {
        if (sh::state().initialized) {
            (void)hipStreamSynchronize(sh::state().stream);
            if (sh::state().comm_stream) (void)hipStreamSynchronize(sh::state().comm_stream);
        }
        r.CommDestroy(r.comm);
        r.comm = nullptr;
        for (int k = 0; k < Rccl::SLOTS; ++k) {
            if (r.ev_rendered[k]) { (void)hipEventDestroy(r.ev_rendered[k]); r.ev_rendered[k] = nullptr; }
            if (r.ev_rendered2[k]) { (void)hipEventDestroy(r.ev_rendered2[k]); r.ev_rendered2[k] = nullptr; }
            if (r.ev_done[k]) { (void)hipEventDestroy(r.ev_done[k]); r.ev_done[k] = nullptr; }
        }
        if (sh::state().comm_stream) { (void)hipStreamDestroy(sh::state().comm_stream); sh::state().comm_stream = nullptr; }
    }
    if (r.token) {
        (void)hipFree(r.token);
        r.token = nullptr;
    }
    r.rank = -1;
    r.world = 0;
    return SH_OK;
}

int sh_dist_rank(void) { return R().rank; }
int sh_dist_world(void) { return R().world; }

int sh_dist_comm_info(int32_t* out, int n) {
    SH_API_LOCK();
    Rccl& r = R();
    int32_t v[5] = {r.comm ? 1 : 0, -1, -1, 0, r.lib ? 1 : 0};
    if (r.comm) {
        int x = -1;
        if (r.CommCount && r.CommCount(r.comm, &x) == ncclSuccess) v[1] = x;
        x = -1;
        if (r.CommUserRank && r.CommUserRank(r.comm, &x) == ncclSuccess) v[2] = x;
    }
    if (r.lib && r.GetVersion) {
        int x = 0;
        if (r.GetVersion(&x) == ncclSuccess) v[3] = x;
    }
    for (int k = 0; k < n && k < 5 && out; ++k) out[k] = v[k];
    return 5;
}

int sh_dist_reduce_bus(sh_buf* bus_f64, size_t nvalues, int root) {
    SH_REQUIRE_INIT();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_reduce_bus: sh_dist_init not called");
    if (!bus_f64 || bus_f64->bytes < nvalues * 8) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus: buffer too small");
    if (root < 0 || root >= r.world) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus: bad root");
    if (!nvalues) return SH_OK;
    SH_RCCL(r.Reduce(bus_f64->ptr, bus_f64->ptr, nvalues, ncclFloat64, ncclSum, root, r.comm, sh::state().stream));
    return SH_OK;
}

int sh_dist_allreduce_bus(sh_buf* bus_f64, size_t nvalues) {
    SH_REQUIRE_INIT();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_allreduce_bus: sh_dist_init not called");
    if (!bus_f64 || bus_f64->bytes < nvalues * 8) return sh::set_error(SH_ERR_INVALID, "sh_dist_allreduce_bus: buffer too small");
    if (!nvalues) return SH_OK;
    SH_RCCL(r.AllReduce(bus_f64->ptr, bus_f64->ptr, nvalues, ncclFloat64, ncclSum, r.comm, sh::state().stream));
    return SH_OK;
}

int sh_dist_slots(void) { return Rccl::SLOTS; }

int sh_dist_reduce_bus_async(sh_buf* bus_f64, size_t nvalues, int root, sh_buf* bus_f32, int slot) {
    SH_REQUIRE_INIT();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_reduce_bus_async: sh_dist_init not called");
    if (!bus_f64 || bus_f64->bytes < nvalues * 8) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_async: buffer too small");
    if (bus_f32 && bus_f32->bytes < nvalues * 4) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_async: bus_f32 too small");
    if (root < 0 || root >= r.world) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_async: bad root");
    if (slot < 0 || slot >= Rccl::SLOTS) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_async: slot %d outside 0..%d", slot, Rccl::SLOTS - 1);
    r.marked[slot] = false;
    hipStream_t main = sh::state().stream, comm = sh::state().comm_stream;
    SH_HIP(hipEventRecord(r.ev_rendered[slot], main));
    SH_HIP(hipStreamWaitEvent(comm, r.ev_rendered[slot], 0));
    if (nvalues) SH_RCCL(r.Reduce(bus_f64->ptr, bus_f64->ptr, nvalues, ncclFloat64, ncclSum, root, r.comm, comm));
    if (r.rank == root && bus_f32) {
        int rc = sh::bus_finalize_on(comm, (const double*)bus_f64->ptr, nvalues, (float*)bus_f32->ptr);
        if (rc) return rc;
    }
    SH_HIP(hipEventRecord(r.ev_done[slot], comm));
    return SH_OK;
}

// The same in two steps that do not end the run of renders (sh_dist_reduce_bus_async does: SH_REQUIRE_INIT joins the streams
// and folds what is outstanding -- a pipeline drain per batch, 10 us per block at 8 blocks per batch):
//   sh_dist_mark_slot, called once the slot's last render lies two launches back in its bank's run -- every partial bus of the
//     slot has been folded by then (launch n + 2 folds launch n's), on one of the two render streams -- records where both
//     streams stand;
//   sh_dist_reduce_bus_lagged, called a few launches LATER, makes the communication stream wait for those two marks and
//     enqueues the collective.  Later, because the communication stream shares a hardware queue with a render stream (rocprofv3:
//     the rounding kernel ran in stream2's queue): a wait that is not yet satisfied when the queue reaches it holds up the
//     renders behind it -- a mark that fired long ago costs nothing.
// If a fold into the slot's buffer is still owed (the caller did not wait two launches) the reduce IS sh_dist_reduce_bus_async.
int sh_dist_mark_slot(int slot) {
    SH_REQUIRE_INIT_KEEP_PENDING();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_mark_slot: sh_dist_init not called");
    if (slot < 0 || slot >= Rccl::SLOTS) return sh::set_error(SH_ERR_INVALID, "sh_dist_mark_slot: slot %d outside 0..%d", slot, Rccl::SLOTS - 1);
    SH_HIP(hipEventRecord(r.ev_rendered[slot], sh::state().stream));
    SH_HIP(hipEventRecord(r.ev_rendered2[slot], sh::state().stream2));
    r.marked[slot] = true;
    return SH_OK;
}

int sh_dist_reduce_bus_lagged(sh_buf* bus_f64, size_t nvalues, int root, sh_buf* bus_f32, int slot) {
    SH_REQUIRE_INIT_KEEP_PENDING();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_reduce_bus_lagged: sh_dist_init not called");
    if (!bus_f64 || bus_f64->bytes < nvalues * 8) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_lagged: buffer too small");
    if (bus_f32 && bus_f32->bytes < nvalues * 4) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_lagged: bus_f32 too small");
    if (root < 0 || root >= r.world) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_lagged: bad root");
    if (slot < 0 || slot >= Rccl::SLOTS) return sh::set_error(SH_ERR_INVALID, "sh_dist_reduce_bus_lagged: slot %d outside 0..%d", slot, Rccl::SLOTS - 1);
    const bool marked = r.marked[slot];
    r.marked[slot] = false;
    if (!marked || sh::fold_owed_into(bus_f64->ptr, nvalues * 8) || (bus_f32 && sh::fold_owed_into(bus_f32->ptr, nvalues * 4)))
        return sh_dist_reduce_bus_async(bus_f64, nvalues, root, bus_f32, slot);
    hipStream_t comm = sh::state().comm_stream;
    SH_HIP(hipStreamWaitEvent(comm, r.ev_rendered[slot], 0));
    SH_HIP(hipStreamWaitEvent(comm, r.ev_rendered2[slot], 0));
    if (nvalues) SH_RCCL(r.Reduce(bus_f64->ptr, bus_f64->ptr, nvalues, ncclFloat64, ncclSum, root, r.comm, comm));
    if (r.rank == root && bus_f32) {
        int rc = sh::bus_finalize_on(comm, (const double*)bus_f64->ptr, nvalues, (float*)bus_f32->ptr);
        if (rc) return rc;
    }
    SH_HIP(hipEventRecord(r.ev_done[slot], comm));
    return SH_OK;
}

// sh_dist_wait_slot without ending the run of renders: both render streams wait for the slot's collective.
int sh_dist_wait_slot_keep(int slot) {
    SH_REQUIRE_INIT_KEEP_PENDING();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_wait_slot_keep: sh_dist_init not called");
    if (slot < 0 || slot >= Rccl::SLOTS) return sh::set_error(SH_ERR_INVALID, "sh_dist_wait_slot_keep: bad slot");
    SH_HIP(hipStreamWaitEvent(sh::state().stream, r.ev_done[slot], 0));    // no-op until the slot has been used
    SH_HIP(hipStreamWaitEvent(sh::state().stream2, r.ev_done[slot], 0));
    return SH_OK;
}

int sh_dist_wait_slot(int slot) {
    SH_REQUIRE_INIT();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_wait_slot: sh_dist_init not called");
    if (slot < 0 || slot >= Rccl::SLOTS) return sh::set_error(SH_ERR_INVALID, "sh_dist_wait_slot: bad slot");
    SH_HIP(hipStreamWaitEvent(sh::state().stream, r.ev_done[slot], 0));    // no-op until the slot has been used
    return SH_OK;
}

int sh_dist_barrier(void) {
    SH_REQUIRE_INIT();
    Rccl& r = R();
    if (!r.comm) return sh::set_error(SH_ERR_RCCL, "sh_dist_barrier: sh_dist_init not called");
    SH_RCCL(r.AllReduce(r.token, r.token, 1, ncclFloat64, ncclSum, r.comm, sh::state().stream));
    SH_HIP(hipStreamSynchronize(sh::state().stream));
    return SH_OK;
}

}  // extern "C"
