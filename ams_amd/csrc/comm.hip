// RCCL communicator owned by the library: the data-parallel fine-tune step issues its all-reduces (SyncBN sums, loss normaliser,
// the flat gradient arena) straight onto the launch stream, with no host round trip between them (the reference runs
// one process per GPU, run.py:28).  librccl is resolved at run time with dlopen/dlsym: the library has no link-time dependency on
// it (it loads and passes its CPU tests on a box without RCCL), and inside a PyTorch-ROCm process the copy PyTorch already loaded
// (same soname) is the one that serves both.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "kernels.hpp"

struct ams_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int64_t calls = 0, bytes = 0;        // what the steps exchanged so far (tests)
    bool aborted = false;
    // diagnostic timing (ams_comm_set_timing): a HIP event pair around every collective on its stream; read back by ams_comm_timing_read
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;      // at most kMaxSpans: timing left on for a long run stops recording, it does not grow
    std::vector<hipEvent_t> pool;
    int64_t event_failures = 0;          // hipEventCreate failed while timing: ams_comm_timing_read reports it instead of an undercount
    static constexpr size_t kMaxSpans = 4096;      // ~37 steps of 110 collectives
    ~ams_comm() {
        for (auto& sp : spans) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

namespace ams {

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* error = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {        // a copy that is already mapped (PyTorch's) wins over a second instance
            r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.handle) { r.error = "librccl.so.1 not found (dlopen)"; return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");               // optional: error path only
        r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");               // optional: ams_comm_stats
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.handle, "ncclCommUserRank");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) r.error = "librccl lacks a required symbol";
    });
    return r;
}

int fail(const char* what, ncclResult_t rc) {
    Rccl& r = rccl();
    set_error("%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return AMS_E_HIP;
}
}  // namespace

int comm_allreduce(ams_comm* c, void* p, size_t n, int dtype, hipStream_t st) {
    if (c && c->aborted) { set_error("comm: the communicator was aborted after an earlier RCCL error"); return AMS_E_STATE; }
    if (!c || !c->comm) return AMS_OK;             // world 1 without a communicator: the sum over one rank is the value itself
    Rccl& r = rccl();
    const ncclDataType_t dt = dtype == AMS_DT_F64 ? ncclFloat64 : ncclFloat32;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && c->spans.size() < ams_comm::kMaxSpans) {
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) { e = nullptr; c->event_failures += 1; }
            return e;
        };
        e0 = take(); e1 = take();
        if (e0 && e1) (void)hipEventRecord(e0, st);
        else { if (e0) c->pool.push_back(e0); if (e1) c->pool.push_back(e1); e0 = e1 = nullptr; }
    }
    const ncclResult_t rc = r.AllReduce(p, p, n, dt, ncclSum, c->comm, st);
    if (c->timing && e0 && e1) { (void)hipEventRecord(e1, st); c->spans.emplace_back(e0, e1); }
    if (rc != ncclSuccess) {
        // a rank that fails between two collectives of a step leaves its peers waiting inside theirs: abort the communicator so that
        // they return with an error instead of hanging; this handle is dead from here on
        if (r.CommAbort) (void)r.CommAbort(c->comm);
        c->comm = nullptr;
        c->aborted = true;
        return fail("ncclAllReduce (communicator aborted)", rc);
    }
    c->calls += 1;
    c->bytes += (int64_t)n * (dtype == AMS_DT_F64 ? 8 : 4);
    return AMS_OK;
}

}  // namespace ams

using namespace ams;

extern "C" {

int ams_comm_unique_id(uint8_t* id_out, size_t cap) {
    AMS_REQUIRE(id_out && cap >= NCCL_UNIQUE_ID_BYTES, "comm_unique_id: buffer must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
    Rccl& r = rccl();
    if (r.error) { set_error("comm: %s", r.error); return AMS_E_STATE; }
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return AMS_OK;
}

int ams_comm_create(const uint8_t* id_bytes, size_t id_len, int32_t rank, int32_t world, ams_comm** out) {
    AMS_REQUIRE(out && world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d of %d", rank, world);
    // every argument is checked before anything is allocated
    AMS_REQUIRE(world == 1 || id_bytes, "comm_create: a communicator of %d ranks needs the unique id", world);
    AMS_REQUIRE(!id_bytes || id_len >= NCCL_UNIQUE_ID_BYTES, "comm_create: the unique id is %d bytes", NCCL_UNIQUE_ID_BYTES);
    ams_comm* c = new ams_comm();
    c->rank = rank; c->world = world;
    if (world > 1 || id_bytes) {                   // world 1 with an id: a real single-rank communicator (what the GPU test exercises)
        Rccl& r = rccl();
        if (r.error) { set_error("comm: %s", r.error); delete c; return AMS_E_STATE; }
        ncclUniqueId id;
        memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
        const ncclResult_t rc = r.CommInitRank(&c->comm, world, id, rank);      // binds to the calling thread's current device
        if (rc != ncclSuccess) { delete c; return fail("ncclCommInitRank", rc); }
    }
    *out = c;
    return AMS_OK;
}

void ams_comm_destroy(ams_comm* c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
}

int ams_comm_stats(const ams_comm* c, int32_t* rank, int32_t* world, int64_t* calls, int64_t* bytes) {
    AMS_REQUIRE(c, "comm_stats: null communicator");
    // rank / world as RCCL itself reports them for this communicator (what was asked for when there is none: world 1)
    int r_seen = c->rank, w_seen = c->world;
    if (c->comm) {
        Rccl& r = rccl();
        if (r.CommCount && r.CommUserRank) {
            ncclResult_t rc = r.CommCount(c->comm, &w_seen);
            if (rc == ncclSuccess) rc = r.CommUserRank(c->comm, &r_seen);
            if (rc != ncclSuccess) return fail("ncclCommCount / ncclCommUserRank", rc);
        }
    }
    if (rank) *rank = r_seen;
    if (world) *world = w_seen;
    if (calls) *calls = c->calls;
    if (bytes) *bytes = c->bytes;
    return AMS_OK;
}

int ams_comm_set_timing(ams_comm* c, int32_t enable) {
    AMS_REQUIRE(c, "comm_set_timing: null communicator");
    AMS_CHECK_HIP(hipDeviceSynchronize());
    for (auto& sp : c->spans) { c->pool.push_back(sp.first); c->pool.push_back(sp.second); }
    c->spans.clear();
    c->event_failures = 0;
    c->timing = enable != 0;
    return AMS_OK;
}

int ams_comm_timing_read(ams_comm* c, double* total_ms, double* max_ms, int64_t* spans) {
    AMS_REQUIRE(c && total_ms && max_ms && spans, "comm_timing_read: null pointer");
    AMS_CHECK_HIP(hipDeviceSynchronize());
    double tot = 0.0, mx = 0.0;
    for (auto& sp : c->spans) {
        float ms = 0.f;
        AMS_CHECK_HIP(hipEventElapsedTime(&ms, sp.first, sp.second));
        tot += ms;
        if (ms > mx) mx = ms;
    }
    *total_ms = tot; *max_ms = mx; *spans = (int64_t)c->spans.size();
    if (c->event_failures) { set_error("comm_timing_read: hipEventCreate failed for %lld collective(s): the totals undercount", (long long)c->event_failures); return AMS_E_HIP; }
    return AMS_OK;
}

/* sum `count` elements (AMS_DT_F32 | AMS_DT_F64) in place across the communicator, on `stream` */
int ams_comm_allreduce(ams_comm* c, void* buf_dev, size_t count, int32_t dtype, void* stream) {
    AMS_REQUIRE(c && buf_dev && (dtype == AMS_DT_F32 || dtype == AMS_DT_F64), "comm_allreduce: bad argument");
    return comm_allreduce(c, buf_dev, count, dtype, (hipStream_t)stream);
}

}  // extern "C"
