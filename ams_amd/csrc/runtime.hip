// Process-wide launch bookkeeping shared by every kernel launcher:
//   * per-DEVICE one-time function attributes and occupancy figures (hipFuncSetAttribute / hipOccupancy* are per device: a process
//     that holds a server student on one GPU and an edge student on another must set them on both), behind one mutex;
//   * the tuning knobs of the environment, read ONCE (a frame at a time the hot path is ~47 launches: no getenv on it).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>

#include "kernels.hpp"

namespace ams {

namespace {
std::mutex g_mu;
std::map<std::pair<int, const void*>, size_t> g_attr;                       // (device, kernel) -> dynamic LDS limit set so far
std::map<std::tuple<int, const void*, int, size_t>, int> g_occ;             // (device, kernel, threads, lds) -> blocks per CU
std::map<int, int> g_cus;                                                   // device -> compute units
}  // namespace

bool launch_table_needs_attr(int device, const void* fn, size_t lds) {
    std::lock_guard<std::mutex> lk(g_mu);
    size_t& have = g_attr[{device, fn}];
    if (lds <= have) return false;
    have = lds;
    return true;
}

// The grant is recorded only AFTER hipFuncSetAttribute succeeded, and the mutex is held across the call: a second host thread that
// launches the same kernel meanwhile (server and edge students of one process) waits here instead of seeing "already set" and
// launching with more dynamic LDS than the function is allowed yet.
int func_allow_lds(const void* fn, size_t lds) {
    if (lds <= 64 * 1024) return AMS_OK;
    int dev = 0;
    AMS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_attr.find({dev, fn});
    if (it != g_attr.end() && lds <= it->second) return AMS_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(dynamic LDS %zu) -> %s", lds, hipGetErrorString(e));
        return AMS_E_HIP;
    }
    g_attr[{dev, fn}] = lds;
    return AMS_OK;
}

int func_blocks_per_cu(const void* fn, int threads, size_t lds, int* per_cu) {
    int dev = 0;
    AMS_CHECK_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_occ.find({dev, fn, threads, lds});
        if (it != g_occ.end()) { *per_cu = it->second; return AMS_OK; }
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb < 1) nb = 1;
    std::lock_guard<std::mutex> lk(g_mu);
    g_occ[{dev, fn, threads, lds}] = nb;
    *per_cu = nb;
    return AMS_OK;
}

int device_cus(int* cus) {
    int dev = 0;
    AMS_CHECK_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_cus.find(dev);
        if (it != g_cus.end()) { *cus = it->second; return AMS_OK; }
    }
    int n = 0;
    AMS_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_cus[dev] = n;
    *cus = n;
    return AMS_OK;
}

static Knobs read_knobs() {
        Knobs v;
        if (const char* e = getenv("AMS_BLK_TILE")) sscanf(e, "%dx%d", &v.blk_th, &v.blk_tw);
        if (const char* e = getenv("AMS_BLK_HP")) v.blk_hp = atoi(e);
        if (const char* e = getenv("AMS_FB_WALK")) v.fb_walk = atoi(e);
        if (const char* e = getenv("AMS_BLK_TIMED")) v.blk_timed = atoi(e);
        if (const char* e = getenv("AMS_XWR_TIMED")) v.xwr_timed = atoi(e);
        if (const char* e = getenv("AMS_PW_FORCE")) sscanf(e, "%c,%d,%d", &v.pw_force, &v.pw_rm, &v.pw_nt);
        if (const char* e = getenv("AMS_PW_PERCU")) v.pw_percu = atoi(e);
        v.pwx_no_tail = getenv("AMS_PWX_NO_TAIL") != nullptr;
#ifdef AMS_MEASURE
        // measurement build only (make measure -> libams_hip_measure.so): kernels with loads / MFMAs / stores removed — WRONG results by design
        if (const char* e = getenv("AMS_FB_ABL")) v.fb_abl = atoi(e);
        if (const char* e = getenv("AMS_PWH_ABL")) v.pwh_abl = atoi(e);
        if (const char* e = getenv("AMS_XWR_ABL")) v.xwr_abl = atoi(e);
#else
        // the product library has no ablated kernels: the variables are refused loudly, never obeyed
        for (const char* name : {"AMS_FB_ABL", "AMS_PWH_ABL", "AMS_XWR_ABL"})
            if (const char* e = getenv(name))
                if (atoi(e) != 0) fprintf(stderr, "libams_hip: %s=%s ignored — ablated kernels exist only in libams_hip_measure.so (make -C ams_amd/csrc measure)\n", name, e);
#endif
        if (const char* e = getenv("AMS_PWH_VARIANT")) { v.pwh_set = true; sscanf(e, "%d,%d", &v.pwh_nw, &v.pwh_d); }
        if (const char* e = getenv("AMS_PWX_FORCE")) sscanf(e, "%d,%d", &v.pwx_rm, &v.pwx_nt);
        if (const char* e = getenv("AMS_XDS_FORCE")) { v.xds_set = true; sscanf(e, "%d,%d,%d,%d,%d,%d", &v.xds[0], &v.xds[1], &v.xds[2], &v.xds[3], &v.xds[4], &v.xds[5]); }
        if (const char* e = getenv("AMS_XWR_FORCE")) { v.xwr_set = true; sscanf(e, "%d,%d,%d,%d,%d", &v.xwr[0], &v.xwr[1], &v.xwr[2], &v.xwr[3], &v.xwr[4]); }
        if (const char* e = getenv("AMS_WG6_SPLITS")) v.wg6_split_cap = atoi(e);
        v.wg6_eight_waves = getenv("AMS_WG6_EIGHT_WAVES") != nullptr;
        if (const char* e = getenv("AMS_SIDE_CU_MASK")) v.side_cu_mask = (unsigned)strtoul(e, nullptr, 16);
        if (const char* e = getenv("AMS_EVENT_FLAGS")) v.event_flags = (int)strtoul(e, nullptr, 16);
        return v;
}
static Knobs& knobs_storage() {
    static Knobs k = read_knobs();
    return k;
}
const Knobs& knobs() { return knobs_storage(); }

// The events of the engine order streams of one device (fork / join of the weight-gradient stream, of the parts of the two-stream inference
// plan).  Measured on the fine-tune step (~110 records a step, tools/train_ab.sh with AMS_EVENT_FLAGS): hipEventReleaseToDevice changes nothing
// (7.48 ms either way); hipEventDisableSystemFence gives 7.48 -> 7.41 ms, but its contract only covers timing events ("do not require ... to
// synchronize-with the work"), so it is NOT used: plain no-timing events.
int create_sync_event(hipEvent_t* out) {
    const unsigned flags = knobs().event_flags >= 0 ? (unsigned)knobs().event_flags : (unsigned)hipEventDisableTiming;
    AMS_CHECK_HIP(hipEventCreateWithFlags(out, flags));
    return AMS_OK;
}

int create_side_stream(hipStream_t* out) {
    if (knobs().side_cu_mask) {
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = knobs().side_cu_mask;
        AMS_CHECK_HIP(hipExtStreamCreateWithCUMask(out, 8, mask));
        return AMS_OK;
    }
    AMS_CHECK_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return AMS_OK;
}

}  // namespace ams

// tests and tools/ only: read the tuning environment again (the library reads it once, at first use; not thread-safe against running launches)
extern "C" int ams_debug_reload_knobs(void) {
    ams::knobs_storage() = ams::read_knobs();
    return AMS_OK;
}

extern "C" int ams_debug_launch_table_needs_attr(int32_t device, uint64_t kernel_key, size_t lds) {
    return ams::launch_table_needs_attr(device, (const void*)(uintptr_t)kernel_key, lds) ? 1 : 0;
}
