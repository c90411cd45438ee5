// Device data plane of the native engine: RCCL over xGMI called directly, plus the launch side of
// the direct peer-access all-reduce.  Registered with the engine core as a
// `DevicePlane` (device_plane.h): every collective is LAUNCHED BY THE ENGINE'S COMMUNICATION THREAD
// on the plane's own high-priority HIP stream.
//
//   * communicator bootstrap: rank 0 creates the ncclUniqueId and publishes it in the TCP store,
//     every rank calls ncclCommInitRank (no MPI broadcast of the id as in Horovod);
//   * ordering with the framework is by events only: an event recorded on the producer (PyTorch)
//     stream by the submitting thread gates the collective, a `done` event recorded after it is
//     what the consumer stream waits on - the host never blocks in step();
//   * `start`/`done` are timing events: the watchdog turns them into per-bucket device times
//     (metrics.jsonl "bucket_ms") and into the collective timeout (§5.3);
//   * averaging is fused into the reduction (ncclAvg / the kernel's 1/n scale);
//   * HVD_ALLREDUCE_ALGO=twoshot|oneshot|auto routes all-reduces of REGISTERED memory (the gradient
//     arenas) to `xgmi_allreduce_kernel` (csrc/kernels/xgmi_allreduce.hip) over hipIpc peer
//     mappings (xgmi_plane.cpp); anything else goes through RCCL.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "device_plane.h"
#include "xgmi_plane.h"

extern "C" {
int ddl_eng_store_set(const char* key, const void* val, long long n);
long long ddl_eng_store_get(const char* key, int timeout_ms);
const void* ddl_eng_store_data();
}

namespace {
struct Rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, size = 1, device = 0;
    std::mutex mu;
    std::vector<hipEvent_t> plain, timed;  // recycle pools: readiness events / ticket events
    std::string err;
    std::atomic<long long> launched{0};    // collectives issued through ncclAllReduce/...
    std::atomic<long long> launched_xgmi{0};
    std::atomic<bool> aborted{false};
    bool no_comm = false;   // DDL_RCCL_SKIP_COMM=1: stream + events + xGMI plane only (several ranks on ONE GPU,
                            // which RCCL refuses: "duplicate GPU"); RCCL collectives then fail loudly
} g;

hipEvent_t get_event(bool timing) {
    {
        std::lock_guard<std::mutex> lk(g.mu);
        auto& pool = timing ? g.timed : g.plain;
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
void put_event(hipEvent_t e, bool timing) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(g.mu);
    (timing ? g.timed : g.plain).push_back(e);
}

ncclDataType_t nccl_dtype(int dt) {
    switch (dt) {
        case 0: return ncclFloat32;
        case 1: return ncclFloat64;
        case 2: return ncclInt64;
        case 3: return ncclBfloat16;
        case 4: return ncclInt32;
        case 5: return ncclUint8;
        default: return ncclFloat16;
    }
}
ncclRedOp_t nccl_op(int op) {
    switch (op) {
        case 1: return ncclAvg;
        case 2: return ncclMin;
        case 3: return ncclMax;
        default: return ncclSum;
    }
}
std::string what_failed(const char* what, int code) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s failed (%d)", what, code);
    return buf;
}

// ---- DevicePlane implementation ---------------------------------------------------------------
bool plane_bind_thread() { return hipSetDevice(g.device) == hipSuccess; }

void* plane_record_ready(void* producer_stream) {
    hipEvent_t ready = get_event(false);
    if (!ready) return nullptr;
    if (hipEventRecord(ready, (hipStream_t)producer_stream) != hipSuccess) { put_event(ready, false); return nullptr; }
    return ready;
}

bool plane_launch(const ddl::DeviceOp& op, ddl::DeviceTicket* t, std::string* err) {
    if ((!g.comm && !g.no_comm) || g.aborted.load()) { *err = "RCCL communicator is not available"; return false; }
    hipEvent_t ready = (hipEvent_t)op.ready;
    const bool gated = hipStreamWaitEvent(g.stream, ready, 0) == hipSuccess;
    put_event(ready, false);  // safe to recycle: the wait has been enqueued
    if (!gated) { *err = "event gate failed"; return false; }
    hipEvent_t start = get_event(true), done = get_event(true);
    if (!start || !done || hipEventRecord(start, g.stream) != hipSuccess) { *err = "start event could not be recorded"; return false; }
    t->start = start; t->done = done; t->algo = ddl::ALGO_RCCL;
    const size_t es = op.dtype == 0 || op.dtype == 4 ? 4 : (op.dtype == 1 || op.dtype == 2 ? 8 : (op.dtype == 5 ? 1 : 2));
    bool direct = false;
    if (op.kind == 0 && op.algo != ddl::ALGO_RCCL && (op.dtype == 0 || op.dtype == 3) && (op.redop == 0 || op.redop == 1) &&
        g.size > 1 && ddl::xgmi::usable(op.buf, (size_t)op.count * es, op.count, g.size)) {
        std::string xerr;
        if (!ddl::xgmi::launch(op.buf, op.count, op.dtype == 3, op.algo == ddl::ALGO_ONESHOT,
                               op.redop == 1 ? 1.0f / (float)g.size : 1.0f, g.stream, &xerr)) { *err = xerr; return false; }
        direct = true;
        t->algo = op.algo;
        ++g.launched_xgmi;
    }
    if (!direct && !g.comm) {
        *err = "no RCCL communicator (DDL_RCCL_SKIP_COMM=1): only xGMI all-reduces of registered memory can run on the device";
        return false;
    }
    if (!direct) {
        ncclResult_t r = ncclSuccess;
        if (op.kind == 0) r = ncclAllReduce(op.buf, op.buf, (size_t)op.count, nccl_dtype(op.dtype), nccl_op(op.redop), g.comm, g.stream);
        else if (op.kind == 1) r = ncclBroadcast(op.buf, op.buf, (size_t)op.count, nccl_dtype(op.dtype), op.root, g.comm, g.stream);
        else if (op.kind == 2) r = ncclAllGather(op.buf, op.out, (size_t)op.count, nccl_dtype(op.dtype), g.comm, g.stream);
        else { *err = "unknown device collective"; return false; }
        if (r != ncclSuccess) { *err = what_failed(op.kind == 0 ? "ncclAllReduce" : op.kind == 1 ? "ncclBroadcast" : "ncclAllGather", r); return false; }
        ++g.launched;
    }
    if (hipEventRecord(done, g.stream) != hipSuccess) { *err = "completion event could not be recorded"; return false; }
    return true;
}

int plane_query(const ddl::DeviceTicket& t) {
    if (!t.done) return -1;
    hipError_t e = hipEventQuery((hipEvent_t)t.done);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return -1;
}

float plane_elapsed_ms(const ddl::DeviceTicket& t) {
    float ms = 0.f;
    if (!t.start || !t.done || hipEventElapsedTime(&ms, (hipEvent_t)t.start, (hipEvent_t)t.done) != hipSuccess) return -1.f;
    return ms;
}

bool plane_wait_on(const ddl::DeviceTicket& t, void* consumer_stream) {
    return t.done && hipStreamWaitEvent((hipStream_t)consumer_stream, (hipEvent_t)t.done, 0) == hipSuccess;
}

void plane_recycle(ddl::DeviceTicket* t) {
    put_event((hipEvent_t)t->start, true);
    put_event((hipEvent_t)t->done, true);
    t->start = t->done = nullptr;
}

void plane_abort(const char* why) {
    if (g.aborted.exchange(true)) return;
    fprintf(stderr, "[ddl rccl rank %d] aborting the communicator: %s\n", g.rank, why ? why : "");
    // 1. release what can hold the GPU: spinning xGMI kernels see the raised word within 4096 polls,
    //    ncclCommAbort stops the communicator's kernels - so that a wedged collective frees the device
    //    and any host thread blocked in a stream/device synchronisation returns
    const bool xgmi_on = ddl::xgmi::active();
    if (xgmi_on) ddl::xgmi::raise_abort();
    if (g.comm) (void)ncclCommAbort(g.comm);
    g.comm = nullptr;
    // 2. only then the diagnostic snapshot (bounded wait on a private stream: never the null stream,
    //    which may itself sit behind the dead collective's `done` event)
    if (xgmi_on) fprintf(stderr, "[ddl xgmi] %s\n", ddl::xgmi::describe().c_str());
}

int plane_kernel_timeout() { return ddl::xgmi::timeout_flag(); }

const ddl::DevicePlane kPlane = {plane_bind_thread, plane_record_ready, plane_launch, plane_query, plane_elapsed_ms,
                                 plane_wait_on, plane_recycle, plane_abort, plane_kernel_timeout};
}  // namespace

extern "C" {

const char* ddl_rccl_last_error() { return g.err.c_str(); }

// requires ddl_eng_init() first (the store carries the unique id); attaches the plane to the engine
int ddl_rccl_init(int rank, int size, int device, int timeout_ms) {
    if (g.comm || g.stream) return 0;
    g.rank = rank; g.size = size; g.device = device;
    g.aborted.store(false);
    if (hipSetDevice(device) != hipSuccess) { g.err = what_failed("hipSetDevice", device); return -1; }
    const char* skip = getenv("DDL_RCCL_SKIP_COMM");
    g.no_comm = skip && skip[0] == '1';
    ncclUniqueId id;
    if (g.no_comm) {
        // no communicator: nothing to exchange
    } else if (rank == 0) {
        ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) { g.err = what_failed("ncclGetUniqueId", r); return -1; }
        if (ddl_eng_store_set("rccl/unique_id", &id, sizeof(id)) != 0) { g.err = "store set failed"; return -1; }
    } else {
        long long n = ddl_eng_store_get("rccl/unique_id", timeout_ms);
        if (n != (long long)sizeof(id)) { g.err = "unique id not published"; return -1; }
        std::memcpy(&id, ddl_eng_store_data(), sizeof(id));
    }
    if (!g.no_comm) {
        ncclResult_t r = ncclCommInitRank(&g.comm, size, id, rank);
        if (r != ncclSuccess) { g.err = what_failed("ncclCommInitRank", r); g.comm = nullptr; return -1; }
    }
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }  // hi = numerically lowest = highest priority
    // DDL_COMM_STREAM_PRIORITY=high (default) | normal | low.  Several ranks sharing ONE GPU (tests) must
    // not run their spinning xGMI kernels above a peer process's compute queues: a resident
    // high-priority kernel that waits for a peer starves that peer's lower-priority queues.
    int prio = hi;
    if (const char* pr = getenv("DDL_COMM_STREAM_PRIORITY")) {
        if (!strcmp(pr, "normal")) prio = 0;
        else if (!strcmp(pr, "low")) prio = lo;
    }
    if (hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, prio) != hipSuccess) { g.err = "stream create failed"; return -1; }
    ddl::set_device_plane(&kPlane);
    return 0;
}

void ddl_rccl_shutdown() {
    ddl::set_device_plane(nullptr);
    ddl::xgmi::shutdown();
    if (g.stream) (void)hipStreamSynchronize(g.stream);  // teardown: nothing useful can be done with a failure here
    if (g.comm) ncclCommDestroy(g.comm);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    for (auto e : g.plain) (void)hipEventDestroy(e);
    for (auto e : g.timed) (void)hipEventDestroy(e);
    g.plain.clear(); g.timed.clear();
    g.comm = nullptr; g.stream = nullptr;
}

// raw handle of the plane's stream (tests / profiling: "is the collective on the engine's stream?")
void* ddl_rccl_stream() { return (void*)g.stream; }
long long ddl_rccl_launch_count() { return g.launched.load(); }
long long ddl_xgmi_launch_count() { return g.launched_xgmi.load(); }

// ---- xGMI direct all-reduce: set-up calls (collective: every rank, same order) -----------------
// `kernel_fn` = address of ddl_xgmi_allreduce in _ddl_kernels.so (the engine library carries no
// device code of its own)
int ddl_rccl_has_comm() { return g.comm ? 1 : 0; }

int ddl_xgmi_init(void* kernel_fn, int blocks, long long spin_limit) {
    if (!g.stream) { g.err = "device plane not initialised"; return -1; }
    std::string err;
    if (!ddl::xgmi::init(g.rank, g.size, g.device, kernel_fn, blocks, spin_limit, &err)) { g.err = err; return -1; }
    return 0;
}
int ddl_xgmi_register(void* ptr, long long bytes) {
    std::string err;
    int id = ddl::xgmi::register_region(ptr, (size_t)bytes, &err);
    if (id < 0) g.err = err;
    return id;
}

}  // extern "C"
