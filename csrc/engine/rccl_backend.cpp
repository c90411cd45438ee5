// Device data plane of the native engine: RCCL over xGMI called directly.
//
//   * communicator bootstrap: rank 0 creates the ncclUniqueId and publishes it in the TCP store,
//     every rank calls ncclCommInitRank (no MPI broadcast of the id as in Horovod);
//   * collectives run on a dedicated high-priority HIP stream.  Ordering with the framework is
//     by events only: an event recorded on the producer (PyTorch) stream gates the collective,
//     a second event recorded after it is what `ddl_rccl_wait_stream` makes the consumer
//     stream wait on — the host never blocks in step() (SURVEY §5.8 "avoid hipStreamSynchronize");
//   * averaging is fused into the reduction (ncclAvg);
//   * grouped launches (`ddl_rccl_group_start/end`) let a bucket of several dtypes go out as
//     one RCCL group.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
int ddl_eng_store_set(const char* key, const void* val, long long n);
long long ddl_eng_store_get(const char* key, int timeout_ms);
const void* ddl_eng_store_data();
void ddl_eng_timeline_mark(const char* name, const char* cat, int phase);
}

namespace {
struct Rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, size = 1, device = 0;
    std::mutex mu;
    std::map<long long, hipEvent_t> done;
    std::vector<hipEvent_t> pool;
    long long next = 1;
    std::string err;
} g;

// an event from the recycle pool (or a new one); nullptr when HIP cannot create one
hipEvent_t get_event() {
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (!g.pool.empty()) {
            hipEvent_t e = g.pool.back();
            g.pool.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
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
bool fail(const char* what, int code) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s failed (%d)", what, code);
    g.err = buf;
    return false;
}
}  // namespace

extern "C" {

const char* ddl_rccl_last_error() { return g.err.c_str(); }

// requires ddl_eng_init() first (the store carries the unique id)
int ddl_rccl_init(int rank, int size, int device, int timeout_ms) {
    if (g.comm) return 0;
    g.rank = rank; g.size = size; g.device = device;
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice", device) ? 0 : -1;
    ncclUniqueId id;
    if (rank == 0) {
        ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) { fail("ncclGetUniqueId", r); return -1; }
        if (ddl_eng_store_set("rccl/unique_id", &id, sizeof(id)) != 0) { g.err = "store set failed"; return -1; }
    } else {
        long long n = ddl_eng_store_get("rccl/unique_id", timeout_ms);
        if (n != (long long)sizeof(id)) { g.err = "unique id not published"; return -1; }
        std::memcpy(&id, ddl_eng_store_data(), sizeof(id));
    }
    ncclResult_t r = ncclCommInitRank(&g.comm, size, id, rank);
    if (r != ncclSuccess) { fail("ncclCommInitRank", r); g.comm = nullptr; return -1; }
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }  // hi = numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, hi) != hipSuccess) { g.err = "stream create failed"; return -1; }
    return 0;
}

void ddl_rccl_shutdown() {
    if (!g.comm) return;
    // teardown: nothing useful can be done with a failure here
    (void)hipStreamSynchronize(g.stream);
    ncclCommDestroy(g.comm);
    (void)hipStreamDestroy(g.stream);
    for (auto& kv : g.done) (void)hipEventDestroy(kv.second);
    for (auto e : g.pool) (void)hipEventDestroy(e);
    g.done.clear(); g.pool.clear();
    g.comm = nullptr; g.stream = nullptr;
}

int ddl_rccl_group_start() { return ncclGroupStart() == ncclSuccess ? 0 : -1; }
int ddl_rccl_group_end() { return ncclGroupEnd() == ncclSuccess ? 0 : -1; }

static long long finish_launch(const char* name) {
    hipEvent_t done = get_event();
    if (!done || hipEventRecord(done, g.stream) != hipSuccess) { g.err = "completion event could not be recorded"; return -1; }
    std::lock_guard<std::mutex> lk(g.mu);
    long long h = g.next++;
    g.done[h] = done;
    if (name) ddl_eng_timeline_mark(name, "RCCL_ENQUEUE", 2);
    return h;
}

static bool gate(void* producer_stream) {
    hipEvent_t ready = get_event();
    if (!ready) return false;
    bool ok = hipEventRecord(ready, (hipStream_t)producer_stream) == hipSuccess &&
              hipStreamWaitEvent(g.stream, ready, 0) == hipSuccess;
    std::lock_guard<std::mutex> lk(g.mu);
    g.pool.push_back(ready);  // safe to recycle: the wait has been enqueued
    return ok;
}

// in-place all-reduce of `count` elements at device pointer `buf`
long long ddl_rccl_allreduce(void* buf, long long count, int dtype, int redop, void* producer_stream, const char* name) {
    if (!g.comm) { g.err = "RCCL backend not initialised"; return -1; }
    if (!gate(producer_stream)) { g.err = "event gate failed"; return -1; }
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, nccl_dtype(dtype), nccl_op(redop), g.comm, g.stream);
    if (r != ncclSuccess) { fail("ncclAllReduce", r); return -1; }
    return finish_launch(name);
}

long long ddl_rccl_broadcast(void* buf, long long count, int dtype, int root, void* producer_stream, const char* name) {
    if (!g.comm) { g.err = "RCCL backend not initialised"; return -1; }
    if (!gate(producer_stream)) { g.err = "event gate failed"; return -1; }
    ncclResult_t r = ncclBroadcast(buf, buf, (size_t)count, nccl_dtype(dtype), root, g.comm, g.stream);
    if (r != ncclSuccess) { fail("ncclBroadcast", r); return -1; }
    return finish_launch(name);
}

long long ddl_rccl_allgather(const void* in, void* out, long long count, int dtype, void* producer_stream, const char* name) {
    if (!g.comm) { g.err = "RCCL backend not initialised"; return -1; }
    if (!gate(producer_stream)) { g.err = "event gate failed"; return -1; }
    ncclResult_t r = ncclAllGather(in, out, (size_t)count, nccl_dtype(dtype), g.comm, g.stream);
    if (r != ncclSuccess) { fail("ncclAllGather", r); return -1; }
    return finish_launch(name);
}

long long ddl_rccl_reduce_scatter(const void* in, void* out, long long recv_count, int dtype, int redop,
                                  void* producer_stream, const char* name) {
    if (!g.comm) { g.err = "RCCL backend not initialised"; return -1; }
    if (!gate(producer_stream)) { g.err = "event gate failed"; return -1; }
    ncclResult_t r = ncclReduceScatter(in, out, (size_t)recv_count, nccl_dtype(dtype), nccl_op(redop), g.comm, g.stream);
    if (r != ncclSuccess) { fail("ncclReduceScatter", r); return -1; }
    return finish_launch(name);
}

// make `consumer_stream` wait for the collective; the host does not block.  Releases the handle.
int ddl_rccl_wait_stream(long long handle, void* consumer_stream) {
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = g.done.find(handle);
        if (it == g.done.end()) return -1;
        e = it->second;
        g.done.erase(it);
    }
    int rc = hipStreamWaitEvent((hipStream_t)consumer_stream, e, 0) == hipSuccess ? 0 : -1;
    std::lock_guard<std::mutex> lk(g.mu);
    g.pool.push_back(e);
    return rc;
}

// 1 = complete, 0 = still running, -1 = unknown handle
int ddl_rccl_poll(long long handle) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.done.find(handle);
    if (it == g.done.end()) return -1;
    return hipEventQuery(it->second) == hipSuccess ? 1 : 0;
}

// host-blocking wait (tests, shutdown paths)
int ddl_rccl_wait_host(long long handle) {
    hipEvent_t e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = g.done.find(handle);
        if (it == g.done.end()) return -1;
        e = it->second;
        g.done.erase(it);
    }
    int rc = hipEventSynchronize(e) == hipSuccess ? 0 : -1;
    std::lock_guard<std::mutex> lk(g.mu);
    g.pool.push_back(e);
    return rc;
}

}  // extern "C"
