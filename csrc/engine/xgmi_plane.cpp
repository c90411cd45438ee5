#include "xgmi_plane.h"

#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" {
int ddl_eng_store_set(const char* key, const void* val, long long n);
long long ddl_eng_store_get(const char* key, int timeout_ms);
const void* ddl_eng_store_data();
}

namespace ddl {
namespace xgmi {
namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t FLAG_PAGE_BYTES = 4096;

// signature of ddl_xgmi_allreduce (csrc/kernels/xgmi_allreduce.hip)
typedef int (*kernel_fn_t)(void* const* bufs, void* const* flags, void* const* outs, long long count, int rank, int nranks,
                           int is_bf16, int one_shot, unsigned epoch, float scale, long long spin_limit, void* timeout_word,
                           int blocks, int simulate, hipStream_t stream);

struct Wire {              // what travels through the store for one exported allocation
    hipIpcMemHandle_t handle;
    uint64_t offset, bytes;
    int32_t pid, device;
};

struct Region {
    char* local = nullptr;
    size_t bytes = 0;
    char* peer[MAX_RANKS] = {nullptr};   // peer[r] = rank r's counterpart of `local` mapped into this process
    void* opened[MAX_RANKS] = {nullptr}; // bases returned by hipIpcOpenMemHandle (to close at shutdown)
};

struct State {
    bool on = false;
    int rank = 0, size = 1, device = 0, blocks = 16;
    long long spin_limit = 0;
    kernel_fn_t kernel = nullptr;
    void* flag_page = nullptr;            // this rank's page (fine-grained device memory)
    void* flags[MAX_RANKS] = {nullptr};   // every rank's page as seen from here
    void* flag_opened[MAX_RANKS] = {nullptr};
    unsigned* timeout_host = nullptr;     // pinned, mapped: the kernel raises it, the watchdog reads it
    void* timeout_dev = nullptr;
    void* scratch = nullptr;              // one-shot output
    size_t scratch_bytes = 0;
    unsigned epoch = 0;
    std::vector<Region> regions;
    std::mutex mu;
} S;

bool fail(std::string* err, const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    *err = buf;
    return false;
}

// export `ptr` (anywhere inside a hipMalloc-family allocation) under `key`
bool publish(const std::string& key, void* ptr, size_t bytes, std::string* err) {
    void* base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess) return fail(err, "hipMemGetAddressRange", e);
    Wire w;
    std::memset(&w, 0, sizeof(w));
    e = hipIpcGetMemHandle(&w.handle, base);
    if (e != hipSuccess) return fail(err, "hipIpcGetMemHandle", e);
    w.offset = (uint64_t)((char*)ptr - (char*)base);
    w.bytes = bytes;
    w.pid = (int32_t)getpid();
    w.device = S.device;
    if (ddl_eng_store_set(key.c_str(), &w, sizeof(w)) != 0) { *err = "store set failed for " + key; return false; }
    return true;
}

bool open_peer(const std::string& key, size_t expect_bytes, void** opened, char** mapped, std::string* err) {
    long long n = ddl_eng_store_get(key.c_str(), 120000);
    if (n != (long long)sizeof(Wire)) { *err = "peer did not publish " + key; return false; }
    Wire w;
    std::memcpy(&w, ddl_eng_store_data(), sizeof(w));
    if (w.bytes != expect_bytes) { *err = "peer region size differs for " + key; return false; }
    void* base = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&base, w.handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(err, ("hipIpcOpenMemHandle(" + key + ")").c_str(), e);
    *opened = base;
    *mapped = (char*)base + w.offset;
    return true;
}

}  // namespace

bool active() { return S.on; }

bool init(int rank, int size, int device, void* kernel_fn, int blocks, long long spin_limit, std::string* err) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (S.on) return true;
    if (size > MAX_RANKS) { *err = "the direct xGMI all-reduce spans at most 8 ranks"; return false; }
    if (!kernel_fn) { *err = "kernel entry point missing"; return false; }
    S.rank = rank; S.size = size; S.device = device;
    S.blocks = blocks > 0 ? blocks : 16;
    S.spin_limit = spin_limit > 0 ? spin_limit : 20000000LL;
    S.kernel = (kernel_fn_t)kernel_fn;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(err, "hipSetDevice", e);
    // Flags are written by PEER devices while this device polls them inside a running kernel:
    // fine-grained memory, so neither side's L2 may hold them back.
    e = hipExtMallocWithFlags(&S.flag_page, FLAG_PAGE_BYTES, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail(err, "hipExtMallocWithFlags(fine-grained flag page)", e);
    e = hipMemset(S.flag_page, 0, FLAG_PAGE_BYTES);
    if (e != hipSuccess) return fail(err, "hipMemset(flag page)", e);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(err, "hipDeviceSynchronize", e);
    e = hipHostMalloc((void**)&S.timeout_host, 64, hipHostMallocMapped);
    if (e != hipSuccess) return fail(err, "hipHostMalloc(timeout word)", e);
    *S.timeout_host = 0;
    e = hipHostGetDevicePointer(&S.timeout_dev, S.timeout_host, 0);
    if (e != hipSuccess) return fail(err, "hipHostGetDevicePointer", e);
    S.flags[rank] = S.flag_page;
    if (size > 1) {
        if (!publish("xgmi/flags/" + std::to_string(rank), S.flag_page, FLAG_PAGE_BYTES, err)) return false;
        for (int r = 0; r < size; ++r) {
            if (r == rank) continue;
            char* m = nullptr;
            if (!open_peer("xgmi/flags/" + std::to_string(r), FLAG_PAGE_BYTES, &S.flag_opened[r], &m, err)) return false;
            S.flags[r] = m;
        }
    }
    S.on = true;
    return true;
}

int register_region(void* ptr, size_t bytes, std::string* err) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.on) { *err = "xGMI plane not initialised"; return -1; }
    if (!ptr || bytes == 0 || ((uintptr_t)ptr & 15)) { *err = "region must be 16-byte aligned and non-empty"; return -1; }
    Region rg;
    rg.local = (char*)ptr; rg.bytes = bytes;
    rg.peer[S.rank] = (char*)ptr;
    const int id = (int)S.regions.size();
    if (S.size > 1) {
        const std::string stem = "xgmi/region/" + std::to_string(id) + "/";
        if (!publish(stem + std::to_string(S.rank), ptr, bytes, err)) return -1;
        for (int r = 0; r < S.size; ++r) {
            if (r == S.rank) continue;
            if (!open_peer(stem + std::to_string(r), bytes, &rg.opened[r], &rg.peer[r], err)) return -1;
        }
    }
    S.regions.push_back(rg);
    return id;
}

static const Region* find_region(const void* buf, size_t bytes) {
    for (const Region& r : S.regions)
        if ((const char*)buf >= r.local && (const char*)buf + bytes <= r.local + r.bytes) return &r;
    return nullptr;
}

bool usable(const void* buf, size_t bytes, long long count, int nranks) {
    if (!S.on || nranks != S.size || count <= 0 || (count % (8LL * nranks)) != 0 || ((uintptr_t)buf & 15)) return false;
    std::lock_guard<std::mutex> lk(S.mu);
    return find_region(buf, bytes) != nullptr;
}

bool launch(void* buf, long long count, bool bf16, bool one_shot, float scale, hipStream_t stream, std::string* err) {
    std::lock_guard<std::mutex> lk(S.mu);
    const size_t bytes = (size_t)count * (bf16 ? 2 : 4);
    const Region* rg = find_region(buf, bytes);
    if (!rg) { *err = "buffer is not inside a registered xGMI region"; return false; }
    const size_t off = (size_t)((char*)buf - rg->local);
    void* bufs[MAX_RANKS] = {nullptr};
    void* outs[MAX_RANKS] = {nullptr};
    for (int r = 0; r < S.size; ++r) bufs[r] = rg->peer[r] + off;
    if (one_shot) {
        if (S.scratch_bytes < bytes) {
            if (S.scratch) (void)hipFree(S.scratch);
            S.scratch = nullptr; S.scratch_bytes = 0;
            hipError_t e = hipMalloc(&S.scratch, bytes);
            if (e != hipSuccess) return fail(err, "hipMalloc(one-shot scratch)", e);
            S.scratch_bytes = bytes;
        }
        outs[S.rank] = S.scratch;
    }
    const unsigned epoch = ++S.epoch;
    int rc = S.kernel(bufs, S.flags, one_shot ? outs : nullptr, count, S.rank, S.size, bf16 ? 1 : 0, one_shot ? 1 : 0, epoch,
                      scale, S.spin_limit, S.timeout_dev, S.blocks, 0, stream);
    if (rc != 0) {
        char b[96];
        snprintf(b, sizeof(b), "xgmi_allreduce launcher returned %d", rc);
        *err = b;
        return false;
    }
    if (one_shot) {
        hipError_t e = hipMemcpyAsync(buf, S.scratch, bytes, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return fail(err, "hipMemcpyAsync(one-shot result)", e);
    }
    return true;
}

int timeout_flag() { return S.timeout_host && *(volatile unsigned*)S.timeout_host != 0 ? 1 : 0; }

// host-side abort: spinning kernels see the word within 4096 polls and return (first diagnosis wins)
void raise_abort() {
    if (S.timeout_host && *(volatile unsigned*)S.timeout_host == 0) *(volatile unsigned*)S.timeout_host = 0x400u;
}

// what the flag page looked like when a kernel gave up: which peer never announced / finished
std::string describe() {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.on) return "xgmi plane off";
    // The null stream may sit behind the very collective that hangs (torch's default stream IS the
    // null stream and waits on `done` events): a synchronous hipMemcpy here would never return and
    // the watchdog that calls this would never reach fail_all().  Snapshot through a private
    // non-blocking stream into pinned memory and wait for it for at most 2 s.
    unsigned words[32] = {0};
    char buf[512];
    {
        hipStream_t st = nullptr;
        unsigned* pinned = nullptr;
        hipEvent_t ev = nullptr;
        bool got = false;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
            hipHostMalloc((void**)&pinned, sizeof(words), hipHostMallocDefault) == hipSuccess &&
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess &&
            hipMemcpyAsync(pinned, S.flag_page, sizeof(words), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipEventRecord(ev, st) == hipSuccess) {
            for (int i = 0; i < 2000 && !got; ++i) {
                if (hipEventQuery(ev) == hipSuccess) got = true;
                else usleep(1000);
            }
            if (got) std::memcpy(words, pinned, sizeof(words));
        }
        // a copy that never finished keeps its stream and buffer (leaked on purpose: the process is
        // about to fail anyway and freeing them could block)
        if (got) { (void)hipEventDestroy(ev); (void)hipHostFree(pinned); (void)hipStreamDestroy(st); }
        if (!got) return "flag page unreadable within 2 s (device busy)";
    }
    int n = snprintf(buf, sizeof(buf), "rank %d epoch %u timeout_word %u ready[", S.rank, S.epoch, S.timeout_host ? *S.timeout_host : 0u);
    for (int i = 0; i < S.size; ++i) n += snprintf(buf + n, sizeof(buf) - n, "%u ", words[i]);
    n += snprintf(buf + n, sizeof(buf) - n, "] done[");
    for (int i = 0; i < S.size; ++i) n += snprintf(buf + n, sizeof(buf) - n, "%u ", words[8 + i]);
    snprintf(buf + n, sizeof(buf) - n, "] ticket %u", words[16]);
    return buf;
}

void shutdown() {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.on) return;
    // teardown is best effort: failures here leave nothing to act on
    for (Region& r : S.regions)
        for (int i = 0; i < MAX_RANKS; ++i)
            if (r.opened[i]) (void)hipIpcCloseMemHandle(r.opened[i]);
    S.regions.clear();
    for (int i = 0; i < MAX_RANKS; ++i) {
        if (S.flag_opened[i]) (void)hipIpcCloseMemHandle(S.flag_opened[i]);
        S.flag_opened[i] = nullptr; S.flags[i] = nullptr;
    }
    if (S.flag_page) (void)hipFree(S.flag_page);
    if (S.timeout_host) (void)hipHostFree(S.timeout_host);
    if (S.scratch) (void)hipFree(S.scratch);
    S.flag_page = nullptr; S.timeout_host = nullptr; S.timeout_dev = nullptr; S.scratch = nullptr; S.scratch_bytes = 0;
    S.on = false;
}

}  // namespace xgmi
}  // namespace ddl
