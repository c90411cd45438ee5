// Native gradient-exchange engine.
//
// What Horovod 0.13's C++ core does for the reference — a background thread per process, a
// control plane, fused collectives, a timeline, stall checks — re-designed for one MI355X node:
//   * rendezvous and control over a TCP key-value store (store.cpp), no MPI;
//   * a background communication thread draining a submit queue; callers get integer handles
//     (allreduce_async_/synchronize/poll semantics), the Python GIL is never held while data
//     moves;
//   * no per-cycle negotiation: data-parallel ranks submit the same collectives in the same
//     order (static bucket plan); every message carries an (op, dtype, count, name-hash) header
//     that is verified on receipt, which is the consistency check Horovod's coordinator did;
//   * host data plane: ring reduce-scatter + all-gather over TCP between neighbours (the
//     GPU-less plumbing configuration and the tests); device data plane: RCCL over xGMI, called
//     directly (rccl_backend.cpp) on a dedicated HIP stream with event hand-offs;
//   * chrome-trace timeline (HVD_TIMELINE) written by the engine, stall warnings after
//     `stall_warning_s`, hard failure of every pending handle after `timeout_s` (a dead peer
//     surfaces as an error in synchronize(), not a hang);
//   * the static bucket planner (bucket_plan) shared with the Python planner.
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device_plane.h"
#include "store.h"

namespace ddl {

enum DType : int { F32 = 0, F64 = 1, I64 = 2, BF16 = 3, I32 = 4, U8 = 5, F16 = 6 };
enum RedOp : int { SUM = 0, AVG = 1, MIN = 2, MAX = 3 };
enum OpKind : int { ALLREDUCE = 0, BROADCAST = 1, ALLGATHER = 2, BARRIER = 3 };
enum Status : int { PENDING = 0, RUNNING = 1, DONE = 2, FAILED = 3 };

static size_t dtype_size(int dt) {
    switch (dt) {
        case F32: case I32: return 4;
        case F64: case I64: return 8;
        case BF16: case F16: return 2;
        default: return 1;
    }
}

static inline float bf16_to_f(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// IEEE binary16 <-> float (round to nearest even); the host ring reduces fp16 in fp32 like bf16
static inline float f16_to_f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {  // subnormal: normalise
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f_to_f16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);           // NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // overflow (incl. inf) -> inf
    if (a < 0x33000001u) return sign;                                 // underflow -> signed zero
    int exp = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;
    int shift = exp < -14 ? 13 + (-14 - exp) : 13;                    // subnormal halves lose more bits
    uint32_t half = exp < -14 ? 0u : (uint32_t)(exp + 15) << 10;
    uint32_t m = man >> shift, rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (exp >= -14) m &= 0x3ffu;
    uint32_t r = half + m;                                            // mantissa carry rolls into the exponent
    if (rem > mid || (rem == mid && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

template <typename T>
static void reduce_typed(T* dst, const T* src, size_t n, int op) {
    if (op == MIN) for (size_t i = 0; i < n; ++i) dst[i] = src[i] < dst[i] ? src[i] : dst[i];
    else if (op == MAX) for (size_t i = 0; i < n; ++i) dst[i] = src[i] > dst[i] ? src[i] : dst[i];
    else for (size_t i = 0; i < n; ++i) dst[i] += src[i];
}
static void reduce_into(void* dst, const void* src, size_t n, int dt, int op) {
    switch (dt) {
        case F32: reduce_typed((float*)dst, (const float*)src, n, op); break;
        case F64: reduce_typed((double*)dst, (const double*)src, n, op); break;
        case I64: reduce_typed((int64_t*)dst, (const int64_t*)src, n, op); break;
        case I32: reduce_typed((int32_t*)dst, (const int32_t*)src, n, op); break;
        case U8: reduce_typed((uint8_t*)dst, (const uint8_t*)src, n, op); break;
        case BF16: {
            uint16_t* d = (uint16_t*)dst;
            const uint16_t* s = (const uint16_t*)src;
            for (size_t i = 0; i < n; ++i) {
                float a = bf16_to_f(d[i]), b = bf16_to_f(s[i]);
                float r = op == MIN ? (b < a ? b : a) : op == MAX ? (b > a ? b : a) : a + b;
                d[i] = f_to_bf16(r);
            }
            break;
        }
        case F16: {
            uint16_t* d = (uint16_t*)dst;
            const uint16_t* s = (const uint16_t*)src;
            for (size_t i = 0; i < n; ++i) {
                float a = f16_to_f(d[i]), b = f16_to_f(s[i]);
                float r = op == MIN ? (b < a ? b : a) : op == MAX ? (b > a ? b : a) : a + b;
                d[i] = f_to_f16(r);
            }
            break;
        }
        default: break;
    }
}
static void scale_avg(void* buf, size_t n, int dt, int world) {
    switch (dt) {
        case F32: { float* p = (float*)buf; float s = 1.f / world; for (size_t i = 0; i < n; ++i) p[i] *= s; break; }
        case F64: { double* p = (double*)buf; double s = 1.0 / world; for (size_t i = 0; i < n; ++i) p[i] *= s; break; }
        case I64: { int64_t* p = (int64_t*)buf; for (size_t i = 0; i < n; ++i) p[i] = (int64_t)std::floor((double)p[i] / world); break; }
        case I32: { int32_t* p = (int32_t*)buf; for (size_t i = 0; i < n; ++i) p[i] = (int32_t)std::floor((double)p[i] / world); break; }
        case BF16: { uint16_t* p = (uint16_t*)buf; float s = 1.f / world; for (size_t i = 0; i < n; ++i) p[i] = f_to_bf16(bf16_to_f(p[i]) * s); break; }
        case F16: { uint16_t* p = (uint16_t*)buf; float s = 1.f / world; for (size_t i = 0; i < n; ++i) p[i] = f_to_f16(f16_to_f(p[i]) * s); break; }
        case U8: { uint8_t* p = (uint8_t*)buf; for (size_t i = 0; i < n; ++i) p[i] = (uint8_t)(p[i] / world); break; }
        default: break;
    }
}

static uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Op {
    int kind, dtype, redop, root;
    void* buf;       // in/out (allreduce, broadcast), input (allgather)
    void* out;       // allgather output [world * count]
    int64_t count;
    std::string name;
    int64_t handle;
    double t_enqueue;
    bool device = false;   // device memory: launched on the device plane's stream by the comm thread
    DeviceOp dop;
};

// A device collective between its launch and the moment both the watchdog has seen it complete
// and the consumer has queued its stream wait (the events are recycled only then).
struct Inflight {
    DeviceTicket ticket;
    std::string name;
    double t_launch = 0.0;
    bool completed = false, released = false, warned = false;
};

struct CollectiveStat { long long count = 0; double total_ms = 0.0, last_ms = 0.0; int algo = 0; long long bytes = 0; };

static const DevicePlane* g_plane = nullptr;
void set_device_plane(const DevicePlane* plane) { g_plane = plane; }

struct Header { uint32_t magic; int32_t kind, dtype; int64_t count; uint64_t name_hash; };

// A store server started BEFORE Engine::init by ddl_eng_listen(): rank 0 lets the kernel pick the
// port (bind to 0), learns the port that was actually bound and publishes it out of band (the
// torchrun agent's store) - no window in which another process can take a probed-and-closed port.
static StoreServer* g_prestarted = nullptr;
static int g_prestarted_port = 0;

class Engine {
public:
    int rank = 0, size = 1;
    double stall_warning_s = 60.0, timeout_s = 300.0;
    bool check_names = true;

    ~Engine() { shutdown(); }

    bool init(int rank_, int size_, const std::string& addr, int port, double timeout_s_, double stall_s,
              std::string* err) {
        rank = rank_; size = size_; timeout_s = timeout_s_; stall_warning_s = stall_s;
        const int tmo = (int)(timeout_s * 1000);
        if (rank == 0 && g_prestarted && (port == g_prestarted_port || port == 0)) {
            server_ = g_prestarted;         // already listening (ddl_eng_listen)
            port = g_prestarted_port;
            g_prestarted = nullptr;
        } else if (rank == 0) {
            int bound = 0;
            // a single-node job (the rendezvous address is the loopback interface) keeps its control
            // plane off the network; multi-node jobs listen on every interface
            const bool loopback = addr == "127.0.0.1" || addr == "localhost";
            server_ = store_server_start(port, &bound, loopback);
            if (!server_) { *err = "cannot listen on control port " + std::to_string(port); return false; }
        }
        if (!store_.connect_to(addr, port, tmo)) { *err = "cannot reach the rendezvous store at " + addr; return false; }
        if (size > 1) {
            // ring wiring: publish my data port, connect to next, accept from prev
            int dport = 0;
            listen_fd_ = tcp_listen(0, &dport, addr == "127.0.0.1" || addr == "localhost");
            if (listen_fd_ < 0) { *err = "cannot open a data socket"; return false; }
            char host[256] = {0};
            const char* adv = getenv("DDL_ADVERTISE_ADDR");
            if (adv) snprintf(host, sizeof(host), "%s", adv);
            else if (addr == "127.0.0.1" || addr == "localhost") snprintf(host, sizeof(host), "127.0.0.1");
            else gethostname(host, sizeof(host) - 1);
            store_.set("data/" + std::to_string(rank), std::string(host) + ":" + std::to_string(dport));
            std::string peer;
            const int next = (rank + 1) % size;
            if (!store_.get("data/" + std::to_string(next), &peer, tmo)) { *err = "peer did not register"; return false; }
            const size_t colon = peer.rfind(':');
            std::thread acc([&] { recv_fd_ = tcp_accept(listen_fd_, tmo); });
            send_fd_ = tcp_connect(peer.substr(0, colon), atoi(peer.c_str() + colon + 1), tmo);
            acc.join();
            if (send_fd_ < 0 || recv_fd_ < 0) { *err = "ring connection failed"; return false; }
            int32_t me = rank, other = -1;
            if (!send_all(send_fd_, &me, 4) || !recv_all(recv_fd_, &other, 4, tmo) ||
                other != (rank + size - 1) % size) { *err = "ring handshake failed"; return false; }
        }
        running_ = true;
        worker_ = std::thread([this] { loop(); });
        watchdog_ = std::thread([this] { watch(); });
        return true;
    }

    void shutdown() {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!running_) return;
            running_ = false;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        if (watchdog_.joinable()) watchdog_.join();
        if (g_plane) {
            std::lock_guard<std::mutex> g(mu_);
            for (auto& kv : inflight_) g_plane->recycle(&kv.second.ticket);
            inflight_.clear();
        }
        timeline_stop();
        if (send_fd_ >= 0) ::close(send_fd_);
        if (recv_fd_ >= 0) ::close(recv_fd_);
        if (listen_fd_ >= 0) ::close(listen_fd_);
        send_fd_ = recv_fd_ = listen_fd_ = -1;
        store_.close_fd();
        if (server_) { store_server_stop(server_); server_ = nullptr; }
    }

    int64_t submit(Op op) {
        std::lock_guard<std::mutex> g(mu_);
        op.handle = next_handle_++;
        op.t_enqueue = now_us();
        status_[op.handle] = PENDING;
        timeline_event(op.name, "ENQUEUE", 'i', op.t_enqueue);
        queue_.push_back(std::move(op));
        cv_.notify_all();
        return queue_.back().handle;
    }

    int poll(int64_t h) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = status_.find(h);
        if (it == status_.end()) return -1;
        if (it->second == RUNNING) {  // a launched device collective: ask its completion event
            auto f = inflight_.find(h);
            if (f != inflight_.end() && g_plane && (f->second.completed || g_plane->query(f->second.ticket) == 1)) return DONE;
        }
        return it->second;
    }

    // Device collectives: the consumer's stream waits for the collective, the host only waits
    // until the comm thread has LAUNCHED it (microseconds).  0 = ordered, 1 = timeout (not yet
    // launched), 2 = failed (a peer died, a collective timed out ...), -1 = unknown handle.
    // `order` false: bookkeeping only (the host already saw the completion event).  A null
    // consumer_stream with `order` true is HIP's default stream - torch's default stream IS that.
    int device_wait(int64_t h, void* consumer_stream, double timeout_s_, std::string* err, bool order = true) {
        std::unique_lock<std::mutex> lk(mu_);
        auto launched = [&] { auto it = status_.find(h); return it == status_.end() || it->second >= RUNNING; };
        if (!done_cv_.wait_for(lk, std::chrono::duration<double>(timeout_s_), launched)) return 1;
        auto it = status_.find(h);
        if (it == status_.end()) return -1;
        if (it->second == FAILED) { *err = last_error_; status_.erase(it); inflight_.erase(h); return 2; }
        auto f = inflight_.find(h);
        if (f == inflight_.end() || !g_plane) return -1;
        const bool ok = !order || g_plane->wait_on(f->second.ticket, consumer_stream);
        f->second.released = true;
        if (f->second.completed) { g_plane->recycle(&f->second.ticket); inflight_.erase(f); }
        status_.erase(it);
        if (!ok) { *err = "hipStreamWaitEvent failed"; return 2; }
        if (device_failed_.load()) { *err = last_error_; return 2; }
        return 0;
    }

    // host-blocking completion of a device collective (tests, shutdown paths, hvd.synchronize on
    // a handle whose result the HOST reads next)
    int device_wait_host(int64_t h, double timeout_s_, std::string* err) {
        const double t0 = now_us();
        for (;;) {
            int st = poll(h);
            if (st == -1) return -1;
            if (st == FAILED || device_failed_.load()) { std::lock_guard<std::mutex> g(mu_); *err = last_error_; return 2; }
            if (st == DONE) break;
            if ((now_us() - t0) * 1e-6 > timeout_s_) return 1;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        return device_wait(h, nullptr, timeout_s_, err, false);
    }

    bool failed() const { return device_failed_.load(); }
    void harvest() { watch_device(); }

    // "name count last_ms mean_ms algo bytes" per line: device time of every named collective
    std::string stats_text() {
        std::lock_guard<std::mutex> g(mu_);
        std::string out;
        char line[320];
        for (auto& kv : stats_) {
            snprintf(line, sizeof(line), "%s %lld %.4f %.4f %d %lld\n", kv.first.c_str(), kv.second.count, kv.second.last_ms,
                     kv.second.count ? kv.second.total_ms / kv.second.count : 0.0, kv.second.algo, kv.second.bytes);
            out += line;
        }
        return out;
    }

    // 0 = done, 1 = timeout, 2 = failed, -1 = unknown handle
    int wait(int64_t h, double timeout_s_, std::string* err) {
        std::unique_lock<std::mutex> lk(mu_);
        auto done = [&] { auto it = status_.find(h); return it == status_.end() || it->second >= DONE; };
        if (!done_cv_.wait_for(lk, std::chrono::duration<double>(timeout_s_), done)) return 1;
        auto it = status_.find(h);
        if (it == status_.end()) return -1;
        int st = it->second;
        status_.erase(it);
        if (st == FAILED) { *err = last_error_; return 2; }
        return 0;
    }

    // store passthroughs (object broadcast, barrier keys ...)
    StoreClient& store() { return store_; }
    std::mutex store_mu;

    // ---- timeline --------------------------------------------------------------------------
    bool timeline_start(const std::string& path) {
        std::lock_guard<std::mutex> g(tl_mu_);
        if (tl_) return true;
        tl_ = fopen(path.c_str(), "w");
        if (!tl_) return false;
        fprintf(tl_, "[\n");
        tl_first_ = true;
        return true;
    }
    void timeline_stop() {
        std::lock_guard<std::mutex> g(tl_mu_);
        if (!tl_) return;
        fprintf(tl_, "\n]\n");
        fclose(tl_);
        tl_ = nullptr;
    }
    void timeline_event(const std::string& name, const char* cat, char ph, double ts) {
        std::lock_guard<std::mutex> g(tl_mu_);
        if (!tl_) return;
        fprintf(tl_, "%s{\"name\": \"%s\", \"cat\": \"%s\", \"ph\": \"%c\", \"ts\": %.1f, \"pid\": %d, \"tid\": 0%s}",
                tl_first_ ? "" : ",\n", name.c_str(), cat, ph, ts, rank, ph == 'i' ? ", \"s\": \"t\"" : "");
        tl_first_ = false;
        fflush(tl_);
    }

    std::atomic<int> stall_warnings{0};

private:
    StoreServer* server_ = nullptr;
    StoreClient store_;
    int listen_fd_ = -1, send_fd_ = -1, recv_fd_ = -1;
    std::thread worker_, watchdog_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Op> queue_;
    std::map<int64_t, int> status_;
    int64_t next_handle_ = 1;
    std::atomic<bool> running_{false};  // written under mu_, read lock-free inside socket loops
    std::string last_error_;
    std::atomic<double> current_start_{0.0};
    std::string current_name_;  // guarded by cur_mu_ (worker writes, watchdog reads)
    std::mutex cur_mu_;
    std::mutex tl_mu_;
    FILE* tl_ = nullptr;
    bool tl_first_ = true;
    std::vector<char> scratch_;
    std::map<int64_t, Inflight> inflight_;        // guarded by mu_
    std::map<std::string, CollectiveStat> stats_; // guarded by mu_
    std::atomic<bool> device_failed_{false};
    bool comm_bound_ = false;

    void fail_all(const std::string& why) {
        std::lock_guard<std::mutex> g(mu_);
        last_error_ = why;
        for (auto& kv : status_)
            if (kv.second < DONE) kv.second = FAILED;
        queue_.clear();
        done_cv_.notify_all();
    }

    void watch() {
        bool warned = false;
        double warned_for = 0.0;
        // wake at least twice per warning interval (200 ms at most, 1 ms at least)
        const int period_ms = (int)std::min(200.0, std::max(1.0, stall_warning_s * 500.0));
        while (true) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (cv_.wait_for(lk, std::chrono::milliseconds(period_ms), [&] { return !running_; })) return;
            }
            watch_device();
            const double st = current_start_.load();
            if (st <= 0.0) { warned = false; continue; }
            const double el = (now_us() - st) * 1e-6;
            if (el > stall_warning_s && !(warned && warned_for == st)) {
                std::string name;
                {
                    std::lock_guard<std::mutex> g(cur_mu_);
                    name = current_name_;
                }
                fprintf(stderr, "[ddl engine rank %d] WARNING: collective '%s' has been waiting for %.1f s "
                                "(a peer may have stalled or submitted collectives in a different order)\n",
                        rank, name.c_str(), el);
                warned = true; warned_for = st;
                ++stall_warnings;
            }
            if (el > timeout_s) {
                // break the ring: blocked socket calls return, the worker fails every handle
                if (send_fd_ >= 0) ::shutdown(send_fd_, SHUT_RDWR);
                if (recv_fd_ >= 0) ::shutdown(recv_fd_, SHUT_RDWR);
            }
        }
    }

    // Device collectives in flight: completion (-> per-collective device time), stall warning, and
    // the hard timeout.  A collective that does not finish within `timeout_s` (a dead or wedged
    // peer) is aborted and every handle fails: synchronize() raises instead of hanging.
    // bound PER THREAD: harvest() runs this from Python threads as well as from the watchdog thread,
    // and a flag shared between them left the watchdog thread without its hipSetDevice
    static bool& thread_bound() { static thread_local bool b = false; return b; }
    // per-collective device times are kept for stable names only (the gradient buckets, the init
    // self-test, named user collectives): auto-named ops (`allreduce.noname.<n>`) would grow the map
    // without bound and push the grad.* lines out of the caller's buffer
    static bool keeps_stats(const std::string& name) { return name.find(".noname.") == std::string::npos; }
    static constexpr size_t kMaxStats = 512;
    void watch_device() {
        if (!g_plane) return;
        std::string dead;
        double dead_for = 0.0;
        {
            std::lock_guard<std::mutex> g(mu_);
            // a kernel that gave up on a peer (bounded spins) has already completed: its ticket may
            // be gone, the host-visible timeout word is what is left of it
            const bool ktmo = g_plane->kernel_timeout() != 0;
            if (ktmo && inflight_.empty()) dead = "xgmi all-reduce (a peer never arrived: spin limit reached)";
            if (inflight_.empty() && !ktmo) return;
            if (!thread_bound()) thread_bound() = g_plane->bind_thread();
            const double now = now_us();
            for (auto it = inflight_.begin(); it != inflight_.end();) {
                Inflight& f = it->second;
                if (!f.completed) {
                    const int q = g_plane->query(f.ticket);
                    if (q == 1 && !ktmo) {
                        f.completed = true;
                        const float ms = g_plane->elapsed_ms(f.ticket);
                        auto sit = stats_.find(f.name);
                        if (sit != stats_.end()) { auto& st = sit->second; ++st.count; st.last_ms = ms; st.total_ms += ms; }
                        timeline_event(f.name, "DEVICE_DONE", 'i', now);
                    } else {
                        const double el = (now - f.t_launch) * 1e-6;
                        if (el > stall_warning_s && !f.warned) {
                            fprintf(stderr, "[ddl engine rank %d] WARNING: device collective '%s' has been in flight for %.1f s\n",
                                    rank, f.name.c_str(), el);
                            f.warned = true;
                            ++stall_warnings;
                        }
                        if (q < 0 || ktmo || el > timeout_s) { dead = f.name; dead_for = el; break; }
                    }
                }
                if (f.completed && f.released) {
                    g_plane->recycle(&f.ticket);
                    it = inflight_.erase(it);
                } else {
                    ++it;
                }
            }
        }
        if (!dead.empty() && !device_failed_.exchange(true)) {
            char why[320];
            snprintf(why, sizeof(why), "device collective '%s' did not complete within %.1f s (in flight for %.1f s); aborting the communicator",
                     dead.c_str(), timeout_s, dead_for);
            fprintf(stderr, "[ddl engine rank %d] ERROR: %s\n", rank, why);
            // Python first: every waiter raises NOW.  The plane's abort (ncclCommAbort, then the
            // diagnostic snapshot) may itself take time on a wedged device and must never stand
            // between a dead collective and the error the job is waiting for.
            fail_all(why);
            g_plane->abort(why);
        }
    }

    // simultaneous send (to next) / receive (from prev) of equally sized blocks
    bool sendrecv(const void* sbuf, void* rbuf, size_t n) {
        const char* sp = (const char*)sbuf;
        char* rp = (char*)rbuf;
        size_t sl = n, rl = n;
        while (sl > 0 || rl > 0) {
            pollfd pf[2] = {{send_fd_, (short)(sl ? POLLOUT : 0), 0}, {recv_fd_, (short)(rl ? POLLIN : 0), 0}};
            int r = ::poll(pf, 2, 1000);
            if (r < 0) { if (errno == EINTR) continue; return false; }
            if (!running_) return false;
            if ((pf[0].revents | pf[1].revents) & (POLLERR | POLLNVAL)) return false;
            if (sl && (pf[0].revents & POLLOUT)) {
                ssize_t k = ::send(send_fd_, sp, sl, MSG_NOSIGNAL | MSG_DONTWAIT);
                if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return false;
                if (k > 0) { sp += k; sl -= (size_t)k; }
            }
            if (rl && (pf[1].revents & (POLLIN | POLLHUP))) {
                ssize_t k = ::recv(recv_fd_, rp, rl, MSG_DONTWAIT);
                if (k == 0) return false;
                if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return false;
                if (k > 0) { rp += k; rl -= (size_t)k; }
            }
        }
        return true;
    }

    bool check_header(const Op& op, std::string* err) {
        Header mine{0x44444c31u, op.kind, op.dtype, op.count, check_names ? fnv1a(op.name) : 0}, theirs{};
        if (!sendrecv(&mine, &theirs, sizeof(Header))) { *err = "connection to a peer was lost"; return false; }
        if (theirs.magic != mine.magic || theirs.kind != mine.kind || theirs.dtype != mine.dtype ||
            (op.kind != ALLGATHER && theirs.count != mine.count) || theirs.name_hash != mine.name_hash) {
            *err = "collective mismatch across ranks at '" + op.name + "' (kind/dtype/count/name differ)";
            return false;
        }
        return true;
    }

    bool do_allreduce(const Op& op, std::string* err) {
        const size_t es = dtype_size(op.dtype);
        const int n = size;
        const int64_t cnt = op.count;
        char* data = (char*)op.buf;
        // chunk c covers [off[c], off[c+1])
        std::vector<int64_t> off(n + 1);
        for (int c = 0; c <= n; ++c) off[c] = cnt * c / n;
        int64_t maxc = 0;
        for (int c = 0; c < n; ++c) maxc = std::max(maxc, off[c + 1] - off[c]);
        scratch_.resize((size_t)maxc * es);
        auto len = [&](int c) { return (size_t)(off[c + 1] - off[c]); };
        // every step exchanges max-size blocks so that both directions agree on the byte count
        std::vector<char> sendblk((size_t)maxc * es);
        for (int s = 0; s < n - 1; ++s) {  // reduce-scatter
            int sc = ((rank - s) % n + n) % n, rc = ((rank - s - 1) % n + n) % n;
            std::memcpy(sendblk.data(), data + off[sc] * es, len(sc) * es);
            if (!sendrecv(sendblk.data(), scratch_.data(), (size_t)maxc * es)) { *err = "connection to a peer was lost"; return false; }
            reduce_into(data + off[rc] * es, scratch_.data(), len(rc), op.dtype, op.redop == AVG ? SUM : op.redop);
        }
        {
            int own = ((rank + 1) % n + n) % n;  // fully reduced chunk held by this rank
            if (op.redop == AVG) scale_avg(data + off[own] * es, len(own), op.dtype, n);
        }
        for (int s = 0; s < n - 1; ++s) {  // all-gather
            int sc = ((rank + 1 - s) % n + n) % n, rc = ((rank - s) % n + n) % n;
            std::memcpy(sendblk.data(), data + off[sc] * es, len(sc) * es);
            if (!sendrecv(sendblk.data(), scratch_.data(), (size_t)maxc * es)) { *err = "connection to a peer was lost"; return false; }
            std::memcpy(data + off[rc] * es, scratch_.data(), len(rc) * es);
        }
        return true;
    }

    bool do_broadcast(const Op& op, std::string* err) {
        const size_t bytes = (size_t)op.count * dtype_size(op.dtype);
        const int last = (op.root + size - 1) % size;  // the rank just before the root forwards nothing
        if (rank != op.root && !recv_all(recv_fd_, op.buf, bytes, (int)(timeout_s * 1000))) { *err = "connection to a peer was lost"; return false; }
        if (rank != last && !send_all(send_fd_, op.buf, bytes)) { *err = "connection to a peer was lost"; return false; }
        return true;
    }

    bool do_allgather(const Op& op, std::string* err) {
        const size_t bytes = (size_t)op.count * dtype_size(op.dtype);
        char* out = (char*)op.out;
        std::memcpy(out + (size_t)rank * bytes, op.buf, bytes);
        for (int s = 0; s < size - 1; ++s) {
            int sc = ((rank - s) % size + size) % size, rc = ((rank - s - 1) % size + size) % size;
            if (!sendrecv(out + (size_t)sc * bytes, out + (size_t)rc * bytes, bytes)) { *err = "connection to a peer was lost"; return false; }
        }
        return true;
    }

    void loop() {
        for (;;) {
            Op op;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !running_ || !queue_.empty(); });
                if (!running_ && queue_.empty()) return;
                op = std::move(queue_.front());
                queue_.pop_front();
                if (!op.device) status_[op.handle] = RUNNING;
            }
            {
                std::lock_guard<std::mutex> g(cur_mu_);
                current_name_ = op.name;
            }
            const double t0 = now_us();
            if (op.device) {
                // device collective: launch only (stream-ordered), completion is the watchdog's and
                // the consumer stream's business; the comm thread goes straight on to the next op
                std::string derr;
                Inflight fl;
                fl.name = op.name;
                bool ok = g_plane != nullptr;
                if (ok && !comm_bound_) { ok = g_plane->bind_thread(); comm_bound_ = ok; }
                if (!ok) derr = "no device plane attached to the engine";
                else ok = g_plane->launch(op.dop, &fl.ticket, &derr);
                fl.t_launch = now_us();
                timeline_event(op.name, op.dop.algo == ALGO_RCCL ? "RCCL_LAUNCH" : "XGMI_LAUNCH", 'i', fl.t_launch);
                if (!ok) {
                    {
                        std::lock_guard<std::mutex> g(mu_);
                        status_[op.handle] = FAILED;
                    }
                    fail_all("device collective '" + op.name + "' could not be launched: " + derr);
                    continue;
                }
                {
                    std::lock_guard<std::mutex> g(mu_);
                    if (keeps_stats(op.name) && (stats_.count(op.name) || stats_.size() < kMaxStats)) {
                        auto& st = stats_[op.name];
                        st.algo = fl.ticket.algo;
                        st.bytes = op.count * (long long)dtype_size(op.dtype);
                    }
                    inflight_[op.handle] = fl;
                    status_[op.handle] = RUNNING;
                }
                done_cv_.notify_all();
                continue;
            }
            current_start_.store(t0);
            timeline_event(op.name, op.kind == ALLREDUCE ? "ALLREDUCE" : op.kind == BROADCAST ? "BROADCAST"
                                   : op.kind == ALLGATHER ? "ALLGATHER" : "BARRIER", 'B', t0);
            std::string err;
            bool ok = true;
            if (size > 1) {
                ok = check_header(op, &err);
                if (ok) {
                    if (op.kind == ALLREDUCE) ok = do_allreduce(op, &err);
                    else if (op.kind == BROADCAST) ok = do_broadcast(op, &err);
                    else if (op.kind == ALLGATHER) ok = do_allgather(op, &err);
                }
            } else if (op.kind == ALLGATHER) {
                std::memcpy(op.out, op.buf, (size_t)op.count * dtype_size(op.dtype));
            }
            timeline_event(op.name, "", 'E', now_us());
            current_start_.store(0.0);
            if (!ok) {
                {
                    std::lock_guard<std::mutex> g(mu_);
                    status_[op.handle] = FAILED;
                }
                fail_all(err);
                continue;
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                status_[op.handle] = DONE;
            }
            done_cv_.notify_all();
        }
    }
};

}  // namespace ddl

// ---------------------------------------------------------------------------------------------
// C API (ctypes).  One engine per process.
// ---------------------------------------------------------------------------------------------
static ddl::Engine* g_engine = nullptr;
static std::string g_err;
static std::string g_buf;

extern "C" {

int ddl_eng_init(int rank, int size, const char* addr, int port, double timeout_s, double stall_warning_s) {
    if (g_engine) return 0;
    auto* e = new ddl::Engine();
    if (!e->init(rank, size, addr ? addr : "127.0.0.1", port, timeout_s, stall_warning_s, &g_err)) {
        delete e;
        return -1;
    }
    g_engine = e;
    return 0;
}

// Rank 0 only, before ddl_eng_init: start the control-plane store on `port` (0 = any free port)
// and return the port actually bound (-1 on failure).  ddl_eng_init(…, that port, …) adopts it.
int ddl_eng_listen(int port, int loopback_only) {
    if (ddl::g_prestarted) return ddl::g_prestarted_port;
    int bound = 0;
    ddl::StoreServer* s = ddl::store_server_start(port, &bound, loopback_only != 0);
    if (!s) { g_err = "cannot listen on control port " + std::to_string(port); return -1; }
    ddl::g_prestarted = s;
    ddl::g_prestarted_port = bound;
    return bound;
}

void ddl_eng_shutdown() {
    if (g_engine) {
        g_engine->shutdown();
        delete g_engine;
        g_engine = nullptr;
    }
}

const char* ddl_eng_last_error() { return g_err.c_str(); }
int ddl_eng_rank() { return g_engine ? g_engine->rank : -1; }
int ddl_eng_size() { return g_engine ? g_engine->size : -1; }
int ddl_eng_stall_warnings() { return g_engine ? g_engine->stall_warnings.load() : 0; }

long long ddl_eng_allreduce_async(void* buf, long long count, int dtype, int redop, const char* name) {
    if (!g_engine) return -1;
    ddl::Op op{ddl::ALLREDUCE, dtype, redop, 0, buf, nullptr, count, name ? name : "", 0, 0.0};
    return g_engine->submit(std::move(op));
}
long long ddl_eng_broadcast_async(void* buf, long long count, int dtype, int root, const char* name) {
    if (!g_engine || root < 0 || root >= g_engine->size) return -1;
    ddl::Op op{ddl::BROADCAST, dtype, 0, root, buf, nullptr, count, name ? name : "", 0, 0.0};
    return g_engine->submit(std::move(op));
}
long long ddl_eng_allgather_async(const void* in, void* out, long long count, int dtype, const char* name) {
    if (!g_engine) return -1;
    ddl::Op op{ddl::ALLGATHER, dtype, 0, 0, const_cast<void*>(in), out, count, name ? name : "", 0, 0.0};
    return g_engine->submit(std::move(op));
}
long long ddl_eng_barrier_async(const char* name) {
    if (!g_engine) return -1;
    static long long token;  // 8-byte dummy all-reduce
    ddl::Op op{ddl::ALLREDUCE, ddl::I64, ddl::SUM, 0, &token, nullptr, 1, name ? name : "barrier", 0, 0.0};
    return g_engine->submit(std::move(op));
}
// Device collective through the engine: the submitting thread records the readiness event on its
// producer stream, the comm thread launches on the device plane's stream (rccl_backend.cpp).
// kind: 0 all-reduce, 1 broadcast, 2 all-gather; algo: 0 RCCL, 1 xGMI two-shot, 2 xGMI one-shot.
long long ddl_eng_device_collective(int kind, void* buf, void* out, long long count, int dtype, int redop, int root,
                                    int algo, void* producer_stream, const char* name) {
    if (!g_engine) return -1;
    if (!ddl::g_plane) { g_err = "device tensor given but no device plane is attached (RCCL backend not initialised)"; return -1; }
    if (kind < 0 || kind > 2 || !buf || count < 0 || (kind == ddl::ALLGATHER && !out) ||
        (kind == ddl::BROADCAST && (root < 0 || root >= g_engine->size))) { g_err = "invalid device collective"; return -1; }
    if (g_engine->failed()) { g_err = "the engine has failed; no further collectives are accepted"; return -1; }
    ddl::Op op{kind, dtype, redop, root, buf, out, count, name ? name : "", 0, 0.0};
    op.device = true;
    op.dop.kind = kind; op.dop.dtype = dtype; op.dop.redop = redop; op.dop.root = root; op.dop.algo = algo;
    op.dop.buf = buf; op.dop.out = out; op.dop.count = count;
    op.dop.ready = ddl::g_plane->record_ready(producer_stream);
    if (!op.dop.ready) { g_err = "readiness event could not be recorded"; return -1; }
    return g_engine->submit(std::move(op));
}
int ddl_eng_device_wait(long long handle, void* consumer_stream, double timeout_s) {
    if (!g_engine) return -1;
    return g_engine->device_wait(handle, consumer_stream, timeout_s, &g_err);
}
int ddl_eng_device_wait_host(long long handle, double timeout_s) {
    if (!g_engine) return -1;
    return g_engine->device_wait_host(handle, timeout_s, &g_err);
}
int ddl_eng_failed() { return g_engine && g_engine->failed() ? 1 : 0; }
// per-collective device times ("name count last_ms mean_ms algo bytes" lines); returns the length
long long ddl_eng_stats(char* buf, long long cap) {
    if (!g_engine || !buf || cap <= 0) return -1;
    g_engine->harvest();   // record collectives that completed since the watchdog's last pass
    const std::string t = g_engine->stats_text();
    const long long n = (long long)t.size() < cap - 1 ? (long long)t.size() : cap - 1;
    std::memcpy(buf, t.data(), (size_t)n);
    buf[n] = 0;
    return (long long)t.size();   // the REQUIRED length: a caller that sees >= cap grows its buffer and asks again
}

int ddl_eng_poll(long long handle) { return g_engine ? g_engine->poll(handle) : -1; }
int ddl_eng_wait(long long handle, double timeout_s) {
    if (!g_engine) return -1;
    return g_engine->wait(handle, timeout_s, &g_err);
}

// control-plane key-value access (object broadcast, resume epoch ...)
int ddl_eng_store_set(const char* key, const void* val, long long n) {
    if (!g_engine) return -1;
    std::lock_guard<std::mutex> g(g_engine->store_mu);
    return g_engine->store().set(key, std::string((const char*)val, (size_t)n)) ? 0 : -1;
}
long long ddl_eng_store_get(const char* key, int timeout_ms) {  // returns length, data via ddl_eng_store_data
    if (!g_engine) return -1;
    std::lock_guard<std::mutex> g(g_engine->store_mu);
    return g_engine->store().get(key, &g_buf, timeout_ms) ? (long long)g_buf.size() : -1;
}
const void* ddl_eng_store_data() { return g_buf.data(); }
int ddl_eng_store_del(const char* key) {
    if (!g_engine) return -1;
    std::lock_guard<std::mutex> g(g_engine->store_mu);
    return g_engine->store().del(key) ? 0 : -1;
}
long long ddl_eng_store_add(const char* key, long long delta) {
    if (!g_engine) return -1;
    std::lock_guard<std::mutex> g(g_engine->store_mu);
    int64_t r = 0;
    return g_engine->store().add(key, delta, &r) ? (long long)r : -1;
}

int ddl_eng_timeline_start(const char* path) { return g_engine && g_engine->timeline_start(path) ? 0 : -1; }
void ddl_eng_timeline_stop() { if (g_engine) g_engine->timeline_stop(); }
void ddl_eng_timeline_mark(const char* name, const char* cat, int phase) {
    if (g_engine) g_engine->timeline_event(name, cat, phase == 0 ? 'B' : phase == 1 ? 'E' : 'i', ddl::now_us());
}

// ---------------------------------------------------------------------------------------------
// Static bucket planner (native twin of parallel/bucket.py:plan_arenas for one dtype).
// numel[i] in *planning order* (already reversed); outputs offsets[i] (elements) and bucket[i].
// Returns the number of buckets; *arena_numel gets the padded arena length.
// ---------------------------------------------------------------------------------------------
int ddl_bucket_plan(const long long* numel, int n, int elem_size, long long first_bucket_bytes,
                    long long bucket_cap_bytes, long long tail_bucket_bytes, int seg_align_bytes, int bucket_align_bytes,
                    long long* offsets, int* bucket, long long* bucket_end, long long* arena_numel) {
    if (n <= 0 || elem_size <= 0) return 0;
    const long long seg_align = std::max<long long>(seg_align_bytes / elem_size, 1);
    const long long bkt_align = std::max<long long>(bucket_align_bytes / elem_size, 1);
    auto round_up = [](long long x, long long m) { return (x + m - 1) / m * m; };
    // small tail bucket: the last tensors (first layers) only become ready when backward ends
    int tail_start = -1;
    long long total = 0;
    for (int i = 0; i < n; ++i) total += numel[i] * elem_size;
    if (tail_bucket_bytes > 0 && total > 4 * tail_bucket_bytes) {
        long long acc = 0;
        for (int i = n - 1; i >= 1; --i) {
            const long long sz = numel[i] * elem_size;
            if (acc > 0 && acc + sz > tail_bucket_bytes) break;
            acc += sz;
            tail_start = i;
        }
    }
    long long cursor = 0, start = 0;
    int nb = 0, in_cur = 0;
    for (int i = 0; i < n; ++i) {
        const long long cap = nb == 0 ? first_bucket_bytes : bucket_cap_bytes;
        const long long used = (cursor - start) * elem_size;
        if (in_cur > 0 && (used + numel[i] * elem_size > cap || i == tail_start)) {
            cursor = round_up(cursor, bkt_align);
            bucket_end[nb] = cursor;
            ++nb;
            start = cursor;
            in_cur = 0;
        }
        offsets[i] = cursor;
        bucket[i] = nb;
        ++in_cur;
        cursor = round_up(cursor + numel[i], seg_align);
    }
    cursor = round_up(cursor, bkt_align);
    bucket_end[nb] = cursor;
    *arena_numel = cursor;
    return nb + 1;
}

}  // extern "C"
