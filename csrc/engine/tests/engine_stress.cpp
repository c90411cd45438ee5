// Native stress driver for the gradient-exchange engine ("building the C++ engine
// with -fsanitize=thread against the CPU backend").  Linked against engine.cpp + store.cpp only
// (no HIP, no RCCL), so it runs anywhere; `python build.py --tsan` builds it with
// ThreadSanitizer and tests/test_native_engine.py runs it.
//
//   engine_stress [ranks=3] [rounds=40] [stall_warning_s=30] [fault_round=-1]
//
// A tiny stall_warning_s (e.g. 0.0005) makes the watchdog thread report on nearly every
// collective, which exercises its reads of the worker's state.  With fault_round >= 0 the last
// rank leaves without a word at that round; every other rank must then see its waits FAIL (not
// hang) and still shut down cleanly - that run passes when the loss surfaced everywhere.
//
// The parent forks one process per rank.  Every rank starts the engine on the TCP data plane and
//   * one thread submits a fixed sequence of mixed collectives (all-reduce sum/avg/min/max over
//     f32/f64/i64/i32/bf16 with uneven counts, broadcasts from rotating roots, all-gathers),
//   * a second thread polls the handles while they are in flight,
//   * a third thread hammers the control-plane store and the timeline,
// then every result word is checked.  Exit status 0 = every rank passed.
#include <sys/socket.h>
#include <sys/wait.h>
#include <netinet/in.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
int ddl_eng_init(int rank, int size, const char* addr, int port, double timeout_s, double stall_warning_s);
void ddl_eng_shutdown();
const char* ddl_eng_last_error();
int ddl_eng_stall_warnings();
long long ddl_eng_allreduce_async(void* buf, long long count, int dtype, int redop, const char* name);
long long ddl_eng_broadcast_async(void* buf, long long count, int dtype, int root, const char* name);
long long ddl_eng_allgather_async(const void* in, void* out, long long count, int dtype, const char* name);
long long ddl_eng_barrier_async(const char* name);
int ddl_eng_poll(long long handle);
int ddl_eng_wait(long long handle, double timeout_s);
int ddl_eng_store_set(const char* key, const void* val, long long n);
long long ddl_eng_store_get(const char* key, int timeout_ms);
long long ddl_eng_store_add(const char* key, long long delta);
int ddl_eng_timeline_start(const char* path);
void ddl_eng_timeline_stop();
void ddl_eng_timeline_mark(const char* name, const char* cat, int phase);
}

enum { F32 = 0, F64 = 1, I64 = 2, BF16 = 3, I32 = 4 };
enum { SUM = 0, AVG = 1, MIN = 2, MAX = 3 };

static uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float from_bf16(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

static int free_port() {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = 0;
    if (::bind(fd, (sockaddr*)&a, sizeof(a)) != 0) { ::close(fd); return 29517; }
    socklen_t l = sizeof(a);
    ::getsockname(fd, (sockaddr*)&a, &l);
    int p = ntohs(a.sin_port);
    ::close(fd);
    return p;
}

struct Item {
    long long handle;
    int kind;  // 0 allreduce f32, 1 f64, 2 i64, 3 bf16, 4 i32, 5 broadcast, 6 allgather
    int redop, root;
    long long count;
    std::vector<char> buf, out;
};

// value contributed by `rank` at element i of round r (small integers: exact in bf16 too)
static inline int contrib(int rank, long long i, int r) { return (int)((i * 7 + r * 3 + rank * 5) % 13) - 6; }

static int run_rank(int rank, int size, int port, int rounds, double stall_s, int fault_round) {
    if (ddl_eng_init(rank, size, "127.0.0.1", port, fault_round >= 0 ? 5.0 : 60.0, stall_s) != 0) {
        fprintf(stderr, "[stress rank %d] init failed: %s\n", rank, ddl_eng_last_error());
        return 2;
    }
    char tl[128];
    snprintf(tl, sizeof(tl), "/tmp/ddl_engine_stress_%d_%d.json", (int)getppid(), rank);
    ddl_eng_timeline_start(tl);

    std::vector<Item> items((size_t)rounds * 7);
    std::atomic<int> submitted{0};
    std::atomic<bool> stop{false};
    std::atomic<long long> polls{0};

    // uneven load: ranks start submitting at different times and pause at different rounds
    std::thread poller([&] {
        while (!stop.load()) {
            const int n = submitted.load(std::memory_order_acquire);
            for (int k = std::max(0, n - 8); k < n; ++k) {
                if (ddl_eng_poll(items[(size_t)k].handle) >= 0) polls.fetch_add(1);
            }
            std::this_thread::yield();
        }
    });
    std::thread control([&] {
        int it = 0;
        while (!stop.load()) {
            std::string key = "stress/" + std::to_string(rank) + "/" + std::to_string(it % 4);
            std::string val = "v" + std::to_string(it);
            ddl_eng_store_set(key.c_str(), val.data(), (long long)val.size());
            ddl_eng_store_get(key.c_str(), 1000);
            ddl_eng_store_add("stress/counter", 1);
            ddl_eng_timeline_mark("control", "STRESS", it & 1);
            ++it;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });

    std::this_thread::sleep_for(std::chrono::milliseconds(3 * rank));
    int idx = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r == fault_round && rank == size - 1) _exit(0);  // dies mid-job: no shutdown, sockets just close
        if (r % (rank + 2) == 0) std::this_thread::sleep_for(std::chrono::microseconds(300 * (rank + 1)));
        for (int kind = 0; kind < 7; ++kind) {
            Item& it = items[(size_t)idx];
            it.kind = kind;
            it.redop = (r + kind) % 4;
            it.root = (r + kind) % size;
            it.count = 1 + (long long)((r * 131 + kind * 977) % 5003);  // not divisible by the ring size
            if (r % 9 == 0 && kind == 0) it.count = 300007;             // one long message per few rounds
            char name[64];
            snprintf(name, sizeof(name), "r%d.k%d", r, kind);
            const long long n = it.count;
            switch (kind) {
                case 0: { it.buf.resize((size_t)n * 4); float* p = (float*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = (float)contrib(rank, i, r);
                          it.handle = ddl_eng_allreduce_async(p, n, F32, it.redop, name); break; }
                case 1: { it.buf.resize((size_t)n * 8); double* p = (double*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = (double)contrib(rank, i, r);
                          it.handle = ddl_eng_allreduce_async(p, n, F64, it.redop, name); break; }
                case 2: { it.buf.resize((size_t)n * 8); int64_t* p = (int64_t*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = contrib(rank, i, r);
                          if (it.redop == AVG) it.redop = SUM;
                          it.handle = ddl_eng_allreduce_async(p, n, I64, it.redop, name); break; }
                case 3: { it.buf.resize((size_t)n * 2); uint16_t* p = (uint16_t*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = to_bf16((float)contrib(rank, i, r));
                          if (it.redop == AVG) it.redop = SUM;
                          it.handle = ddl_eng_allreduce_async(p, n, BF16, it.redop, name); break; }
                case 4: { it.buf.resize((size_t)n * 4); int32_t* p = (int32_t*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = contrib(rank, i, r);
                          if (it.redop == AVG) it.redop = SUM;
                          it.handle = ddl_eng_allreduce_async(p, n, I32, it.redop, name); break; }
                case 5: { it.buf.resize((size_t)n * 4); float* p = (float*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = rank == it.root ? (float)contrib(it.root, i, r) : -99.f;
                          it.handle = ddl_eng_broadcast_async(p, n, F32, it.root, name); break; }
                default: { it.buf.resize((size_t)n * 4); it.out.resize((size_t)n * 4 * size);
                          int32_t* p = (int32_t*)it.buf.data();
                          for (long long i = 0; i < n; ++i) p[i] = contrib(rank, i, r);
                          it.handle = ddl_eng_allgather_async(p, it.out.data(), n, I32, name); break; }
            }
            if (it.handle < 0) { fprintf(stderr, "[stress rank %d] submit failed\n", rank); return 3; }
            ++idx;
            submitted.store(idx, std::memory_order_release);
        }
    }
    long long bar = ddl_eng_barrier_async("end");

    int bad = 0;
    for (int k = 0; k < idx && bad == 0; ++k) {
        Item& it = items[(size_t)k];
        const int rc = ddl_eng_wait(it.handle, 120.0);
        if (rc != 0 && fault_round >= 0) {
            fprintf(stderr, "[stress rank %d] peer loss surfaced at item %d: %s\n", rank, k, ddl_eng_last_error());
            bad = -1;
            break;
        }
        if (rc != 0) { fprintf(stderr, "[stress rank %d] wait(%d) -> %d: %s\n", rank, k, rc, ddl_eng_last_error()); bad = 4; break; }
        const int r = k / 7;
        for (long long i = 0; i < it.count && bad == 0; ++i) {
            double want = 0.0, got = 0.0;
            if (it.kind <= 4) {
                double acc = it.redop == MIN ? 1e30 : it.redop == MAX ? -1e30 : 0.0;
                for (int q = 0; q < size; ++q) {
                    const double v = contrib(q, i, r);
                    acc = it.redop == MIN ? std::min(acc, v) : it.redop == MAX ? std::max(acc, v) : acc + v;
                }
                want = it.redop == AVG ? acc / size : acc;
                switch (it.kind) {
                    case 0: got = ((float*)it.buf.data())[i]; break;
                    case 1: got = ((double*)it.buf.data())[i]; break;
                    case 2: got = (double)((int64_t*)it.buf.data())[i]; break;
                    case 3: got = from_bf16(((uint16_t*)it.buf.data())[i]); break;
                    default: got = ((int32_t*)it.buf.data())[i]; break;
                }
                if (std::fabs(got - want) > 1e-5 * (1.0 + std::fabs(want))) bad = 5;
            } else if (it.kind == 5) {
                want = contrib(it.root, i, r);
                got = ((float*)it.buf.data())[i];
                if (got != want) bad = 6;
            } else {
                for (int q = 0; q < size && bad == 0; ++q) {
                    want = contrib(q, i, r);
                    got = ((int32_t*)it.out.data())[(size_t)q * it.count + i];
                    if (got != want) bad = 7;
                }
            }
            if (bad) fprintf(stderr, "[stress rank %d] item %d (kind %d op %d) word %lld: got %g want %g\n",
                             rank, k, it.kind, it.redop, i, got, want);
        }
    }
    if (bad == 0 && ddl_eng_wait(bar, 120.0) != 0) bad = fault_round >= 0 ? -1 : 8;
    stop.store(true);
    poller.join();
    control.join();
    ddl_eng_timeline_stop();
    const int warned = ddl_eng_stall_warnings();
    ddl_eng_shutdown();
    ::unlink(tl);
    if (fault_round >= 0) return bad == -1 ? 0 : 9;  // 9: the lost peer went unnoticed
    if (bad == 0) fprintf(stderr, "[stress rank %d] ok: %d collectives, %lld polls, %d stall warnings\n", rank, idx, polls.load(), warned);
    return bad;
}

int main(int argc, char** argv) {
    const int size = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    const double stall_s = argc > 3 ? atof(argv[3]) : 30.0;
    const int fault_round = argc > 4 ? atoi(argv[4]) : -1;
    if (size < 1 || size > 16 || rounds < 1) { fprintf(stderr, "usage: engine_stress [ranks 1..16] [rounds]\n"); return 64; }
    const int port = free_port();
    std::vector<pid_t> kids;
    for (int r = 0; r < size; ++r) {
        pid_t p = fork();
        if (p < 0) { perror("fork"); return 65; }
        if (p == 0) exit(run_rank(r, size, port, rounds, stall_s, fault_round));  // exit(), not _exit(): the sanitizer sets its status at exit
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) {
        int st = 0;
        ::waitpid(p, &st, 0);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        if (code != 0 && rc == 0) rc = code;
    }
    fprintf(stderr, "[stress] %s (ranks=%d rounds=%d fault_round=%d)\n", rc == 0 ? "PASS" : "FAIL", size, rounds, fault_round);
    return rc;
}
