// Launch tape: storage and the C API (see tape.h).
#include "tape.h"

#include <mutex>
#include <vector>

#include "common.h"

namespace ddl {
namespace tape {

std::atomic<int> g_recording{0};

namespace {
struct Tape {
    std::vector<std::function<void()>> calls;
    std::vector<size_t> seg_end;      // seg_end[k] = index one past segment k's last call
};
std::mutex g_mu;
Tape* g_open = nullptr;               // the recording in progress
std::vector<Tape*> g_tapes;           // finished tapes by id (nullptr = freed)
bool g_paused = false;
}  // namespace

void push(std::function<void()> fn) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open && !g_paused) g_open->calls.emplace_back(std::move(fn));
}

}  // namespace tape
}  // namespace ddl

using namespace ddl::tape;

extern "C" {

// 0 on success, -1 when a recording is already open
int ddl_tape_begin() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open) return -1;
    g_open = new Tape();
    g_paused = false;
    g_recording.store(1);
    return 0;
}

// close the current segment; returns the number of segments closed so far (= index of the next one)
int ddl_tape_mark() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open) return -1;
    g_open->seg_end.push_back(g_open->calls.size());
    return (int)g_open->seg_end.size();
}

// launches between pause and resume are executed but not recorded (live actions re-run at replay)
int ddl_tape_pause() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open) return -1;
    g_paused = true;
    g_recording.store(0);
    return 0;
}
int ddl_tape_resume() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open) return -1;
    g_paused = false;
    g_recording.store(1);
    return 0;
}

// finish the recording (closing the last segment); returns the tape id, -1 without a recording
int ddl_tape_end() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_open) return -1;
    g_recording.store(0);
    if (g_open->seg_end.empty() || g_open->seg_end.back() != g_open->calls.size()) g_open->seg_end.push_back(g_open->calls.size());
    g_tapes.push_back(g_open);
    g_open = nullptr;
    return (int)g_tapes.size() - 1;
}

// drop an open recording (an exception interrupted the recorded step)
void ddl_tape_abort() {
    std::lock_guard<std::mutex> lk(g_mu);
    g_recording.store(0);
    delete g_open;
    g_open = nullptr;
}

int ddl_tape_segments(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) return -1;
    return (int)g_tapes[id]->seg_end.size();
}
long long ddl_tape_calls(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) return -1;
    return (long long)g_tapes[id]->calls.size();
}

// re-issue segments [seg_lo, seg_hi) of tape `id`; DDL_OK, or DDL_ELAUNCH when the runtime reports a launch error
int ddl_tape_replay(int id, int seg_lo, int seg_hi) {
    Tape* t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (id < 0 || id >= (int)g_tapes.size() || !g_tapes[id]) return DDL_EINVAL;
        if (g_open) return DDL_EINVAL;      // never replay into a recording
        t = g_tapes[id];
    }
    const int nseg = (int)t->seg_end.size();
    if (seg_lo < 0 || seg_hi > nseg || seg_lo > seg_hi) return DDL_EINVAL;
    size_t i = seg_lo == 0 ? 0 : t->seg_end[seg_lo - 1];
    const size_t end = seg_hi == 0 ? 0 : t->seg_end[seg_hi - 1];
    for (; i < end; ++i) t->calls[i]();
    return ddl_launch_status();
}

void ddl_tape_free(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 0 || id >= (int)g_tapes.size()) return;
    delete g_tapes[id];
    g_tapes[id] = nullptr;
}

}  // extern "C"
