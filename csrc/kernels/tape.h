// Launch tape: record the device work of one training step once, replay it from native code.
//
// The reference issues every device operation of its hot loop from Python, every iteration
// (HorovodPytorch/src/imagenet_pytorch_horovod.py:210-218); so does this framework's eager path:
// ~600 launches through Python + ctypes per ResNet-50 step, 11 of its 18 ms in host time, 28 of 44 ms
// for ResNet-152 (profiles/r2_notes.md section 13).  With static shapes the sequence never changes.
// A hipGraph would be the textbook answer, but ROCm 7.2 replays a graph with ~50 cross-stream edges
// at ~25 us per node (29 ms for the 18 ms step, training/graph.py), and a one-stream graph loses
// the overlap of the weight-gradient stream.  The tape keeps the streams: every kernel launch of
// the library goes through DDL_KLAUNCH, which - while a recording is open - also stores a closure
// (kernel pointer, launch geometry, stream, arguments by value).  Stream-ordering calls and memsets
// record themselves the same way.  Replaying a tape re-issues exactly the recorded calls on the
// recorded streams, in the recorded order, from one native loop: no Python, no ctypes, no argument
// checks, no allocator - about 2 us of host time per launch instead of ~15.
//
// Python actions that cannot be recorded (hand-offs to the communication engine) cut the tape into
// SEGMENTS: ddl_tape_mark() closes the current segment, the caller replays segment by segment and
// runs the live action in between (training/tape.py).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

namespace ddl {
namespace tape {

extern std::atomic<int> g_recording;          // > 0 while a recording is open and not paused
void push(std::function<void()> fn);          // appends to the open recording (thread-safe)

inline bool recording() { return g_recording.load(std::memory_order_relaxed) > 0; }

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    std::tuple<std::decay_t<KArgs>...> tup(static_cast<std::decay_t<KArgs>>(args)...);
    std::apply([&](auto&... a) { kernel<<<grid, block, shmem, stream>>>(a...); }, tup);
    if (recording())
        push([=]() mutable { std::apply([&](auto&... a) { kernel<<<grid, block, shmem, stream>>>(a...); }, tup); });
}

}  // namespace tape
}  // namespace ddl

#define DDL_KLAUNCH(kernel, grid, block, shmem, stream, ...) ::ddl::tape::launch(kernel, grid, block, shmem, stream, __VA_ARGS__)
