// Interface between the engine core (engine.cpp: queue, communication thread, watchdog, handles)
// and the device data plane (rccl_backend.cpp + xgmi_plane.cpp: RCCL over xGMI and the direct
// peer-access all-reduce kernel).  engine.cpp itself never includes a HIP header, so the host
// data plane and the ThreadSanitizer stress driver build without ROCm.
//
// Division of labour for one device collective:
//   submitting thread  record_ready(producer stream)  -> event that gates the collective
//   comm thread        launch(): wait(ready) on the engine's own high-priority stream, issue
//                      ncclAllReduce / the xGMI kernel there, record start/done events
//   consumer           wait_on(): the consumer's stream waits for `done` (the host never blocks)
//   watchdog           query(): completion, per-collective device time, timeout -> abort()
#pragma once
#include <cstdint>
#include <string>

namespace ddl {

enum Algo : int { ALGO_RCCL = 0, ALGO_TWOSHOT = 1, ALGO_ONESHOT = 2 };

struct DeviceOp {
    int kind = 0, dtype = 0, redop = 0, root = 0, algo = ALGO_RCCL;
    void* buf = nullptr;   // in/out (all-reduce, broadcast), input (all-gather)
    void* out = nullptr;   // all-gather output
    int64_t count = 0;
    void* ready = nullptr;  // event recorded on the producer stream by the submitting thread
};

struct DeviceTicket {
    void* start = nullptr;  // recorded on the engine stream right before the collective
    void* done = nullptr;   // ... and right after it
    int algo = ALGO_RCCL;   // what actually ran (an xGMI request falls back to RCCL for unregistered memory)
};

struct DevicePlane {
    bool (*bind_thread)();                                  // make the plane's device current on the calling thread
    void* (*record_ready)(void* producer_stream);           // nullptr on failure
    bool (*launch)(const DeviceOp& op, DeviceTicket* t, std::string* err);
    int (*query)(const DeviceTicket& t);                    // 1 complete, 0 still running, -1 error
    float (*elapsed_ms)(const DeviceTicket& t);             // device time start -> done (after completion)
    bool (*wait_on)(const DeviceTicket& t, void* consumer_stream);
    void (*recycle)(DeviceTicket* t);
    void (*abort)(const char* why);                         // tear down whatever a hung collective holds
    int (*kernel_timeout)();                                // 1 when the xGMI kernel reported an expired spin
};

// engine.cpp; nullptr detaches the plane (device ops are then refused)
void set_device_plane(const DevicePlane* plane);

}  // namespace ddl
