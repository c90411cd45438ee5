// Host side of the direct xGMI all-reduce (kernel K18, csrc/kernels/xgmi_allreduce.hip; SURVEY
// §5.8): hipIpc peer mappings of the gradient arenas and of one flag page per rank, exchanged
// through the engine's TCP store, and the launch of the kernel on the engine's stream.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <string>

namespace ddl {
namespace xgmi {

// Collective (every rank): allocates this rank's fine-grained flag page and host-visible timeout
// word, publishes the page's IPC handle and maps every peer's page.
bool init(int rank, int size, int device, void* kernel_fn, int blocks, long long spin_limit, std::string* err);
// Collective, same order on every rank: export [ptr, ptr+bytes) and map every peer's counterpart.
// Returns the region id (>= 0) or -1.
int register_region(void* ptr, size_t bytes, std::string* err);
// true when [buf, buf+bytes) lies in a registered region and the kernel's shape rules hold
bool usable(const void* buf, size_t bytes, long long count, int nranks);
bool launch(void* buf, long long count, bool bf16, bool one_shot, float scale, hipStream_t stream, std::string* err);
int timeout_flag();   // 1 once any launch of this process reported an expired spin
void raise_abort();   // host-side: make every spinning kernel of this process give up (bounded by 4096 polls)
std::string describe();  // flag-page snapshot for the watchdog's timeout report
bool active();
void shutdown();

}  // namespace xgmi
}  // namespace ddl
