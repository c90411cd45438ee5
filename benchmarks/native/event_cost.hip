// Host cost of ordering one stream behind another: hipEventRecord + hipStreamWaitEvent against
// hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory.  Build:
//   hipcc -O2 --offload-arch=gfx950 benchmarks/native/event_cost.hip -o benchmarks/native/event_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) p[0] += 1; }

int main() {
    hipStream_t a, b;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int* scratch;
    CK(hipMalloc(&scratch, 64));
    CK(hipMemset(scratch, 0, 64));
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, scratch);
            CK(hipEventRecord(ev, a));
            CK(hipStreamWaitEvent(b, ev, 0));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, scratch + 8);
        }
        auto t1 = now();
        CK(hipDeviceSynchronize());
        auto t2 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, scratch);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, scratch + 8);
        }
        auto t3 = now();
        CK(hipDeviceSynchronize());
        printf("event pair: %.2f us per iteration with, %.2f us without -> %.2f us per pair\n", us(t1 - t0) / N, us(t3 - t2) / N,
               (us(t1 - t0) - us(t3 - t2)) / N);
    }
    unsigned* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    if (e != hipSuccess) { printf("hipMallocSignalMemory: %s\n", hipGetErrorString(e)); return 0; }
    CK(hipMemset(sig, 0, 8));
    CK(hipDeviceSynchronize());
    unsigned v = 0;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, scratch);
            ++v;
            e = hipStreamWriteValue32(a, sig, v, 0);
            if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); return 0; }
            e = hipStreamWaitValue32(b, sig, v, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 0; }
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, scratch + 8);
        }
        auto t1 = now();
        CK(hipDeviceSynchronize());
        printf("value pair: %.2f us per iteration (incl. the two launches)\n", us(t1 - t0) / N);
    }
    int h[16];
    CK(hipMemcpy(h, scratch, 64, hipMemcpyDeviceToHost));
    printf("counters %d %d (expected %d each)\n", h[0], h[8], 6 * N);
    return 0;
}
