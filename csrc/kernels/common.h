// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes; all vector accesses are 16 B per lane where alignment allows.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#include "tape.h"

#define DDL_OK 0
#define DDL_EINVAL (-1)
#define DDL_EALIGN (-2)
#define DDL_ELAUNCH (-3)

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE, NaN-preserving

__device__ __forceinline__ void load8f(const float* p, float* v) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}

// The one BN affine used by forward and by the backward's mask recomputation: identical
// rounding on both sides, so "z > 0" can be re-derived from y without reading z.
__device__ __forceinline__ float bn_affine(float v, float sc, float sh) { return __builtin_fmaf(v, sc, sh); }

// The one BatchNorm-backward apply dy = ca*g + cb*y + cc: explicit operation order, so that the kernels which fold
// the apply into their own input staging (conv_rows.hip, conv_wgrad.hip) round exactly like the separate pass.
__device__ __forceinline__ float bn_bwd_affine(float g, float y, float ca, float cb, float cc) {
    return __builtin_fmaf(ca, g, __builtin_fmaf(cb, y, cc));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One 16-byte-per-lane LDS-DMA (`buffer_load_dwordx4 ... lds`) issued from inline asm, i.e. invisible to
// hipcc's wait-count insertion: the compiler orders LDS accesses it considers stores (the transposing
// `ds_read_b64_tr_b16` builtin among them) behind ALL LDS-DMA it knows to be in flight with `s_waitcnt vmcnt(0)`,
// which empties a multi-stage ring; hidden, the DMA is ordered by the kernel's own counted waits.
// `lds_addr` (byte address of the wave's 1 KiB piece) and `soff` are wave-uniform; one wait state separates
// the write of M0 from its use.  `rsrc`: raw buffer descriptor (raw_rsrc).
__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // (already uniform: tells the compiler so)
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

static inline int ddl_launch_status_at(const char* file, int line) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DDL_OK;
    fprintf(stderr, "[ddl] kernel launch failed at %s:%d: %s\n", file, line, hipGetErrorString(e));
    return DDL_ELAUNCH;
}
#define ddl_launch_status() ddl_launch_status_at(__FILE__, __LINE__)
static inline bool ddl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Grid sizing for memory-bound grid-stride kernels.  Measured with benchmarks/native/stream_bench
// (BN-apply shaped kernel, 411 MB in + 411 MB out): 2048 blocks 5.08 TB/s, 16384 5.43, 65536 5.90 —
// many short-lived blocks beat few long grid-stride loops on 256 CUs (tensors that fit the
// 256 MB infinity cache run at ~7.8 TB/s for any grid), so the cap is 65536 blocks.
static inline int ddl_stream_grid(long long work_items, int block) {
    static const int cap = getenv("DDL_STREAM_GRID_CAP") ? atoi(getenv("DDL_STREAM_GRID_CAP")) : 65536;  // tuning knob
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
