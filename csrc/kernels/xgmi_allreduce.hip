// K18 / K17: direct-access all-reduce over peer-mapped buffers, and multi-tensor pack/unpack.
//
// xGMI on an 8-GPU MI355X node is a full mesh of point-to-point links (7 x ~153 GB/s per GPU).
// A ring uses one link per GPU; the two-shot direct algorithm uses all seven at once
// (SURVEY §5.8): rank r owns shard r of the bucket;
//   phase 1  r reads shard r from every peer's buffer (7 links in, concurrently), reduces in fp32;
//   phase 2  r writes the reduced (and averaged) shard into every peer's buffer (7 links out).
// Per-link traffic is 2 x S/8 instead of the ring's 2 x 7/8 S.  One-shot (every rank reads every
// buffer whole and keeps the sum) serves latency-bound buckets.
//
// Synchronisation is by epoch-stamped flags in each rank's own flag page (peers push, the owner
// polls locally): system-scope release before every flag store, system-scope acquire after every
// successful poll.  Every spin is bounded; on expiry the kernel raises `timeout_word` and
// returns (the host turns that into an error) — it can never hang the device.  The grid is
// small (default 16 workgroups) so it can overlap with backward convolutions without starving
// them, and every workgroup of the grid is resident by construction (16 << 256 CUs).
//
// Status: every word of every result has been checked between 2 and 4 REAL processes over REAL
// hipIpc mappings (tests/test_gpu_engine.py), all of them on ONE GPU - the only hardware the
// development boxes have.  What one device cannot show is cross-device visibility of the payload
// (arenas are ordinary coarse-grained allocator memory; the kernel reads and writes them with
// sc0 sc1 accesses and the flags live in fine-grained memory) and the achieved link bandwidth.
// The algorithm is therefore opt-in (HVD_ALLREDUCE_ALGO=twoshot|oneshot, or auto with a table
// MEASURED on the node by benchmarks/allreduce_bw.py), every job that enables it runs a checked
// all-reduce through it at init (parallel/backend.py::_selftest) and a replica checksum after
// the first optimizer steps (parallel/optimizer.py), and RCCL stays the default data plane.
#include "common.h"

namespace {

constexpr int MAX_RANKS = 8;

struct XgmiParams {
    void* bufs[MAX_RANKS];        // peer-mapped data buffers (index = rank)
    unsigned* flags[MAX_RANKS];   // peer-mapped flag pages: [0..7] ready, [8..15] done, [16] block ticket
    void* outs[MAX_RANKS];        // one-shot: output buffer of each rank (only [rank] is used outside simulation)
    int simulate;                 // test mode: ONE launch plays every rank, rank = blockIdx.y
    long long count;              // elements, multiple of 8 * nranks
    int rank, nranks, is_bf16, one_shot;
    unsigned epoch;               // > 0, strictly increasing per call
    float scale;                  // 1/nranks for averaging, 1 for sum
    long long spin_limit;
    unsigned* timeout_word;
};

// `abort_word` is the host-mapped timeout word: another workgroup that gave up, or the host's
// watchdog when it declares the collective dead, raises it - checked every 4096 polls (a read
// over PCIe) so that a long spin budget (it follows HVD_COLLECTIVE_TIMEOUT_S) never outlives an abort.
__device__ __forceinline__ bool wait_flag(unsigned* p, unsigned epoch, long long limit, const unsigned* abort_word) {
    for (long long i = 0; i < limit; ++i) {
        unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - epoch) >= 0) return true;
        if ((i & 4095) == 4095 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

// All payload traffic is system-coherent (sc0 sc1 = aux 17): loads miss every cache level that a
// peer's store cannot reach, stores are written through - the data path does not depend on what a
// fence does to lines the L2 believes to be local.  8 elements per lane and access.
constexpr int AUX_SYS = 17;

template <bool BF16>
__device__ __forceinline__ void load8(__amdgpu_buffer_rsrc_t rs, long long i8, float* v) {
    if (BF16) {
        u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i8 * 16), 0, AUX_SYS);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, x[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, x[e] & 0xffff0000u);
        }
    } else {
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i8 * 32), 0, AUX_SYS));
        const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i8 * 32 + 16), 0, AUX_SYS));
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
}
template <bool BF16>
__device__ __forceinline__ void store8(__amdgpu_buffer_rsrc_t rs, long long i8, const float* v) {
    if (BF16) {
        u32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x2 pr = {f2bf(v[2 * e]), f2bf(v[2 * e + 1])};
            x[e] = __builtin_bit_cast(unsigned, pr);
        }
        __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)(i8 * 16), 0, AUX_SYS);
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rs, (int)(i8 * 32), 0, AUX_SYS);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rs, (int)(i8 * 32 + 16), 0, AUX_SYS);
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256) xgmi_allreduce_kernel(const XgmiParams p) {
    __shared__ int ok_s;
    const int n = p.nranks;
    const int r = p.simulate ? (int)blockIdx.y : p.rank;
    // 1. announce "my buffer holds this step's input" to every rank (including myself)
    if (blockIdx.x == 0 && (int)threadIdx.x < n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
        __hip_atomic_store(p.flags[threadIdx.x] + r, p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. wait until every rank has announced (poll my own flag page)
    if (threadIdx.x == 0) {
        bool ok = true;
        int q = 0;
        for (; q < n && ok; ++q) ok = wait_flag(p.flags[r] + q, p.epoch, p.spin_limit, p.timeout_word);
        // timeout word = 0x100 | (peer + 1): the peer whose "ready" never arrived
        if (!ok && __hip_atomic_load(p.timeout_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
            __hip_atomic_store(p.timeout_word, 0x100u | (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    if (!ok_s) return;  // uniform per block; peers time out on their own bounded spins
    // Acquire AFTER the poll has matched, by every wave, and waited for: the invalidate completes
    // asynchronously, a load issued before it has finished can still hit a stale line (the peers'
    // buffers have the same addresses every step and this kernel may have been spinning since
    // before they were written).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const long long total8 = p.count >> 3;
    const long long shard8 = p.one_shot ? total8 : total8 / n;
    const long long base8 = p.one_shot ? 0 : shard8 * r;
    const int nbytes = (int)(p.count * (BF16 ? 2 : 4));   // < 2^31, checked by the launcher
    __amdgpu_buffer_rsrc_t rs[MAX_RANKS];
#pragma unroll
    for (int q = 0; q < MAX_RANKS; ++q)
        rs[q] = __builtin_amdgcn_make_buffer_rsrc(q < n ? p.bufs[q] : p.bufs[0], 0, q < n ? nbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.one_shot ? p.outs[r] : p.bufs[0], 0, nbytes, 0x00020000);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < shard8; i += stride) {
        // fixed summation order (rank 0, 1, ...): every rank computes bit-identical sums
        float acc[8], v[MAX_RANKS][8];
#pragma unroll
        for (int q = 0; q < MAX_RANKS; ++q)
            if (q < n) load8<BF16>(rs[q], base8 + i, v[q]);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = v[0][e];
#pragma unroll
        for (int q = 1; q < MAX_RANKS; ++q)
            if (q < n) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[q][e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= p.scale;
        if (p.one_shot) {
            store8<BF16>(ro, i, acc);
        } else {
#pragma unroll
            for (int q = 0; q < MAX_RANKS; ++q)
                if (q < n) store8<BF16>(rs[q], base8 + i, acc);
        }
    }
    // 3. publish: every writing wave drains its stores, the last workgroup signals "done"
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = p.flags[r] + 16;
        unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        ok_s = (t == gridDim.x - 1) ? 1 : 0;
        if (ok_s) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!ok_s) return;
    if ((int)threadIdx.x < n)
        __hip_atomic_store(p.flags[threadIdx.x] + 8 + r, p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 4. the kernel may only complete once every rank's writes into MY buffer have landed
    //    (two-shot) / every rank has finished reading my buffer (one-shot)
    if (threadIdx.x == 0) {
        bool ok = true;
        int q = 0;
        for (; q < n && ok; ++q) ok = wait_flag(p.flags[r] + 8 + q, p.epoch, p.spin_limit, p.timeout_word);
        // 0x200 | (peer + 1): the peer whose "done" never arrived
        if (!ok && __hip_atomic_load(p.timeout_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
            __hip_atomic_store(p.timeout_word, 0x200u | (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
}

// ---- K17: multi-tensor pack (gather, scale, cast) and unpack ---------------------------------
struct PackEntry { long long src; long long dst; long long numel; };  // byte address, element offset, elements

template <typename SrcT, typename DstT, bool UNPACK>
__global__ void __launch_bounds__(256) pack_kernel(const PackEntry* __restrict__ table, int nentries, void* flat, float scale) {
    for (int k = blockIdx.y; k < nentries; k += gridDim.y) {
        const PackEntry e = table[k];
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e.numel; i += stride) {
            if (UNPACK) {
                float v = (float)((const DstT*)flat)[e.dst + i] * scale;
                ((SrcT*)e.src)[i] = (SrcT)v;
            } else {
                float v = (float)((const SrcT*)e.src)[i] * scale;
                ((DstT*)flat)[e.dst + i] = (DstT)v;
            }
        }
    }
}

}  // namespace

extern "C" int ddl_xgmi_allreduce(void* const* bufs, void* const* flags, void* const* outs, long long count, int rank, int nranks,
                                  int is_bf16, int one_shot, unsigned epoch, float scale, long long spin_limit,
                                  void* timeout_word, int blocks, int simulate, hipStream_t stream) {
    if (!bufs || !flags || !timeout_word || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return DDL_EINVAL;
    if (count <= 0 || (count % (8LL * nranks)) != 0 || epoch == 0 || spin_limit <= 0) return DDL_EINVAL;
    if (count * (is_bf16 ? 2 : 4) >= 0x7fffffffLL) return DDL_EINVAL;  // 32-bit buffer offsets
    if (one_shot && !outs) return DDL_EINVAL;
    if (blocks < 1 || blocks > 64) return DDL_EINVAL;  // small by design: always fully resident
    XgmiParams p;
    for (int i = 0; i < MAX_RANKS; ++i) { p.bufs[i] = nullptr; p.flags[i] = nullptr; p.outs[i] = nullptr; }
    for (int i = 0; i < nranks; ++i) {
        if (!bufs[i] || !flags[i] || !ddl_aligned16(bufs[i])) return DDL_EALIGN;
        p.bufs[i] = bufs[i]; p.flags[i] = (unsigned*)flags[i];
        if (one_shot && (simulate || i == rank)) {
            if (!outs[i] || !ddl_aligned16(outs[i])) return DDL_EALIGN;
            p.outs[i] = outs[i];
        }
    }
    p.simulate = simulate; p.count = count; p.rank = rank; p.nranks = nranks; p.is_bf16 = is_bf16; p.one_shot = one_shot;
    p.epoch = epoch; p.scale = scale; p.spin_limit = spin_limit; p.timeout_word = (unsigned*)timeout_word;
    if (is_bf16) hipLaunchKernelGGL(xgmi_allreduce_kernel<true>, dim3(blocks, simulate ? nranks : 1), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(xgmi_allreduce_kernel<false>, dim3(blocks, simulate ? nranks : 1), dim3(256), 0, stream, p);
    return ddl_launch_status();
}

// table: device array of {src byte address, dst element offset, numel}; src dtype fp32 or bf16,
// flat dtype fp32 or bf16; unpack writes flat*scale back into the sources.
extern "C" int ddl_pack_scale_cast(const void* table, int nentries, void* flat, int src_is_bf16, int flat_is_bf16,
                                   float scale, int unpack, hipStream_t stream) {
    if (!table || !flat || nentries <= 0) return DDL_EINVAL;
    dim3 grid(64, nentries < 256 ? nentries : 256), block(256);
#define PK(S, D, U) DDL_KLAUNCH((pack_kernel<S, D, U>), grid, block, 0, stream, (const PackEntry*)table, nentries, flat, scale)
    if (src_is_bf16) {
        if (flat_is_bf16) { if (unpack) PK(bf16_t, bf16_t, true); else PK(bf16_t, bf16_t, false); }
        else { if (unpack) PK(bf16_t, float, true); else PK(bf16_t, float, false); }
    } else {
        if (flat_is_bf16) { if (unpack) PK(float, bf16_t, true); else PK(float, bf16_t, false); }
        else { if (unpack) PK(float, float, true); else PK(float, float, false); }
    }
#undef PK
    return ddl_launch_status();
}
