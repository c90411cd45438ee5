// Small kernels that keep the training step free of framework (ATen) dispatches: every device
// operation between two optimizer updates is an in-tree kernel (profiles/r3_notes.md, "one code
// path").  What they replace, per ResNet-50 step: the loader's index draw + two gathers, the stem's
// filter repack (fill + strided copy) and gradient unpack (strided copy + accumulate), the padding
// of the head's logit gradient, the BatchNorm step counters' multi-tensor add.
#include "common.h"

namespace {

// Philox4x32-10, same constants as elementwise.hip (one counter -> 4 x u32)
__device__ __forceinline__ void px_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ uint32_t px_first(uint64_t counter, uint64_t seed) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        px_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// ---- synthetic loader: index draw + gather of images and labels in ONE launch -------------------
// state[0] = batches drawn so far (device-side, so that a recorded launch replays with fresh
// indices).  Sample b of the batch takes pool row philox(seed, state[0] * batch + b) mod pool_size -
// the reference's random translation index (HorovodPytorch/src/imagenet_pytorch_horovod.py:162-166,
// 177-188), drawn where it is used.  Grid (chunks, batch): every block copies one 16-byte-vector chunk
// of one image, eight vectors per thread in flight.  The counter is advanced by a one-thread kernel
// queued behind this one: an arrival ticket (one same-address device-scope atomic per block, 2560 of
// them) cost 150 of this kernel's 176 us (profiles/r3_notes.md).
__global__ void __launch_bounds__(256) pool_gather_kernel(const u32x4* __restrict__ pool, const long long* __restrict__ pool_labels,
                                                          u32x4* __restrict__ out, long long* __restrict__ out_labels,
                                                          const unsigned long long* __restrict__ state, int pool_size, int batch,
                                                          long long row_vec, uint64_t seed) {
    const int b = blockIdx.y;
    const unsigned long long step = state[0];
    const unsigned row = px_first(step * (unsigned long long)batch + (unsigned long long)b, seed) % (unsigned)pool_size;
    const long long per = (row_vec + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < row_vec ? lo + per : row_vec;
    const u32x4* src = pool + (long long)row * row_vec;
    u32x4* dst = out + (long long)b * row_vec;
    long long i = lo + threadIdx.x;
    for (; i + 7 * 256 < hi; i += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[i + u * 256] = v[u];
    }
    for (; i < hi; i += 256) dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && threadIdx.x == 0) out_labels[b] = pool_labels[row];
}
__global__ void pool_gather_advance_kernel(unsigned long long* __restrict__ state) { state[0] += 1; }

// ---- stem filter: [K][7][7][C<=4] (KRSC bf16) -> [K][7][8][4] zero-padded "row of taps" image ----
__global__ void __launch_bounds__(256) stem_weight_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int K, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // one output element
    if (i >= K * 7 * 8 * 4) return;
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % 7, k = i / (7 * 32);
    float v = 0.f;
    if (s < 7 && c < C) v = bf2f(w[((k * 7 + r) * 7 + s) * C + c]);
    wp[i] = f2bf(v);
}

// ---- stem filter gradient: fp32 [K][7][8][4] -> [K][7][7][C] (bf16 or fp32), optionally += -------
template <typename OutT>
__global__ void __launch_bounds__(256) stem_dw_unpack_kernel(const float* __restrict__ dwp, OutT* __restrict__ dw, int K, int C,
                                                             int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // one gradient element
    if (i >= K * 7 * 7 * C) return;
    const int c = i % C, s = (i / C) % 7, r = (i / (C * 7)) % 7, k = i / (C * 49);
    float v = dwp[((k * 7 + r) * 8 + s) * 4 + c];
    if (accumulate) v += (float)dw[i];
    dw[i] = (OutT)v;
}

// ---- BatchNorm step counters: table of int64 addresses, each += 1 --------------------------------
__global__ void __launch_bounds__(256) bump_counters_kernel(long long* const* __restrict__ table, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) *table[i] += 1;
}

// ---- every second row and column of an NHWC tensor (input of a stride-2 1x1 projection) ---------
__global__ void __launch_bounds__(256) subsample2_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, int H, int W, int Ho,
                                                         int Wo, int cvec, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one 16-byte vector of the output
    if (i >= total) return;
    const int c = (int)(i % cvec);
    long long p = i / cvec;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const long long b = p / Ho;
    out[i] = x[((b * H + 2 * ho) * W + 2 * wo) * cvec + c];
}

}  // namespace

extern "C" int ddl_subsample2(const void* x, void* out, int B, int H, int W, int C, hipStream_t stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return DDL_EINVAL;
    if ((C & 7) || !ddl_aligned16(x) || !ddl_aligned16(out)) return DDL_EALIGN;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, cvec = C / 8;
    const long long total = (long long)B * Ho * Wo * cvec;
    if ((total + 255) / 256 > 0x7fffffffLL) return DDL_EINVAL;
    DDL_KLAUNCH(subsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const u32x4*)x, (u32x4*)out, H, W,
                Ho, Wo, cvec, total);
    return ddl_launch_status();
}

extern "C" int ddl_pool_gather(const void* pool, const void* pool_labels, void* out, void* out_labels, void* state,
                               int pool_size, int batch, long long row_bytes, unsigned long long seed, hipStream_t stream) {
    if (!pool || !pool_labels || !out || !out_labels || !state || pool_size <= 0 || batch <= 0 || batch > 65535) return DDL_EINVAL;
    if (row_bytes <= 0 || (row_bytes & 15) || !ddl_aligned16(pool) || !ddl_aligned16(out)) return DDL_EALIGN;
    const long long row_vec = row_bytes >> 4;
    int chunks = (int)((row_vec + 2047) / 2048);          // ~32 KB per block
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    DDL_KLAUNCH(pool_gather_kernel, dim3(chunks, batch), dim3(256), 0, stream, (const u32x4*)pool,
                       (const long long*)pool_labels, (u32x4*)out, (long long*)out_labels, (const unsigned long long*)state,
                       pool_size, batch, row_vec, (uint64_t)seed);
    DDL_KLAUNCH(pool_gather_advance_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long*)state);
    return ddl_launch_status();
}

extern "C" int ddl_stem_weight_pack(const void* w, void* wp, int K, int C, hipStream_t stream) {
    if (!w || !wp || K <= 0 || C < 1 || C > 4) return DDL_EINVAL;
    DDL_KLAUNCH(stem_weight_pack_kernel, dim3((K * 7 * 8 * 4 + 255) / 256), dim3(256), 0, stream, (const bf16_t*)w,
                       (bf16_t*)wp, K, C);
    return ddl_launch_status();
}

extern "C" int ddl_stem_dw_unpack(const void* dwp, void* dw, int K, int C, int out_is_bf16, int accumulate, hipStream_t stream) {
    if (!dwp || !dw || K <= 0 || C < 1 || C > 4) return DDL_EINVAL;
    const dim3 grid((K * 49 * C + 255) / 256);
    if (out_is_bf16) DDL_KLAUNCH(stem_dw_unpack_kernel<bf16_t>, grid, dim3(256), 0, stream, (const float*)dwp, (bf16_t*)dw, K, C, accumulate);
    else DDL_KLAUNCH(stem_dw_unpack_kernel<float>, grid, dim3(256), 0, stream, (const float*)dwp, (float*)dw, K, C, accumulate);
    return ddl_launch_status();
}

extern "C" int ddl_bump_counters(const void* table, int n, hipStream_t stream) {
    if (!table || n <= 0) return DDL_EINVAL;
    DDL_KLAUNCH(bump_counters_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (long long* const*)table, n);
    return ddl_launch_status();
}

// K16: zero a gradient arena (one dispatch; the reference's optimizer.zero_grad() issues 161).  An own kernel rather
// than hipMemsetAsync: the runtime's fill arrives with ~15 us of idle queue on either side of it (two arenas at the head
// of every step: 58 us from the loader's gather to the stem's first kernel, ~16 us of it work - profiles/r3_notes.md
// section 35); unaligned spans still go through the runtime.
namespace {
__global__ __launch_bounds__(256) void fill16_kernel(u32x4* __restrict__ p, long long nvec, uint32_t word) {
    const u32x4 v = {word, word, word, word};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) p[i] = v;
}
}  // namespace

extern "C" int ddl_memset_async(void* p, int value, long long bytes, hipStream_t stream) {
    if (!p || bytes <= 0) return DDL_EINVAL;
    static const int own = getenv("DDL_FILL_KERNEL") ? atoi(getenv("DDL_FILL_KERNEL")) : 1;  // tuning knob
    if (own && ddl_aligned16(p) && (bytes & 15) == 0) {
        const long long nvec = bytes >> 4;
        long long blocks = (nvec + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        const uint32_t b = (uint32_t)(value & 0xff), word = b | (b << 8) | (b << 16) | (b << 24);
        DDL_KLAUNCH(fill16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (u32x4*)p, nvec, word);
        return ddl_launch_status();
    }
    if (hipMemsetAsync(p, value, (size_t)bytes, stream) != hipSuccess) return DDL_ELAUNCH;
    if (ddl::tape::recording()) ddl::tape::push([=]() { (void)hipMemsetAsync(p, value, (size_t)bytes, stream); });
    return DDL_OK;
}
