// Memory-bound support kernels: K19 Philox synthetic data, K14 fused
// SGD-momentum-weight-decay over flat arenas, K15 fused LARS (segmented norms + update),
// K16 is one fill launch per gradient arena (ddl_memset_async in step_aux.hip), K20 NCHW fp32 -> NHWC bf16.
// All kernels: wave64, 256-thread blocks, 16-byte vector accesses, grid-stride.
#include "common.h"

// ---------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al.); one counter -> 4 x u32.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ u32x4 philox4x32(uint64_t counter, uint64_t seed) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 out = {c0, c1, c2, c3};
    return out;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__global__ void __launch_bounds__(256) philox_uniform_bf16_kernel(bf16_t* __restrict__ out, long long n, uint64_t seed) {
    long long n8 = n >> 3;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        u32x4 a = philox4x32(2 * (uint64_t)i, seed), b = philox4x32(2 * (uint64_t)i + 1, seed);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float fa = fminf(u32_to_unit(a[j]), 0.99609375f), fb = fminf(u32_to_unit(b[j]), 0.99609375f);
            v[j] = f2bf(fa); v[4 + j] = f2bf(fb);
        }
        *reinterpret_cast<bf16x8*>(out + (i << 3)) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        long long i = (n8 << 3) + threadIdx.x;
        u32x4 a = philox4x32(2 * (uint64_t)n8 + 4096 + threadIdx.x, seed ^ 0x5bd1e995u);
        out[i] = f2bf(fminf(u32_to_unit(a[0]), 0.99609375f));
    }
}
__global__ void __launch_bounds__(256) philox_uniform_f32_kernel(float* __restrict__ out, long long n, uint64_t seed) {
    long long n4 = n >> 2;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        u32x4 a = philox4x32((uint64_t)i, seed);
        f32x4 v = {u32_to_unit(a[0]), u32_to_unit(a[1]), u32_to_unit(a[2]), u32_to_unit(a[3])};
        *reinterpret_cast<f32x4*>(out + (i << 2)) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        u32x4 a = philox4x32((uint64_t)n4 + 4096 + threadIdx.x, seed ^ 0x5bd1e995u);
        out[(n4 << 2) + threadIdx.x] = u32_to_unit(a[0]);
    }
}
__global__ void __launch_bounds__(256) philox_labels_kernel(long long* __restrict__ out, long long n, int n_classes, uint64_t seed) {
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u32x4 a = philox4x32((uint64_t)i, seed);
        out[i] = (long long)(((uint64_t)a[0] * (uint64_t)n_classes) >> 32);  // unbiased to 2^-32
    }
}

extern "C" int ddl_philox_uniform(void* out, long long n, int kind, uint64_t seed, hipStream_t stream) {
    if (n < 0 || !out) return DDL_EINVAL;
    if (!ddl_aligned16(out)) return DDL_EALIGN;
    if (n == 0) return DDL_OK;
    if (kind == 0) {
        int grid = ddl_stream_grid((n + 7) / 8, 256);
        DDL_KLAUNCH(philox_uniform_bf16_kernel, dim3(grid), dim3(256), 0, stream, (bf16_t*)out, n, seed);
    } else if (kind == 1) {
        int grid = ddl_stream_grid((n + 3) / 4, 256);
        DDL_KLAUNCH(philox_uniform_f32_kernel, dim3(grid), dim3(256), 0, stream, (float*)out, n, seed);
    } else {
        return DDL_EINVAL;
    }
    return ddl_launch_status();
}
extern "C" int ddl_philox_labels(void* out, long long n, int n_classes, uint64_t seed, hipStream_t stream) {
    if (n < 0 || !out || n_classes <= 0) return DDL_EINVAL;
    if (n == 0) return DDL_OK;
    DDL_KLAUNCH(philox_labels_kernel, dim3(ddl_stream_grid(n, 256)), dim3(256), 0, stream,
                       (long long*)out, n, n_classes, seed);
    return ddl_launch_status();
}

// ---------------------------------------------------------------------------------------
// K14: fused SGD.  g' = g + wd*w ; v = mu*v + g' ; w -= lr*(nesterov ? g' + mu*v : v)
// w, v fp32 (masters); g fp32 or bf16; optional bf16 shadow copy of w written in the same pass.
// ---------------------------------------------------------------------------------------
template <bool GRAD_BF16, bool HAS_LOWP>
__global__ void __launch_bounds__(256) fused_sgd_kernel(float* __restrict__ w, const void* __restrict__ gv,
                                                       float* __restrict__ v, bf16_t* __restrict__ lowp,
                                                       long long n, float lr, float mu, float wd, int nesterov,
                                                       const float* __restrict__ lr_scale) {
    // optional device-side multiplier: a captured HIP graph keeps `lr` by value, the schedule
    // then moves the learning rate by rewriting one float on the device
    if (lr_scale) lr *= *lr_scale;
    long long n8 = n >> 3;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        long long e = i << 3;
        f32x4 w0 = *reinterpret_cast<f32x4*>(w + e), w1 = *reinterpret_cast<f32x4*>(w + e + 4);
        f32x4 v0 = *reinterpret_cast<f32x4*>(v + e), v1 = *reinterpret_cast<f32x4*>(v + e + 4);
        float g[8];
        if (GRAD_BF16) {
            bf16x8 gb = *reinterpret_cast<const bf16x8*>((const bf16_t*)gv + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = bf2f(gb[j]);
        } else {
            f32x4 g0 = *reinterpret_cast<const f32x4*>((const float*)gv + e);
            f32x4 g1 = *reinterpret_cast<const f32x4*>((const float*)gv + e + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = g0[j]; g[4 + j] = g1[j]; }
        }
        bf16x8 lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float wj = j < 4 ? w0[j] : w1[j - 4];
            float vj = j < 4 ? v0[j] : v1[j - 4];
            float gj = g[j] + wd * wj;
            vj = mu * vj + gj;
            wj -= lr * (nesterov ? gj + mu * vj : vj);
            if (j < 4) { w0[j] = wj; v0[j] = vj; } else { w1[j - 4] = wj; v1[j - 4] = vj; }
            lo[j] = f2bf(wj);
        }
        *reinterpret_cast<f32x4*>(w + e) = w0; *reinterpret_cast<f32x4*>(w + e + 4) = w1;
        *reinterpret_cast<f32x4*>(v + e) = v0; *reinterpret_cast<f32x4*>(v + e + 4) = v1;
        if (HAS_LOWP) *reinterpret_cast<bf16x8*>(lowp + e) = lo;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        long long e = (n8 << 3) + threadIdx.x;
        float gj = (GRAD_BF16 ? bf2f(((const bf16_t*)gv)[e]) : ((const float*)gv)[e]) + wd * w[e];
        float vj = mu * v[e] + gj;
        float wj = w[e] - lr * (nesterov ? gj + mu * vj : vj);
        w[e] = wj; v[e] = vj;
        if (HAS_LOWP) lowp[e] = f2bf(wj);
    }
}

extern "C" int ddl_fused_sgd_ex(void* w, const void* g, void* v, void* lowp, long long n, int grad_is_bf16,
                                float lr, float mu, float wd, int nesterov, const void* lr_scale, hipStream_t stream) {
    if (!w || !g || !v || n < 0) return DDL_EINVAL;
    if (n == 0) return DDL_OK;
    if (!ddl_aligned16(w) || !ddl_aligned16(g) || !ddl_aligned16(v) || (lowp && !ddl_aligned16(lowp))) return DDL_EALIGN;
    dim3 grid(ddl_stream_grid((n + 7) / 8, 256)), block(256);
#define LAUNCH(GB, LP) DDL_KLAUNCH((fused_sgd_kernel<GB, LP>), grid, block, 0, stream, (float*)w, g, (float*)v, (bf16_t*)lowp, n, lr, mu, wd, nesterov, (const float*)lr_scale)
    if (grad_is_bf16) { if (lowp) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (lowp) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    return ddl_launch_status();
}

extern "C" int ddl_fused_sgd(void* w, const void* g, void* v, void* lowp, long long n, int grad_is_bf16,
                             float lr, float mu, float wd, int nesterov, hipStream_t stream) {
    return ddl_fused_sgd_ex(w, g, v, lowp, n, grad_is_bf16, lr, mu, wd, nesterov, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// K15: LARS.  Pass 1: per-segment sum(w^2), sum(g^2), chunked and DETERMINISTIC (no float atomics: a segment
// that lies inside one chunk is summed by that chunk's block and stored; a segment that crosses chunk borders
// gets one partial per chunk - a chunk has at most one segment entering from the left and one leaving to the
// right, two slots - and a second small kernel adds a segment's partials in chunk order).
// Pass 2: v = mu*v + lr*ratio[seg]*(g + wd*w); w -= v.
// Segments are [seg_start[k], seg_end[k]) element ranges relative to the run base; padding
// between segments holds zeros and is never touched.
// ---------------------------------------------------------------------------------------
#define LARS_CHUNK 8192

__device__ __forceinline__ int find_segment(const long long* __restrict__ seg_start, int nseg, long long pos) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (seg_start[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <bool GRAD_BF16>
__global__ void __launch_bounds__(256) segmented_sqnorm_kernel(const float* __restrict__ w, const void* __restrict__ gv,
                                                              const long long* __restrict__ seg_start,
                                                              const long long* __restrict__ seg_end, int nseg,
                                                              long long n, float* __restrict__ out /*[2][nseg]*/,
                                                              float* __restrict__ part /*[chunks][2 slots][2]*/) {
    __shared__ float red[2][4];
    long long c0 = (long long)blockIdx.x * LARS_CHUNK;
    long long c1 = c0 + LARS_CHUNK < n ? c0 + LARS_CHUNK : n;
    int k = find_segment(seg_start, nseg, c0);
    for (; k < nseg && seg_start[k] < c1; ++k) {
        long long a = seg_start[k] > c0 ? seg_start[k] : c0;
        long long b = seg_end[k] < c1 ? seg_end[k] : c1;
        float sw = 0.f, sg = 0.f;
        for (long long i = a + threadIdx.x; i < b; i += 256) {
            float wi = w[i];
            float gi = GRAD_BF16 ? bf2f(((const bf16_t*)gv)[i]) : ((const float*)gv)[i];
            sw += wi * wi; sg += gi * gi;
        }
        sw = wave_sum(sw); sg = wave_sum(sg);
        int wid = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][wid] = sw; red[1][wid] = sg; }
        __syncthreads();
        if (threadIdx.x == 0 && b > a) {
            const float tw = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            const float tg = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            if (seg_start[k] >= c0 && seg_end[k] <= c1) {            // the whole segment: final
                out[k] = tw; out[nseg + k] = tg;
            } else {
                float* slot = part + ((long long)blockIdx.x * 2 + (seg_start[k] < c0 ? 0 : 1)) * 2;
                slot[0] = tw; slot[1] = tg;
            }
        }
        __syncthreads();
    }
}

// segments that cross chunk borders: their partials in chunk order (the first chunk's right slot, then left slots);
// empty segments: zero
__global__ void __launch_bounds__(256) segmented_sqnorm_join_kernel(const long long* __restrict__ seg_start,
                                                                   const long long* __restrict__ seg_end, int nseg,
                                                                   const float* __restrict__ part, float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nseg) return;
    const long long a = seg_start[k], b = seg_end[k];
    if (b <= a) { out[k] = 0.f; out[nseg + k] = 0.f; return; }
    const long long first = a / LARS_CHUNK, last = (b - 1) / LARS_CHUNK;
    if (first == last) return;                                     // written by the chunk's block
    float tw = part[(first * 2 + 1) * 2], tg = part[(first * 2 + 1) * 2 + 1];
    for (long long c = first + 1; c <= last; ++c) { tw += part[(c * 2) * 2]; tg += part[(c * 2) * 2 + 1]; }
    out[k] = tw; out[nseg + k] = tg;
}

template <bool GRAD_BF16, bool HAS_LOWP>
__global__ void __launch_bounds__(256) fused_lars_kernel(float* __restrict__ w, const void* __restrict__ gv, float* __restrict__ v,
                                                        bf16_t* __restrict__ lowp, const float* __restrict__ ratio,
                                                        const long long* __restrict__ seg_start,
                                                        const long long* __restrict__ seg_end, int nseg, long long n,
                                                        float lr, float mu, float wd,
                                                        const float* __restrict__ lr_scale) {
    if (lr_scale) lr *= *lr_scale;
    long long c0 = (long long)blockIdx.x * LARS_CHUNK;
    long long c1 = c0 + LARS_CHUNK < n ? c0 + LARS_CHUNK : n;
    int k = find_segment(seg_start, nseg, c0);
    for (; k < nseg && seg_start[k] < c1; ++k) {
        long long a = seg_start[k] > c0 ? seg_start[k] : c0;
        long long b = seg_end[k] < c1 ? seg_end[k] : c1;
        float scale = lr * ratio[k];
        for (long long i = a + threadIdx.x; i < b; i += 256) {
            float wi = w[i];
            float gi = (GRAD_BF16 ? bf2f(((const bf16_t*)gv)[i]) : ((const float*)gv)[i]) + wd * wi;
            float vi = mu * v[i] + scale * gi;
            wi -= vi;
            v[i] = vi; w[i] = wi;
            if (HAS_LOWP) lowp[i] = f2bf(wi);
        }
    }
}

// trust ratio per segment from the squared norms: tc*|w| / (|g| + wd*|w| + eps), 1 where either norm is 0
// (in-tree so that a LARS step contains no framework operator and can be replayed from the launch tape)
__global__ void __launch_bounds__(256) lars_trust_ratio_kernel(const float* __restrict__ sq /*[2][nseg]*/, int nseg, float tc,
                                                              float wd, float eps, float* __restrict__ ratio) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nseg) return;
    const float wn = sqrtf(sq[k]), gn = sqrtf(sq[nseg + k]);
    ratio[k] = (wn > 0.f && gn > 0.f) ? tc * wn / (gn + wd * wn + eps) : 1.f;
}

extern "C" int ddl_lars_trust_ratios(const void* sq, int nseg, float tc, float wd, float eps, void* ratio, hipStream_t stream) {
    if (!sq || !ratio || nseg <= 0) return DDL_EINVAL;
    DDL_KLAUNCH(lars_trust_ratio_kernel, dim3((nseg + 255) / 256), dim3(256), 0, stream, (const float*)sq, nseg, tc, wd, eps,
                (float*)ratio);
    return ddl_launch_status();
}

// floats of scratch `ddl_segmented_sqnorms` needs for a run of n elements
extern "C" long long ddl_segmented_sqnorms_scratch(long long n) { return 4 * ((n + LARS_CHUNK - 1) / LARS_CHUNK); }

// out [2][nseg] = sum w^2 | sum g^2 per segment; segments must be sorted and disjoint; `scratch`: fp32 workspace of
// ddl_segmented_sqnorms_scratch(n) floats
extern "C" int ddl_segmented_sqnorms(const void* w, const void* g, int grad_is_bf16, const void* seg_start,
                                     const void* seg_end, int nseg, long long n, void* out, void* scratch, hipStream_t stream) {
    if (!w || !g || !seg_start || !seg_end || !out || !scratch || nseg <= 0 || n <= 0) return DDL_EINVAL;
    dim3 grid((unsigned)((n + LARS_CHUNK - 1) / LARS_CHUNK)), block(256);
    if (grad_is_bf16)
        DDL_KLAUNCH(segmented_sqnorm_kernel<true>, grid, block, 0, stream, (const float*)w, g,
                           (const long long*)seg_start, (const long long*)seg_end, nseg, n, (float*)out, (float*)scratch);
    else
        DDL_KLAUNCH(segmented_sqnorm_kernel<false>, grid, block, 0, stream, (const float*)w, g,
                           (const long long*)seg_start, (const long long*)seg_end, nseg, n, (float*)out, (float*)scratch);
    DDL_KLAUNCH(segmented_sqnorm_join_kernel, dim3((nseg + 255) / 256), dim3(256), 0, stream, (const long long*)seg_start,
                (const long long*)seg_end, nseg, (const float*)scratch, (float*)out);
    return ddl_launch_status();
}

extern "C" int ddl_fused_lars_ex(void* w, const void* g, void* v, void* lowp, const void* ratio, const void* seg_start,
                                 const void* seg_end, int nseg, long long n, int grad_is_bf16, float lr, float mu,
                                 float wd, const void* lr_scale, hipStream_t stream) {
    if (!w || !g || !v || !ratio || !seg_start || !seg_end || nseg <= 0 || n <= 0) return DDL_EINVAL;
    dim3 grid((unsigned)((n + LARS_CHUNK - 1) / LARS_CHUNK)), block(256);
#define LAUNCH(GB, LP) DDL_KLAUNCH((fused_lars_kernel<GB, LP>), grid, block, 0, stream, (float*)w, g, (float*)v, (bf16_t*)lowp, (const float*)ratio, (const long long*)seg_start, (const long long*)seg_end, nseg, n, lr, mu, wd, (const float*)lr_scale)
    if (grad_is_bf16) { if (lowp) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (lowp) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    return ddl_launch_status();
}

extern "C" int ddl_fused_lars(void* w, const void* g, void* v, void* lowp, const void* ratio, const void* seg_start,
                              const void* seg_end, int nseg, long long n, int grad_is_bf16, float lr, float mu,
                              float wd, hipStream_t stream) {
    return ddl_fused_lars_ex(w, g, v, lowp, ratio, seg_start, seg_end, nseg, n, grad_is_bf16, lr, mu, wd, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// K20: NCHW fp32 (or uint8) -> NHWC bf16 with fused (x*scale - mean)/sd normalisation.
// One thread per output pixel; C <= 4.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_norm_kernel(const T* __restrict__ in, bf16_t* __restrict__ out,
                                                               long long npix_total, int hw, int C, float in_scale,
                                                               float m0, float m1, float m2, float m3,
                                                               float r0, float r1, float r2, float r3) {
    const float mean[4] = {m0, m1, m2, m3}, rsd[4] = {r0, r1, r2, r3};
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix_total; p += stride) {
        long long b = p / hw, q = p - b * hw;
        const T* src = in + b * (long long)C * hw + q;
        bf16_t* dst = out + p * C;
        for (int c = 0; c < C; ++c) dst[c] = f2bf(((float)src[(long long)c * hw] * in_scale - mean[c]) * rsd[c]);
    }
}

extern "C" int ddl_nchw_to_nhwc_norm(const void* in, void* out, long long B, int C, int H, int W, int in_kind,
                                     float in_scale, const float* mean, const float* sd, hipStream_t stream) {
    if (!in || !out || B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return DDL_EINVAL;
    float m[4] = {0, 0, 0, 0}, r[4] = {1, 1, 1, 1};
    for (int c = 0; c < C; ++c) { m[c] = mean ? mean[c] : 0.f; r[c] = sd ? 1.f / sd[c] : 1.f; }
    long long npix = B * H * W;
    dim3 grid(ddl_stream_grid(npix, 256)), block(256);
    if (in_kind == 0)
        DDL_KLAUNCH(nchw_to_nhwc_norm_kernel<float>, grid, block, 0, stream, (const float*)in, (bf16_t*)out, npix,
                           H * W, C, in_scale, m[0], m[1], m[2], m[3], r[0], r[1], r[2], r[3]);
    else if (in_kind == 1)
        DDL_KLAUNCH(nchw_to_nhwc_norm_kernel<unsigned char>, grid, block, 0, stream, (const unsigned char*)in,
                           (bf16_t*)out, npix, H * W, C, in_scale, m[0], m[1], m[2], m[3], r[0], r[1], r[2], r[3]);
    else
        return DDL_EINVAL;
    return ddl_launch_status();
}

// Stream helpers for the Python side (created through the same HIP runtime instance the kernels
// are launched with; torch.cuda.ExternalStream wraps the handle).
extern "C" int ddl_stream_priority_range(int* least, int* greatest) {
    return hipDeviceGetStreamPriorityRange(least, greatest) == hipSuccess ? DDL_OK : DDL_ELAUNCH;
}

// Event helpers for the two-stream backward (ops/hip_ops.py): one ctypes call orders `waiter`
// after everything queued on `producer` so far.  The event may be reused: a wait refers to the
// record that preceded it, later records do not disturb it.
extern "C" int ddl_event_create(void** out) {
    hipEvent_t e = nullptr;
    // DDL_EVENT_FLAGS (profiles/r3_notes.md section 29): 0 = plain events, 1 = device-scope release when the event is recorded
    // (16.20 -> 16.18 ms), 2 = no system-scope fences (16.20 -> 16.13 ms, bit-identical over 6 800 single-process steps and
    // in the two-rank xGMI rig).  These events only order the two streams of ONE device; what leaves the device passes an
    // event of the engine (csrc/engine, plain flags) first.  Default: 2 in a one-process job, 0 when WORLD_SIZE > 1 -
    // nothing between real devices could be tested.
    static const int mode = getenv("DDL_EVENT_FLAGS") ? atoi(getenv("DDL_EVENT_FLAGS"))
                          : (getenv("WORLD_SIZE") && atoi(getenv("WORLD_SIZE")) > 1 ? 0 : 2);
    unsigned flags = hipEventDisableTiming;
    if (mode == 1) flags |= hipEventReleaseToDevice;
    if (mode == 2) flags |= hipEventDisableSystemFence;
    if (!out || hipEventCreateWithFlags(&e, flags) != hipSuccess) return DDL_ELAUNCH;
    *out = (void*)e;
    return DDL_OK;
}

extern "C" int ddl_stream_wait_stream(void* waiter, void* producer, void* event) {
    if (!event) return DDL_EINVAL;
    if (hipEventRecord((hipEvent_t)event, (hipStream_t)producer) != hipSuccess) return DDL_ELAUNCH;
    if (hipStreamWaitEvent((hipStream_t)waiter, (hipEvent_t)event, 0) != hipSuccess) return DDL_ELAUNCH;
    if (ddl::tape::recording())
        ddl::tape::push([=]() {
            (void)hipEventRecord((hipEvent_t)event, (hipStream_t)producer);
            (void)hipStreamWaitEvent((hipStream_t)waiter, (hipEvent_t)event, 0);
        });
    return DDL_OK;
}

extern "C" int ddl_stream_create(int priority, void** out) {
    hipStream_t s = nullptr;
    if (!out || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) return DDL_ELAUNCH;
    *out = (void*)s;
    return DDL_OK;
}

