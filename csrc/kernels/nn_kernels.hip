// Memory-bound network kernels on NHWC bf16 activations:
//   K6  bn_stats_partial / bn_fwd_finalize   per-channel batch statistics (fp32, Chan-combined)
//   K7  bn_apply_fwd                         z = relu(y*scale + shift [+ residual])
//   K8  bn_bwd_reduce / _finalize / _apply   dgamma, dbeta, dy (+ masked grad for the shortcut)
//   K10 maxpool3x3s2 fwd/bwd (argmax byte kept), K11 global average pool fwd/bwd
//   K13 softmax cross-entropy fwd+bwd in one pass, bias add, K3 helper: weight flip-transpose
// All: wave64, 256-thread blocks unless noted, 16-byte (8 x bf16) lanes, fp32 math.
#include "common.h"
#include <cstdlib>

namespace {

// Streaming loads / stores with the non-temporal hint.  In isolation (benchmarks/native/
// stream_bench) the hint gains 10-15 % GB/s on tensors larger than the 256 MB infinity cache and
// loses on smaller ones; inside the training step, where ~30 GB stream through per iteration, it
// wins for every size (bench.py: 21.61 ms never, 21.42 ms >= 256 MB, 21.18 >= 128 MB, 21.01 always),
// so the threshold defaults to 0.
template <bool NT> __device__ __forceinline__ bf16x8 ld8(const bf16_t* p) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(p);
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT> __device__ __forceinline__ void st8(bf16_t* p, bf16x8 v) {
    bf16x8* q = reinterpret_cast<bf16x8*>(p);
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
}
static long long nt_threshold_bytes() {
    static const long long v = getenv("DDL_NT_BYTES") ? atoll(getenv("DDL_NT_BYTES")) : 0LL;  // tuning knob
    return v;
}

// ---------------------------------------------------------------------------------------------
// K6a: standalone statistics partials [T][2][C] over tiles of `rows_per_tile` rows (used when the
// producer is not our conv kernel).  256 threads = (C/8 chunk columns) x row lanes.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_stats_partial_kernel(const bf16_t* __restrict__ y, float* __restrict__ partial,
                                                              long long M, int C, int rows_per_tile) {
    __shared__ float red[2][256 * 8 / 1];  // [2][rowlane*C8*8] folded below (max 2048 floats each)
    const int cchunks = C >> 3, rlanes = 256 / cchunks;
    const int cc = threadIdx.x % cchunks, rl = threadIdx.x / cchunks;
    const long long r0 = (long long)blockIdx.x * rows_per_tile;
    long long r1 = r0 + rows_per_tile;
    if (r1 > M) r1 = M;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    for (long long r = r0 + rl; r < r1; r += rlanes) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(y + r * C + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { float f = bf2f(v[e]); s1[e] += f; s2[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = s1[e]; red[1][threadIdx.x * 8 + e] = s2[e]; }
    __syncthreads();
    // thread t < 2*C sums the row lanes of channel (t % C)
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        int which = t / C, c = t - which * C;
        float acc = 0.f;
        for (int l = 0; l < rlanes; ++l) acc += red[which][(l * cchunks + (c >> 3)) * 8 + (c & 7)];
        partial[((long long)blockIdx.x * 2 + which) * C + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// K6b: combine the per-tile partials [T][2][C] (sum, sum of squares of `rows_per_tile` rows each)
// into mean / invstd.  Two small launches so that the 3-13 MB partial table of the 56x56 layers
// is read by many CUs: stage 1 folds T/P tiles per block in fp64 (no cancellation across tiles),
// stage 2 (one thread per channel) folds the P slices, updates the running statistics (unbiased
// variance, like torch) and emits the fused scale/shift of K7.
// ---------------------------------------------------------------------------------------------
// Fold tiles t0+ty, t0+ty+16, ... < t1 of channel c: eight tiles (16 loads) in flight per round.
// These kernels sit on dependent chains (conv -> fold -> finalize -> apply) and their run time
// is load latency, which under a busy weight-gradient stream is several microseconds a round.
template <typename Acc>
__device__ __forceinline__ void bn_fold_lane16(const float* __restrict__ partial, int t0, int t1, int C, int c, int ty,
                                               Acc& s1, Acc& s2) {
    int t = t0 + ty;
    for (; t + 112 < t1; t += 128) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = partial[((long long)(t + 16 * j) * 2) * C + c];
            b[j] = partial[((long long)(t + 16 * j) * 2 + 1) * C + c];
        }
        s1 += (((Acc)a[0] + (Acc)a[1]) + ((Acc)a[2] + (Acc)a[3])) + (((Acc)a[4] + (Acc)a[5]) + ((Acc)a[6] + (Acc)a[7]));
        s2 += (((Acc)b[0] + (Acc)b[1]) + ((Acc)b[2] + (Acc)b[3])) + (((Acc)b[4] + (Acc)b[5]) + ((Acc)b[6] + (Acc)b[7]));
    }
    for (; t + 16 < t1; t += 32) {
        float a0 = partial[((long long)t * 2) * C + c], b0 = partial[((long long)t * 2 + 1) * C + c];
        float a1 = partial[((long long)(t + 16) * 2) * C + c], b1 = partial[((long long)(t + 16) * 2 + 1) * C + c];
        s1 += (Acc)a0 + (Acc)a1;
        s2 += (Acc)b0 + (Acc)b1;
    }
    if (t < t1) {
        s1 += (Acc)partial[((long long)t * 2) * C + c];
        s2 += (Acc)partial[((long long)t * 2 + 1) * C + c];
    }
}

// grid (C/16, P), 256 threads = 16 channels x 16 tile lanes
template <typename OutT>
__global__ void __launch_bounds__(256) bn_stats_fold_kernel(const float* __restrict__ partial, int T, int C, int P,
                                                           OutT* __restrict__ folded /*[P][2][C]*/) {
    __shared__ double r1[16][16], r2[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    const int per = (T + P - 1) / P;
    const int t0 = blockIdx.y * per;
    int t1 = t0 + per;
    if (t1 > T) t1 = T;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) bn_fold_lane16<double>(partial, t0, t1, C, c, ty, s1, s2);
    r1[ty][tx] = s1; r2[ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int l = 1; l < 16; ++l) { s1 += r1[l][tx]; s2 += r2[l][tx]; }
        folded[((long long)blockIdx.y * 2) * C + c] = (OutT)s1;
        folded[((long long)blockIdx.y * 2 + 1) * C + c] = (OutT)s2;
    }
}

__device__ __forceinline__ void bn_fwd_finish(int c, double s1, double s2, long long M, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float* __restrict__ rmean,
                                              float* __restrict__ rvar, float momentum, float eps,
                                              float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                              float* __restrict__ scale, float* __restrict__ shift) {
    const double n = (double)M;
    const double mean = s1 / n;
    double m2 = s2 - s1 * mean;
    if (m2 < 0.0) m2 = 0.0;
    const float var = (float)(m2 / n);
    const float inv = rsqrtf(var + eps);
    mean_out[c] = (float)mean; invstd_out[c] = inv;
    const float g = gamma[c] * inv;
    scale[c] = g; shift[c] = beta[c] - (float)mean * g;
    if (rmean) {
        const float unbiased = M > 1 ? (float)(m2 / (n - 1.0)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}

__global__ void __launch_bounds__(256) bn_fwd_finalize_kernel(const double* __restrict__ folded, int P, int C, long long M,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ rmean, float* __restrict__ rvar,
                                                             float momentum, float eps, float* __restrict__ mean_out,
                                                             float* __restrict__ invstd_out, float* __restrict__ scale,
                                                             float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0, u1 = 0.0, u2 = 0.0, w1 = 0.0, w2 = 0.0;
    int p = 0;
    for (; p + 3 < P; p += 4) {  // independent chains: the loads of one round overlap
        s1 += folded[((long long)p * 2) * C + c];       s2 += folded[((long long)p * 2 + 1) * C + c];
        t1 += folded[((long long)p * 2 + 2) * C + c];   t2 += folded[((long long)p * 2 + 3) * C + c];
        u1 += folded[((long long)p * 2 + 4) * C + c];   u2 += folded[((long long)p * 2 + 5) * C + c];
        w1 += folded[((long long)p * 2 + 6) * C + c];   w2 += folded[((long long)p * 2 + 7) * C + c];
    }
    for (; p < P; ++p) {
        s1 += folded[((long long)p * 2) * C + c];
        s2 += folded[((long long)p * 2 + 1) * C + c];
    }
    s1 = (s1 + t1) + (u1 + w1);
    s2 = (s2 + t2) + (u2 + w2);
    bn_fwd_finish(c, s1, s2, M, gamma, beta, rmean, rvar, momentum, eps, mean_out, invstd_out, scale, shift);
}

// Short partial tables (the 14x14 and 7x7 layers: <= 512 tiles) in ONE launch: 64 channels x 16
// tile lanes per block, fp64 accumulation, LDS fold, then the same finish as above.
__global__ void __launch_bounds__(1024) bn_fwd_finalize_direct_kernel(const float* __restrict__ partial, int T, int C, long long M,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float* __restrict__ rmean, float* __restrict__ rvar,
                                                                     float momentum, float eps, float* __restrict__ mean_out,
                                                                     float* __restrict__ invstd_out, float* __restrict__ scale,
                                                                     float* __restrict__ shift) {
    __shared__ double r1[16][64], r2[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int t = ty;
        for (; t + 48 < T; t += 64) {
            float a0 = partial[((long long)t * 2) * C + c], b0 = partial[((long long)t * 2 + 1) * C + c];
            float a1 = partial[((long long)(t + 16) * 2) * C + c], b1 = partial[((long long)(t + 16) * 2 + 1) * C + c];
            float a2 = partial[((long long)(t + 32) * 2) * C + c], b2 = partial[((long long)(t + 32) * 2 + 1) * C + c];
            float a3 = partial[((long long)(t + 48) * 2) * C + c], b3 = partial[((long long)(t + 48) * 2 + 1) * C + c];
            s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; t < T; t += 16) {
            s1 += (double)partial[((long long)t * 2) * C + c];
            s2 += (double)partial[((long long)t * 2 + 1) * C + c];
        }
    }
    r1[ty][tx] = s1; r2[ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int l = 1; l < 16; ++l) { s1 += r1[l][tx]; s2 += r2[l][tx]; }
        bn_fwd_finish(c, s1, s2, M, gamma, beta, rmean, rvar, momentum, eps, mean_out, invstd_out, scale, shift);
    }
}

// Long partial tables in ONE launch: grid (C/16, P) folds like bn_stats_fold_kernel, then the block
// that finishes LAST for its 16 channels (a ticket per channel group; release by every block, acquire
// by the last) folds the P slices in a fixed order and runs the finish - the separate finalize
// launch and its kernel boundary (25 + 25 per ResNet-50 step, all on the dependent chain conv ->
// statistics -> apply) are gone.  MEASURED SLOWER (19.05-19.11 ms against 18.79 ms per ResNet-50 step,
// profiles/r2_notes.md: the agent-scope release of every block writes its XCD's L2 back while the
// weight-gradient stream keeps it dirty), so it is opt-in: DDL_BN_ONE_LAUNCH=1.  The tickets live in a process-lifetime buffer and are reset by
// the block that consumed them; launches that share it must be stream-ordered (they are: main stream).
// MODE 0: forward (mean / invstd / scale / shift, running statistics); MODE 1: backward
// (dgamma / dbeta, coefficients of dy = ca*g + cb*y + cc).
struct BnFinishArgs {
    long long M;
    const float* gamma; const float* beta;
    float* rmean; float* rvar; float momentum, eps;
    float* mean_out; float* invstd_out; float* scale; float* shift;      // forward outputs
    const float* mean; const float* invstd; float* dgamma; float* dbeta; float* coef; int accumulate;  // backward
};

template <typename Acc, int MODE>
__global__ void __launch_bounds__(256) bn_fold_finish_kernel(const float* __restrict__ partial, int T, int C, int P,
                                                            Acc* __restrict__ folded /*[P][2][C]*/, int* __restrict__ tickets,
                                                            const BnFinishArgs a) {
    __shared__ double r1[16][16], r2[16][16];
    __shared__ int is_last;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    const int per = (T + P - 1) / P;
    const int t0 = blockIdx.y * per;
    int t1 = t0 + per;
    if (t1 > T) t1 = T;
    Acc s1 = 0, s2 = 0;
    if (c < C) bn_fold_lane16<Acc>(partial, t0, t1, C, c, ty, s1, s2);
    r1[ty][tx] = (double)s1; r2[ty][tx] = (double)s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        double d1 = r1[0][tx], d2 = r2[0][tx];
        for (int l = 1; l < 16; ++l) { d1 += r1[l][tx]; d2 += r2[l][tx]; }
        folded[((long long)blockIdx.y * 2) * C + c] = (Acc)d1;
        folded[((long long)blockIdx.y * 2 + 1) * C + c] = (Acc)d2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == P - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) tickets[blockIdx.x] = 0;     // ready for the next launch (stream-ordered)
    double u1 = 0.0, u2 = 0.0;
    if (c < C) {
        for (int p = ty; p < P; p += 16) {
            u1 += (double)folded[((long long)p * 2) * C + c];
            u2 += (double)folded[((long long)p * 2 + 1) * C + c];
        }
    }
    r1[ty][tx] = u1; r2[ty][tx] = u2;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    for (int l = 1; l < 16; ++l) { u1 += r1[l][tx]; u2 += r2[l][tx]; }
    if (MODE == 0) {
        bn_fwd_finish(c, u1, u2, a.M, a.gamma, a.beta, a.rmean, a.rvar, a.momentum, a.eps, a.mean_out, a.invstd_out, a.scale, a.shift);
    } else {
        const float f1 = (float)u1, f2 = (float)u2;
        const float mu = a.mean[c], inv = a.invstd[c];
        const float dg = inv * (f2 - mu * f1);
        if (a.accumulate) { a.dgamma[c] += dg; a.dbeta[c] += f1; } else { a.dgamma[c] = dg; a.dbeta[c] = f1; }
        const float invM = 1.f / (float)a.M;
        const float ca = a.gamma[c] * inv;
        const float cb = -ca * inv * dg * invM;
        a.coef[c] = ca; a.coef[C + c] = cb; a.coef[2 * C + c] = -ca * f1 * invM - cb * mu;
    }
}

static int* bn_tickets() {
    // per device: 2 x 256 tickets (forward half, backward half), zeroed once, self-resetting
    static int* bufs[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!bufs[dev]) {
        int* b = nullptr;
        if (hipMalloc(&b, 512 * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(b, 0, 512 * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
        bufs[dev] = b;
    }
    return bufs[dev];
}

// eval mode: scale/shift straight from the running statistics
__global__ void __launch_bounds__(256) bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                            float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        float g = gamma[c] * rsqrtf(rvar[c] + eps);
        scale[c] = g; shift[c] = beta[c] - rmean[c] * g;
    }
}

// ---------------------------------------------------------------------------------------------
// K7: z = act(y*scale[c] + shift[c] [+ residual])
// ---------------------------------------------------------------------------------------------
// RES: 0 none, 1 residual tensor added as stored, 2 residual = rscale[c]*res + rshift[c] (the raw
// output of the projection-shortcut conv with ITS BatchNorm applied on the fly: the shortcut's
// normalised output is never written to or re-read from memory)
template <bool RELU, int RES, bool NT>
__global__ void __launch_bounds__(256) bn_apply_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                          const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                          bf16_t* __restrict__ z, unsigned char* __restrict__ bits,
                                                          long long nchunks, int cchunks) {
    long long stride = (long long)gridDim.x * blockDim.x;
    // RES == 2 reads four coefficient vectors per chunk: 128 bytes through the vector memory path for
    // 32 bytes of data (3.5-5.2 TB/s against 6.3 for RES == 1, profiles/r2_pmc_training_step.md).  The
    // block stages them in LDS once (dynamic LDS: 4 * C floats) and every thread reads its slices there.
    extern __shared__ __attribute__((aligned(16))) float co_lds[];
    if (RES == 2) {
        const int C = cchunks * 8;
        for (int t = threadIdx.x * 4; t < C; t += blockDim.x * 4) {
            *reinterpret_cast<f32x4*>(co_lds + t) = *reinterpret_cast<const f32x4*>(scale + t);
            *reinterpret_cast<f32x4*>(co_lds + C + t) = *reinterpret_cast<const f32x4*>(shift + t);
            *reinterpret_cast<f32x4*>(co_lds + 2 * C + t) = *reinterpret_cast<const f32x4*>(rscale + t);
            *reinterpret_cast<f32x4*>(co_lds + 3 * C + t) = *reinterpret_cast<const f32x4*>(rshift + t);
        }
        __syncthreads();
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        int c = (int)(i % cchunks) * 8;
        bf16x8 v = ld8<NT>(y + (i << 3));
        float sc[8], sh[8], rsc[8], rsh[8];
        bf16x8 r;
        if (RES) r = ld8<NT>(res + (i << 3));
        if (RES == 2) {
            const int C = cchunks * 8;
            load8f(co_lds + c, sc); load8f(co_lds + C + c, sh);
            load8f(co_lds + 2 * C + c, rsc); load8f(co_lds + 3 * C + c, rsh);
        } else {
            load8f(scale + c, sc); load8f(shift + c, sh);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = bn_affine(bf2f(v[e]), sc[e], sh[e]);
            if (RES == 1) f += bf2f(r[e]);
            if (RES == 2) f += bn_affine(bf2f(r[e]), rsc[e], rsh[e]);
            if (RELU) f = fmaxf(f, 0.f);
            o[e] = f2bf(f);
        }
        st8<NT>(z + (i << 3), o);
        if (bits) {  // ReLU mask of this chunk, one bit per element, for the backward epilogues
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
            bits[i] = (unsigned char)m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K8a: per-block partial sums  s1[c] = sum g,  s2[c] = sum g*y,  g = dz * (z > 0) (RELU) or dz.
// grid = G blocks over contiguous row ranges; 256 threads = C/8 chunk columns x row lanes.
// ---------------------------------------------------------------------------------------------
// MASK: 0 = no ReLU, 1 = mask from the stored activation z, 2 = mask recomputed from y with the
// forward scale/shift (non-residual BN+ReLU: saves the read of z in both backward passes).
template <int MASK>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y,
                                                           const bf16_t* __restrict__ z, const float* __restrict__ fcoef,
                                                           float* __restrict__ partial,
                                                           long long M, int C, long long rows_per_block) {
    __shared__ float red[2][2048];
    const int cchunks = C >> 3, rlanes = 256 / cchunks;
    const int cc = threadIdx.x % cchunks, rl = threadIdx.x / cchunks;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s1[8], s2[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (MASK == 2) { load8f(fcoef + cc * 8, sc); load8f(fcoef + C + cc * 8, sh); }
    for (long long r = r0 + rl; r < r1; r += rlanes) {
        long long o = r * C + cc * 8;
        bf16x8 d = *reinterpret_cast<const bf16x8*>(dz + o);
        bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + o);
        bf16x8 zv;
        if (MASK == 1) zv = *reinterpret_cast<const bf16x8*>(z + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = bf2f(d[e]), yf = bf2f(yv[e]);
            if (MASK == 1) g = bf2f(zv[e]) > 0.f ? g : 0.f;
            if (MASK == 2) g = bn_affine(yf, sc[e], sh[e]) > 0.f ? g : 0.f;
            s1[e] += g; s2[e] += g * yf;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = s1[e]; red[1][threadIdx.x * 8 + e] = s2[e]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        int which = t / C, c = t - which * C;
        float acc = 0.f;
        for (int l = 0; l < rlanes; ++l) acc += red[which][(l * cchunks + (c >> 3)) * 8 + (c & 7)];
        partial[((long long)blockIdx.x * 2 + which) * C + c] = acc;
    }
}

// K8b: dgamma = invstd*(s2 - mean*s1), dbeta = s1, and the per-channel coefficients of
// dy = ca*g + cb*y + cc  (ca = gamma*invstd, cb = -ca*invstd*dgamma/M, cc = -ca*s1/M - cb*mean)
// 16 channels x 16 tile lanes per block of 256 threads (2 KB of LDS): the kernel sits on the
// dependent chain dgrad -> finalize -> apply while the weight-gradient stream keeps the CUs full,
// and a 1024-thread block with 8 KB of LDS waited 10-60 us for a CU that could hold it
// (kernel trace, profiles/r1_notes.md); small blocks slip into the holes.
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, int G, int C, long long M,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef /*[3][C]*/, int accumulate) {
    __shared__ float a1[16][16], a2[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) bn_fold_lane16<float>(partial, 0, G, C, c, ty, s1, s2);
    a1[ty][tx] = s1; a2[ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int l = 1; l < 16; ++l) { s1 += a1[l][tx]; s2 += a2[l][tx]; }
        float mu = mean[c], inv = invstd[c];
        float dg = inv * (s2 - mu * s1);
        if (accumulate) { dgamma[c] += dg; dbeta[c] += s1; } else { dgamma[c] = dg; dbeta[c] = s1; }
        float invM = 1.f / (float)M;
        float ca = gamma[c] * inv;
        float cb = -ca * inv * dg * invM;
        coef[c] = ca; coef[C + c] = cb; coef[2 * C + c] = -ca * s1 * invM - cb * mu;
    }
}

// K8c: dy = ca*g + cb*y + cc ; optionally also store g (gradient of the residual branch)
template <int MASK, bool WRITE_G, bool NT>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ z, const float* __restrict__ fcoef,
                                                          const float* __restrict__ coef,
                                                          bf16_t* __restrict__ dy, bf16_t* __restrict__ gout,
                                                          long long nchunks, int cchunks) {
    const int C = cchunks * 8;
    long long stride = (long long)gridDim.x * blockDim.x;
    // coefficients through LDS (3 * C floats, 5 * C with the recomputed mask): see bn_apply_fwd_kernel
    extern __shared__ __attribute__((aligned(16))) float co_lds[];
    for (int t = threadIdx.x * 4; t < 3 * C; t += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(co_lds + t) = *reinterpret_cast<const f32x4*>(coef + t);
    if (MASK == 2) {
        for (int t = threadIdx.x * 4; t < 2 * C; t += blockDim.x * 4)
            *reinterpret_cast<f32x4*>(co_lds + 3 * C + t) = *reinterpret_cast<const f32x4*>(fcoef + t);
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        int c = (int)(i % cchunks) * 8;
        bf16x8 d = ld8<NT>(dz + (i << 3));
        bf16x8 yv = ld8<NT>(y + (i << 3));
        bf16x8 zv;
        if (MASK == 1) zv = ld8<NT>(z + (i << 3));
        float ca[8], cb[8], c0[8], sc[8], sh[8];
        load8f(co_lds + c, ca); load8f(co_lds + C + c, cb); load8f(co_lds + 2 * C + c, c0);
        if (MASK == 2) { load8f(co_lds + 3 * C + c, sc); load8f(co_lds + 4 * C + c, sh); }
        bf16x8 o, go;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = bf2f(d[e]), yf = bf2f(yv[e]);
            if (MASK == 1) g = bf2f(zv[e]) > 0.f ? g : 0.f;
            if (MASK == 2) g = bn_affine(yf, sc[e], sh[e]) > 0.f ? g : 0.f;
            o[e] = f2bf(bn_bwd_affine(g, yf, ca[e], cb[e], c0[e]));
            if (WRITE_G) go[e] = f2bf(g);
        }
        st8<NT>(dy + (i << 3), o);
        if (WRITE_G) st8<NT>(gout + (i << 3), go);
    }
}

// ---------------------------------------------------------------------------------------------
// K10: 3x3 / stride 2 / pad 1 max pooling, NHWC.  Forward keeps the winning tap (0..8, first
// maximum in scan order like torch) as one byte per element; backward gathers.
// ---------------------------------------------------------------------------------------------
// BN = true: the input is a pre-BatchNorm tensor and z = relu(x*scale + shift) is applied on the fly
// (stem fusion: the 112x112 activation is never written).  All nine taps are loaded with clamped
// addresses before the first use (a branch per tap made hipcc wait for every load in turn).
template <bool BN>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, bf16_t* __restrict__ out,
                                                         unsigned char* __restrict__ idx, int B, int H, int W, int C,
                                                         int Ho, int Wo) {
    const int cchunks = C >> 3;
    long long total = (long long)B * Ho * Wo * cchunks;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned iu = (unsigned)i;              // total < 2^32 (checked by the launcher)
        int cc = (int)(iu % (unsigned)cchunks);
        unsigned pix = iu / (unsigned)cchunks;
        int wo = (int)(pix % (unsigned)Wo);
        unsigned t = pix / (unsigned)Wo;
        int ho = (int)(t % (unsigned)Ho), b = (int)(t / (unsigned)Ho);
        // Taps outside the image are read from the clamped position, which is another tap of the
        // same window: the duplicate never changes the maximum, and it carries the window index of
        // the pixel it duplicates (kk), so the recorded argmax is a real tap.  Scan order keeps the
        // first maximum among the valid taps, exactly as with explicit validity flags - which cost
        // a mask AND/OR per tap and element and spilled scalar registers (64 v_readlane + 63
        // v_writelane in the disassembly; the kernel is bound by vector-instruction issue).
        bf16x8 v[9];
        unsigned kk[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
                int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
                kk[r * 3 + s] = (unsigned)((hc - (ho * 2 - 1)) * 3 + (wc - (wo * 2 - 1)));
                v[r * 3 + s] = ld8<true>(x + (((long long)b * H + hc) * W + wc) * C + cc * 8);
            }
        bf16x8 o;
        u32x2 packed = {0u, 0u};
        if (BN) {
            // One integer maximum per tap and element finds value AND argmax: the key is the
            // bf16-rounded activation in the high half (what the unfused path stored and compared)
            // over 15 - tap in the low bits.  As SIGNED integers positive floats order like floats and
            // every negative one sorts below them; among equal values the earlier tap wins (window
            // indices grow in scan order).  A window without a positive tap yields some negative
            // winner: its output is relu(.) = 0 either way and its gradient is masked in backward,
            // so which tap is recorded there does not matter.  NaN keys are the largest: they propagate.
            float sc[8], sh[8];
            load8f(scale + cc * 8, sc); load8f(shift + cc * 8, sh);
            int key[8];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const unsigned low = 15u - kk[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x2 pr;
                    pr[0] = f2bf(bn_affine(bf2f(v[k][2 * q]), sc[2 * q], sh[2 * q]));
                    pr[1] = f2bf(bn_affine(bf2f(v[k][2 * q + 1]), sc[2 * q + 1], sh[2 * q + 1]));
                    const unsigned u = __builtin_bit_cast(unsigned, pr);
                    const int k0 = (int)((u << 16) | low), k1 = (int)((u & 0xffff0000u) | low);
                    key[2 * q] = k == 0 ? k0 : (key[2 * q] > k0 ? key[2 * q] : k0);
                    key[2 * q + 1] = k == 0 ? k1 : (key[2 * q + 1] > k1 ? key[2 * q + 1] : k1);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float win = __builtin_bit_cast(float, (unsigned)key[e] & 0xffff0000u);
                o[e] = f2bf(fmaxf(win, 0.f));   // (fmaxf drops a NaN operand: keep it)
                if (win != win) o[e] = f2bf(win);
                packed[e >> 2] |= (15u - ((unsigned)key[e] & 15u)) << (8 * (e & 3));
            }
        } else {
            float best[8];
            unsigned bi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = 0.f; bi[e] = kk[0]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(v[k][e]);
                    if (k == 0) { best[e] = f; continue; }   // the first valid pixel (or its duplicate) starts the scan
                    const float prev = best[e];
                    best[e] = fmaxf(prev, f);
                    bi[e] = f > prev ? kk[k] : bi[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = f2bf(best[e]);
                packed[e >> 2] |= bi[e] << (8 * (e & 3));
            }
        }
        *reinterpret_cast<bf16x8*>(out + (i << 3)) = o;
        if (idx) *reinterpret_cast<u32x2*>(idx + (i << 3)) = packed;
    }
}

// Gradient of the 3x3/2 max pool at input pixel (hi, wi), 8 channels: the up to four windows that
// contain the pixel are read with clamped addresses (all loads first), a window contributes where
// its stored argmax is this pixel's tap.  Rounded to bf16 like the tensor the unfused path wrote.
__device__ __forceinline__ void pool_grad8(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ idx, int b,
                                           int hi, int wi, int cc, int cchunks, int Ho, int Wo, float (&g)[8]) {
    // Pixel (hi, wi) lies in window row h0 = hi >> 1 at tap row (hi & 1) + 1 and, when hi is odd, also
    // in window row h0 + 1 at tap row 0 (if that row exists); the same for columns.  Window (h0, w0)
    // always exists.  A window that does not count gets the tap code 0xff, which no argmax byte holds,
    // and is read from the address of window (h0, w0) (all loads before the first use, no branches).
    const int h0 = hi >> 1, w0 = wi >> 1;
    const bool hv = (hi & 1) && (h0 + 1 < Ho), wv = (wi & 1) && (w0 + 1 < Wo);
    const unsigned r0 = (unsigned)(hi & 1) + 1u, c0 = (unsigned)(wi & 1) + 1u;
    const long long base = ((long long)((unsigned)(b * Ho + h0) * (unsigned)Wo + (unsigned)w0) * cchunks + cc) << 3;
    const int dw = wv ? cchunks * 8 : 0, dh = hv ? Wo * cchunks * 8 : 0;
    unsigned tap[4];
    tap[0] = r0 * 3u + c0;
    tap[1] = wv ? r0 * 3u : 0xffu;
    tap[2] = hv ? c0 : 0xffu;
    tap[3] = (hv && wv) ? 0u : 0xffu;
    bf16x8 d[4];
    u32x2 pk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long o = base + ((k & 1) ? dw : 0) + ((k >> 1) ? dh : 0);
        pk[k] = *reinterpret_cast<const u32x2*>(idx + o);
        d[k] = *reinterpret_cast<const bf16x8*>(dpool + o);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned w = (pk[k][e >> 2] >> (8 * (e & 3))) & 0xffu;
            g[e] += (w == tap[k]) ? bf2f(d[k][e]) : 0.f;
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = bf2f(f2bf(g[e]));
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                         bf16_t* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    const int cchunks = C >> 3;
    long long total = (long long)B * H * W * cchunks;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned iu = (unsigned)i;
        int cc = (int)(iu % (unsigned)cchunks);
        unsigned pix = iu / (unsigned)cchunks;
        int wi = (int)(pix % (unsigned)W);
        unsigned t = pix / (unsigned)W;
        int hi = (int)(t % (unsigned)H), b = (int)(t / (unsigned)H);
        float g[8];
        pool_grad8(dy, idx, b, hi, wi, cc, cchunks, Ho, Wo, g);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e]);
        st8<true>(dx + (i << 3), o);
    }
}

// Stem backward, fused: g = relu'(bn(y)) * maxpool_backward(dpool) is recomputed from the pooled
// gradient and the argmax bytes instead of being read from a 112x112 tensor.
//   pass 1 (APPLY = false): per-block partial sums  s1 = sum g, s2 = sum g*y   -> [G][2][C]
//   pass 2 (APPLY = true):  dy = ca*g + cb*y + c0
template <bool APPLY>
__global__ void __launch_bounds__(256) stem_pool_bn_bwd_kernel(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                              const bf16_t* __restrict__ y, const float* __restrict__ fcoef,
                                                              const float* __restrict__ coef, float* __restrict__ partial,
                                                              bf16_t* __restrict__ dy, int B, int H, int W, int C, int Ho, int Wo,
                                                              long long patches_per_block) {
    // A thread owns a 2x2 pixel patch (rows 2a, 2a+1, columns 2q, 2q+1) of 8 channels.  The patch
    // touches exactly four pool windows - (a, q) holds all four pixels, (a, q+1) the odd column,
    // (a+1, q) the odd row, (a+1, q+1) the odd/odd pixel - so 4 + 4 window loads serve 4 pixels.
    // (One pixel per thread read four windows each, 16 + 16 loads per patch, most of them clamped
    // duplicates: the kernel sat on the vector memory path at 4.1 TB/s.)
    __shared__ float red[2][2048];
    const int cchunks = C >> 3, planes = 256 / cchunks;
    const int cc = threadIdx.x % cchunks, pl = threadIdx.x / cchunks;
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const long long NP = (long long)B * PH * PW;
    const long long p0 = (long long)blockIdx.x * patches_per_block;
    long long p1 = p0 + patches_per_block;
    if (p1 > NP) p1 = NP;
    float s1[8], s2[8], sc[8], sh[8], ca[8], cb[8], c0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    load8f(fcoef + cc * 8, sc); load8f(fcoef + C + cc * 8, sh);
    if (APPLY) { load8f(coef + cc * 8, ca); load8f(coef + C + cc * 8, cb); load8f(coef + 2 * C + cc * 8, c0); }
    // patch coordinates of the first patch, then stepped by `planes` patches per trip
    int q, a, b;
    {
        const unsigned pu = (unsigned)(p0 + pl);
        q = (int)(pu % (unsigned)PW);
        const unsigned t = pu / (unsigned)PW;
        a = (int)(t % (unsigned)PH); b = (int)(t / (unsigned)PH);
    }
    for (long long p = p0 + pl; p < p1; p += planes) {
        const bool rv = 2 * a + 1 < H, cv = 2 * q + 1 < W;        // the patch's odd row / column are inside the image
        const bool wv = q + 1 < Wo, hv = a + 1 < Ho;               // windows to the right / below exist
        // windows (a, q), (a, q+1), (a+1, q), (a+1, q+1); a window that does not exist is read at (a, q) and never matches
        const long long wbase = ((long long)((unsigned)(b * Ho + a) * (unsigned)Wo + (unsigned)q) * cchunks + cc) << 3;
        const int dwo = wv ? cchunks * 8 : 0, dho = hv ? Wo * cchunks * 8 : 0;
        bf16x8 d[4];
        u32x2 pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long o = wbase + ((k & 1) ? dwo : 0) + ((k >> 1) ? dho : 0);
            pk[k] = *reinterpret_cast<const u32x2*>(idx + o);
            d[k] = *reinterpret_cast<const bf16x8*>(dpool + o);
        }
        // the four pixels (missing ones are read at the patch's first pixel and dropped)
        const long long ybase = (((long long)(b * H + 2 * a) * W + 2 * q) * C) + cc * 8;
        const long long ydw = cv ? C : 0, ydh = rv ? (long long)W * C : 0;
        bf16x8 yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = ld8<true>(y + ybase + ((j & 1) ? ydw : 0) + ((j >> 1) ? ydh : 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dh = j >> 1, dwp = j & 1;
            const bool live = (!dh | rv) & (!dwp | cv);            // uniform per thread and pixel
            // tap code of this pixel in each window that holds it (0xff: the window does not hold it / does not exist)
            unsigned tap[4];
            tap[0] = (unsigned)((dh + 1) * 3 + (dwp + 1));
            tap[1] = (dwp && wv) ? (unsigned)((dh + 1) * 3) : 0xffu;
            tap[2] = (dh && hv) ? (unsigned)(dwp + 1) : 0xffu;
            tap[3] = (dh && dwp && hv && wv) ? 0u : 0xffu;
            float g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsigned w = (pk[k][e >> 2] >> (8 * (e & 3))) & 0xffu;
                    g[e] += (w == tap[k]) ? bf2f(d[k][e]) : 0.f;
                }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gr = bf2f(f2bf(g[e]));   // rounded like the tensor the unfused path wrote
                const float yf = bf2f(yv[j][e]);
                const float ge = (live && bn_affine(yf, sc[e], sh[e]) > 0.f) ? gr : 0.f;
                if (APPLY) o[e] = f2bf(ca[e] * ge + cb[e] * yf + c0[e]);
                else { s1[e] += ge; s2[e] += ge * yf; }
            }
            if (APPLY && live) st8<true>(dy + ybase + (dwp ? ydw : 0) + (dh ? ydh : 0), o);
        }
        q += planes;
        while (q >= PW) { q -= PW; if (++a == PH) { a = 0; ++b; } }
    }
    if (!APPLY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = s1[e]; red[1][threadIdx.x * 8 + e] = s2[e]; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * C; t += 256) {
            int which = t / C, c = t - which * C;
            float acc = 0.f;
            for (int l = 0; l < planes; ++l) acc += red[which][(l * cchunks + (c >> 3)) * 8 + (c & 7)];
            partial[((long long)blockIdx.x * 2 + which) * C + c] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K11: global average pool [B,HW,C] -> [B,C] and its gradient
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int HW, int C) {
    const int cchunks = C >> 3;
    long long total = (long long)B * cchunks;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int cc = (int)(i % cchunks);
    long long b = i / cchunks;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const bf16_t* p = x + b * HW * C + cc * 8;
#pragma unroll 7
    for (int k = 0; k < HW; ++k) {  // 7 loads in flight per trip (HW = 49 in ResNet)
        bf16x8 v = ld8<true>(p + (long long)k * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
    }
    float inv = 1.f / (float)HW;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv);
    *reinterpret_cast<bf16x8*>(out + (i << 3)) = o;
}

__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int HW, int C) {
    const int cchunks = C >> 3;
    long long total = (long long)B * HW * cchunks;
    long long stride = (long long)gridDim.x * blockDim.x;
    float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int cc = (int)(i % cchunks);
        long long b = i / ((long long)cchunks * HW);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (b * cchunks + cc) * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(v[e]) * inv);
        *reinterpret_cast<bf16x8*>(dx + (i << 3)) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// K13: softmax cross-entropy, one wave per row.  loss_rows[b] = logsumexp - logit[label];
// dlogits[b][n] = (softmax - onehot) * grad_scale, written with row stride ldd (columns N..ldd-1
// zero-filled so the padded tensor can feed the dense wgrad/dgrad GEMMs directly).
// ---------------------------------------------------------------------------------------------
// `go` (device scalar, nullable): upstream gradient of the mean loss, multiplied into the stored
// gradient - backward recomputes the row statistics instead of keeping a [B, N] buffer alive and
// scaling it by an elementwise kernel.  `loss_mean` + `ticket` (nullable pair): the block that
// finishes last sums the per-row losses in a fixed order and stores their mean - no separate
// reduction launch, bit-reproducible (placement-independent hand-off: agent-scope release by every
// block before the ticket, agent-scope acquire by the last arriver; the ticket is left at 0).
__global__ void __launch_bounds__(256) softmax_xent_kernel(const bf16_t* __restrict__ logits, const long long* __restrict__ labels,
                                                          float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits,
                                                          int B, int N, int ldl, int ldd, float grad_scale,
                                                          const float* __restrict__ go, float* __restrict__ loss_mean,
                                                          unsigned* __restrict__ ticket) {
    __shared__ int last_s;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < B) {
        const bf16_t* lp = logits + (long long)row * ldl;
        float mx = -3.0e38f;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, bf2f(lp[n]));
        mx = wave_max(mx);
        float sum = 0.f;
        for (int n = lane; n < N; n += 64) sum += __expf(bf2f(lp[n]) - mx);
        sum = wave_sum(sum);
        const int label = (int)labels[row];
        const float lse = mx + __logf(sum);
        if (lane == 0 && loss_rows) loss_rows[row] = (label >= 0 && label < N) ? lse - bf2f(lp[label]) : 0.f;
        if (dlogits) {
            bf16_t* dp = dlogits + (long long)row * ldd;
            const float inv = 1.f / sum;
            const float gs = go ? grad_scale * go[0] : grad_scale;
            for (int n = lane; n < ldd; n += 64) {
                float g = 0.f;
                if (n < N) g = (__expf(bf2f(lp[n]) - mx) * inv - (n == label ? 1.f : 0.f)) * gs;
                dp[n] = f2bf(g);
            }
        }
    }
    if (!loss_mean) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (t == gridDim.x - 1) ? 1 : 0;
        if (last_s) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_s) return;
    if (threadIdx.x < 64) {   // one wave, fixed order: lane-strided partial sums, then the butterfly
        float s = 0.f;
        for (int r = lane; r < B; r += 64) s += loss_rows[r];
        s = wave_sum(s);
        if (lane == 0) {
            loss_mean[0] = s / (float)B;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void __launch_bounds__(256) bias_add_kernel(bf16_t* __restrict__ y, const float* __restrict__ bias, long long nchunks, int cchunks) {
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        int c = (int)(i % cchunks) * 8;
        bf16x8 v = *reinterpret_cast<bf16x8*>(y + (i << 3));
        float b[8];
        load8f(bias + c, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + b[e]);
        *reinterpret_cast<bf16x8*>(y + (i << 3)) = v;
    }
}

__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t* __restrict__ a, float* __restrict__ out, int M, int N, int ld,
                                                     int accumulate) {
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = ty;
        for (; m + 24 < M; m += 32) {
            s0 += bf2f(a[(long long)m * ld + n]);
            s1 += bf2f(a[(long long)(m + 8) * ld + n]);
            s2 += bf2f(a[(long long)(m + 16) * ld + n]);
            s3 += bf2f(a[(long long)(m + 24) * ld + n]);
        }
        for (; m < M; m += 8) s0 += bf2f(a[(long long)m * ld + n]);
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) s += red[l][tx];
        out[n] = accumulate ? out[n] + s : s;
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward by linearity, weight preparation.  For the BatchNorm behind a 1x1 conv
// (y = z W^T, dy = ca*g + cb*y + cc per channel c) the input gradient is
//   dz = g (diag(ca) W) + z (W^T diag(cb) W) + 1 (cc^T W)
// From the transposed filter wt[k][c] = W[c][k] (row k = conv input channel, ld elements apart):
//   b1[k][c] = wt[k][c]*ca[c]   (first-segment weights of the two-segment GEMM)
//   a2[k][c] = wt[k][c]*cb[c]   (left factor of Q = a2 wt^T)
//   bias[k]  = sum_c wt[k][c]*cc[c]
// One block per row k, 8 channels per thread (C <= 2048).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_lin_prep_kernel(const bf16_t* __restrict__ wt, int ld, const float* __restrict__ coef,
                                                         int C, bf16_t* __restrict__ b1, bf16_t* __restrict__ a2,
                                                         float* __restrict__ bias) {
    __shared__ float red[4];
    const int k = blockIdx.x, c = threadIdx.x * 8;
    float acc = 0.f;
    if (c < C) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wt + (long long)k * ld + c);
        float ca[8], cb[8], cc[8];
        load8f(coef + c, ca); load8f(coef + C + c, cb); load8f(coef + 2 * C + c, cc);
        bf16x8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf2f(w[e]);
            o1[e] = f2bf(f * ca[e]); o2[e] = f2bf(f * cb[e]);
            acc += f * cc[e];
        }
        *reinterpret_cast<bf16x8*>(b1 + (long long)k * C + c) = o1;
        *reinterpret_cast<bf16x8*>(a2 + (long long)k * C + c) = o2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bias[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The same preparation AND Q = a2 wt^T (bf16 [rows][rows], the second-segment weights) in ONE launch.  The two
// launches it replaces (the kernel above, then a rows x rows GEMM over C on 1-4 workgroups of the conv
// kernel: 12-39 us of a latency-bound loop) sat on the dependent chain in front of every block's first
// backward-data kernel.  One workgroup per 16 x 16 tile of Q: its four waves split the C axis in steps of
// 32 (one MFMA each, A = wt rows k scaled by cb and rounded to bf16 as a2 was, B = wt rows j) and meet in
// LDS; the workgroups of the first tile column also write their 16 rows of b1 and bias.
__global__ void __launch_bounds__(256) bn_lin_prep_q_kernel(const bf16_t* __restrict__ wt, int ld, const float* __restrict__ coef,
                                                           int C, int rows, bf16_t* __restrict__ b1, bf16_t* __restrict__ q,
                                                           float* __restrict__ bias) {
    __shared__ float red[4][64][4];
    __shared__ float bred[4][16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int r = lane & 15, g = lane >> 4;
    const bool first = blockIdx.y == 0;
    const bf16_t* wk = wt + (long long)(k0 + r) * ld;
    const bf16_t* wj = wt + (long long)(j0 + r) * ld;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bacc = 0.f;
    for (int c = wid * 32 + g * 8; c < C; c += 128) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wk + c);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wj + c);
        float cb[8];
        load8f(coef + C + c, cb);
        bf16x8 a2;
#pragma unroll
        for (int e = 0; e < 8; ++e) a2[e] = f2bf(bf2f(a[e]) * cb[e]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc, 0, 0, 0);
        if (first) {
            float ca[8], cc[8];
            load8f(coef + c, ca); load8f(coef + 2 * C + c, cc);
            bf16x8 o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(a[e]);
                o1[e] = f2bf(f * ca[e]);
                bacc += f * cc[e];
            }
            *reinterpret_cast<bf16x8*>(b1 + (long long)(k0 + r) * C + c) = o1;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][lane][e] = acc[e];
    if (first) {
        bacc += __shfl_xor(bacc, 16, 64);
        bacc += __shfl_xor(bacc, 32, 64);
        if (lane < 16) bred[wid][lane] = bacc;
    }
    __syncthreads();
    if (wid == 0) {
        // D: lane holds rows 4g .. 4g+3 (k), column r (j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (red[0][lane][e] + red[1][lane][e]) + (red[2][lane][e] + red[3][lane][e]);
            q[(long long)(k0 + 4 * g + e) * rows + j0 + r] = f2bf(v);
        }
        if (first && lane < 16) bias[k0 + lane] = (bred[0][lane] + bred[1][lane]) + (bred[2][lane] + bred[3][lane]);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight flip-transpose for backward-data, all tensors in one launch:
//   dst[c][R-1-r][S-1-s][k] = src[k][r][s][c]   (k padded with zeros up to Kpad)
// One block moves a 64(k) x 64(c) tile of one tap through LDS with 16-byte global accesses on
// both sides (25.5 M elements of ResNet-50 per optimizer step: ~100 MB of traffic).
// ---------------------------------------------------------------------------------------------
struct WtEntry { long long src, dst; int K, R, S, C, Kpad, block0; };

__global__ void __launch_bounds__(256) weight_flip_transpose_kernel(const bf16_t* __restrict__ base_src, bf16_t* __restrict__ base_dst,
                                                                   const WtEntry* __restrict__ table, int nentries) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];   // 144-byte rows: 16-byte aligned, 36 banks apart
    int lo = 0, hi = nentries - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= bid) lo = mid; else hi = mid - 1;
    }
    const WtEntry e = table[lo];
    int t = bid - e.block0;
    const int ct = (e.C + 63) / 64, kt = (e.Kpad + 63) / 64;
    const int tc = t % ct; t /= ct;
    const int tk = t % kt; t /= kt;
    const int s = t % e.S, r = t / e.S;
    const bf16_t* src = base_src + e.src;
    bf16_t* dst = base_dst + e.dst;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + i * 256, row = idx >> 3, chn = idx & 7;
        const int k = tk * 64 + row, c = tc * 64 + chn * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2bf(0.f);
        if (k < e.K && c < e.C) v = *reinterpret_cast<const bf16x8*>(src + (((long long)k * e.R + r) * e.S + s) * e.C + c);
        *reinterpret_cast<bf16x8*>(&tile[row][chn * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + i * 256, crow = idx >> 3, kch = idx & 7;
        const int c = tc * 64 + crow, k0 = tk * 64 + kch * 8;
        if (c < e.C && k0 < e.Kpad) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[kch * 8 + j][crow];
            *reinterpret_cast<bf16x8*>(dst + (((long long)c * e.R + (e.R - 1 - r)) * e.S + (e.S - 1 - s)) * e.Kpad + k0) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stem input packing: NHWC (C <= 4) bf16 -> [B][H][W + 8][4] with 3 zero pixels on the left, 5 on
// the right and zero-filled extra channels, so one filter row of the 7x7/2 stem conv is one
// contiguous, 16-byte-aligned 64-byte run (8 taps x 4 channels) for every output column.
__global__ void __launch_bounds__(256) stem_pack_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H,
                                                       int W, int C, long long xs_b, long long xs_h, long long xs_w,
                                                       long long xs_c) {
    const int Wp = W + 8;
    long long total = (long long)B * H * Wp;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int wp = (int)(i % Wp);
        long long row = i / Wp;
        int wi = wp - 3;
        bf16x4 v = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        if (wi >= 0 && wi < W) {
            long long b = row / H, h = row - b * H;
            const bf16_t* p = x + b * xs_b + h * xs_h + (long long)wi * xs_w;
            for (int c = 0; c < C; ++c) v[c] = p[c * xs_c];
        }
        *reinterpret_cast<bf16x4*>(out + (i << 2)) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K21: rank of the label's logit within its row (0 = it is the maximum): one wave per row counts
// the logits that beat it — larger, or equal with a lower index (the order torch.topk reports) —
// so top-k correctness for every k is `rank < k` without sorting anything.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) label_rank_kernel(const bf16_t* __restrict__ logits, const long long* __restrict__ labels,
                                                        int* __restrict__ rank, int B, int N, int ld) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const long long t = labels[row];
    if (t < 0 || t >= N) { if (lane == 0) rank[row] = N; return; }
    const bf16_t* p = logits + (long long)row * ld;
    const float vt = bf2f(p[t]);
    int cnt = 0;
    for (int i = lane; i < N; i += 64) {
        float v = bf2f(p[i]);
        cnt += (v > vt) | ((v == vt) & (i < (int)t));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) rank[row] = cnt;
}

}  // namespace

#define CHK(cond, code) do { if (!(cond)) return (code); } while (0)

static inline bool pow2_chunks_ok(int C) {
    if (C <= 0 || (C & 7)) return false;
    int cch = C >> 3;
    return cch <= 256 && (256 % cch) == 0;
}

extern "C" int ddl_bn_stats_partial(const void* y, void* partial, long long M, int C, int rows_per_tile, hipStream_t stream) {
    CHK(y && partial && M > 0 && rows_per_tile > 0 && pow2_chunks_ok(C), DDL_EINVAL);
    CHK(ddl_aligned16(y), DDL_EALIGN);
    int T = (int)((M + rows_per_tile - 1) / rows_per_tile);
    DDL_KLAUNCH(bn_stats_partial_kernel, dim3(T), dim3(256), 0, stream, (const bf16_t*)y, (float*)partial, M, C, rows_per_tile);
    return ddl_launch_status();
}

// Partial tables of at most this many rows are summed by the finalize kernel itself (one launch); longer ones are
// folded by many CUs first (two launches).  DDL_BN_DIRECT_MAX overrides (sweep in profiles/r3_notes.md).
extern "C" int ddl_bn_direct_max() {
    static const int v = getenv("DDL_BN_DIRECT_MAX") ? atoi(getenv("DDL_BN_DIRECT_MAX")) : 512;  // tuning knob
    return v < 1 ? 1 : v;
}

// workspace (doubles) needed by ddl_bn_fwd_finalize for T tiles of C channels
extern "C" long long ddl_bn_fold_slices(int T) {
    // ~64 tiles per slice, at most 32 slices — 128 for the longest tables (the stem's 25088 tiles
    // of 64 channels gave a 1 x 32 grid that took 33 us)
    int P = T / 64;
    if (P < 1) P = 1;
    const int cap = T >= 16384 ? 128 : 32;
    if (P > cap) P = cap;
    return P;
}

extern "C" int ddl_bn_fwd_finalize(const void* partial, int T, int C, long long M, int rows_per_tile, const void* gamma,
                                   const void* beta, void* rmean, void* rvar, float momentum, float eps, void* mean,
                                   void* invstd, void* scale, void* shift, void* workspace, hipStream_t stream) {
    CHK(partial && gamma && beta && mean && invstd && scale && shift && T > 0 && C > 0 && M > 0, DDL_EINVAL);
    CHK(workspace || T <= ddl_bn_direct_max(), DDL_EINVAL);
    CHK((long long)T * rows_per_tile >= M && (long long)(T - 1) * rows_per_tile < M, DDL_EINVAL);
    CHK((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, DDL_EALIGN);
    if (T <= ddl_bn_direct_max()) {
        DDL_KLAUNCH(bn_fwd_finalize_direct_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, (const float*)partial, T, C,
                           M, (const float*)gamma, (const float*)beta, (float*)rmean, (float*)rvar, momentum, eps,
                           (float*)mean, (float*)invstd, (float*)scale, (float*)shift);
        return ddl_launch_status();
    }
    const int P = (int)ddl_bn_fold_slices(T);
    static const int one = getenv("DDL_BN_ONE_LAUNCH") ? atoi(getenv("DDL_BN_ONE_LAUNCH")) : 0;   // A/B knob: measured slower, profiles/r2_notes.md
    int* tickets = one && C <= 4096 ? bn_tickets() : nullptr;
    if (tickets) {
        BnFinishArgs a = {};
        a.M = M; a.gamma = (const float*)gamma; a.beta = (const float*)beta; a.rmean = (float*)rmean; a.rvar = (float*)rvar;
        a.momentum = momentum; a.eps = eps; a.mean_out = (float*)mean; a.invstd_out = (float*)invstd;
        a.scale = (float*)scale; a.shift = (float*)shift;
        DDL_KLAUNCH((bn_fold_finish_kernel<double, 0>), dim3((C + 15) / 16, P), dim3(256), 0, stream, (const float*)partial,
                           T, C, P, (double*)workspace, tickets, a);
        return ddl_launch_status();
    }
    DDL_KLAUNCH(bn_stats_fold_kernel<double>, dim3((C + 15) / 16, P), dim3(256), 0, stream, (const float*)partial, T, C, P,
                       (double*)workspace);
    DDL_KLAUNCH(bn_fwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const double*)workspace, P, C, M,
                       (const float*)gamma, (const float*)beta, (float*)rmean, (float*)rvar, momentum, eps,
                       (float*)mean, (float*)invstd, (float*)scale, (float*)shift);
    return ddl_launch_status();
}

extern "C" int ddl_bn_eval_coeffs(const void* gamma, const void* beta, const void* rmean, const void* rvar, float eps, int C,
                                  void* scale, void* shift, hipStream_t stream) {
    CHK(gamma && beta && rmean && rvar && scale && shift && C > 0, DDL_EINVAL);
    DDL_KLAUNCH(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)gamma,
                       (const float*)beta, (const float*)rmean, (const float*)rvar, eps, C, (float*)scale, (float*)shift);
    return ddl_launch_status();
}

extern "C" int ddl_bn_apply_fwd_ex2(const void* y, const void* scale, const void* shift, const void* res,
                                    const void* rscale, const void* rshift, void* z, void* bits,
                                    long long M, int C, int relu, hipStream_t stream) {
    CHK(y && scale && shift && z && M > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK((rscale == nullptr) == (rshift == nullptr) && (!rscale || res), DDL_EINVAL);
    CHK(ddl_aligned16(y) && ddl_aligned16(z) && ddl_aligned16(scale) && ddl_aligned16(shift) && (!res || ddl_aligned16(res)), DDL_EALIGN);
    CHK(!rscale || (ddl_aligned16(rscale) && ddl_aligned16(rshift)), DDL_EALIGN);
    CHK(!rscale || C <= 4096, DDL_EINVAL);   // the two-BatchNorm variant stages 4 * C floats in LDS
    long long nchunks = M * (C >> 3);
    dim3 grid(ddl_stream_grid(nchunks, 256)), block(256);
    const bool nt = M * C * 2 >= nt_threshold_bytes();
    const int rmode = !res ? 0 : (rscale ? 2 : 1);
#define L2(R, S, N) DDL_KLAUNCH((bn_apply_fwd_kernel<R, S, N>), grid, block, (S == 2 ? 16 * (size_t)C : 0), stream, (const bf16_t*)y, (const float*)scale, (const float*)shift, (const bf16_t*)res, (const float*)rscale, (const float*)rshift, (bf16_t*)z, (unsigned char*)bits, nchunks, C >> 3)
#define L(R, S) do { if (nt) L2(R, S, true); else L2(R, S, false); } while (0)
#define LR(R) do { if (rmode == 2) L(R, 2); else if (rmode == 1) L(R, 1); else L(R, 0); } while (0)
    if (relu) LR(true); else LR(false);
#undef LR
#undef L2
#undef L
    return ddl_launch_status();
}

extern "C" int ddl_bn_apply_fwd_ex(const void* y, const void* scale, const void* shift, const void* res,
                                   const void* rscale, const void* rshift, void* z,
                                   long long M, int C, int relu, hipStream_t stream) {
    return ddl_bn_apply_fwd_ex2(y, scale, shift, res, rscale, rshift, z, nullptr, M, C, relu, stream);
}

extern "C" int ddl_bn_apply_fwd(const void* y, const void* scale, const void* shift, const void* res, void* z,
                                long long M, int C, int relu, hipStream_t stream) {
    return ddl_bn_apply_fwd_ex(y, scale, shift, res, nullptr, nullptr, z, M, C, relu, stream);
}

extern "C" int ddl_bn_bwd_blocks(long long M) {
    // 32 rows per block at least, up to 1024 blocks: the 7x7 layers (M = 12544) ran 49 blocks of
    // 256 rows and took 142 us for 50 MB (rocprof, profiles/r1_pmc_training_step.md)
    long long g = (M + 31) / 32;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

// mask: 0 none, 1 from z, 2 recomputed from y with fcoef = [scale | shift] (2*C floats)
extern "C" int ddl_bn_bwd_reduce(const void* dz, const void* y, const void* z, const void* fcoef, void* partial, long long M,
                                 int C, int mask, int G, hipStream_t stream) {
    CHK(dz && y && partial && M > 0 && G > 0 && pow2_chunks_ok(C), DDL_EINVAL);
    CHK(mask == 0 || (mask == 1 && z) || (mask == 2 && fcoef), DDL_EINVAL);
    CHK(ddl_aligned16(dz) && ddl_aligned16(y) && (!z || ddl_aligned16(z)) && (!fcoef || ddl_aligned16(fcoef)), DDL_EALIGN);
    long long rpb = (M + G - 1) / G;
#define L(MK) DDL_KLAUNCH(bn_bwd_reduce_kernel<MK>, dim3(G), dim3(256), 0, stream, (const bf16_t*)dz, (const bf16_t*)y, (const bf16_t*)z, (const float*)fcoef, (float*)partial, M, C, rpb)
    if (mask == 1) L(1); else if (mask == 2) L(2); else L(0);
#undef L
    return ddl_launch_status();
}

// `workspace` (optional, [ddl_bn_fold_slices(G)][2][C] floats): long partial tables (the per-tile
// partials of a fused conv epilogue) are first folded by many CUs.
extern "C" int ddl_bn_bwd_finalize(const void* partial, int G, int C, long long M, const void* gamma, const void* mean,
                                   const void* invstd, void* dgamma, void* dbeta, void* coef, int accumulate,
                                   void* workspace, hipStream_t stream) {
    CHK(partial && gamma && mean && invstd && dgamma && dbeta && coef && G > 0 && C > 0 && M > 0, DDL_EINVAL);
    if (workspace && G > ddl_bn_direct_max()) {
        const int P = (int)ddl_bn_fold_slices(G);
        static const int one = getenv("DDL_BN_ONE_LAUNCH") ? atoi(getenv("DDL_BN_ONE_LAUNCH")) : 0;   // A/B knob: measured slower, profiles/r2_notes.md
        int* tickets = one && C <= 4096 ? bn_tickets() : nullptr;
        if (tickets) {
            BnFinishArgs a = {};
            a.M = M; a.gamma = (const float*)gamma; a.mean = (const float*)mean; a.invstd = (const float*)invstd;
            a.dgamma = (float*)dgamma; a.dbeta = (float*)dbeta; a.coef = (float*)coef; a.accumulate = accumulate;
            DDL_KLAUNCH((bn_fold_finish_kernel<float, 1>), dim3((C + 15) / 16, P), dim3(256), 0, stream, (const float*)partial,
                               G, C, P, (float*)workspace, tickets + 256, a);
            return ddl_launch_status();
        }
        DDL_KLAUNCH(bn_stats_fold_kernel<float>, dim3((C + 15) / 16, P), dim3(256), 0, stream, (const float*)partial, G, C,
                           P, (float*)workspace);
        if (ddl_launch_status() != DDL_OK) return DDL_ELAUNCH;
        partial = workspace;
        G = P;
    }
    DDL_KLAUNCH(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, (const float*)partial, G, C, M,
                       (const float*)gamma, (const float*)mean, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef, accumulate);
    return ddl_launch_status();
}

extern "C" int ddl_bn_bwd_apply(const void* dz, const void* y, const void* z, const void* fcoef, const void* coef, void* dy,
                                void* gout, long long M, int C, int mask, hipStream_t stream) {
    CHK(dz && y && coef && dy && M > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK(mask == 0 || (mask == 1 && z) || (mask == 2 && fcoef), DDL_EINVAL);
    CHK(C <= 3072, DDL_EINVAL);   // coefficients are staged in LDS: up to 5 * C floats
    CHK(ddl_aligned16(dz) && ddl_aligned16(y) && ddl_aligned16(dy) && ddl_aligned16(coef) && (!z || ddl_aligned16(z)) &&
        (!gout || ddl_aligned16(gout)) && (!fcoef || ddl_aligned16(fcoef)), DDL_EALIGN);
    long long nchunks = M * (C >> 3);
    dim3 grid(ddl_stream_grid(nchunks, 256)), block(256);
    const bool nt = M * C * 2 >= nt_threshold_bytes();
#define L2(MK, G, N) DDL_KLAUNCH((bn_bwd_apply_kernel<MK, G, N>), grid, block, (MK == 2 ? 20 : 12) * (size_t)C, stream, (const bf16_t*)dz, (const bf16_t*)y, (const bf16_t*)z, (const float*)fcoef, (const float*)coef, (bf16_t*)dy, (bf16_t*)gout, nchunks, C >> 3)
#define L(MK, G) do { if (nt) L2(MK, G, true); else L2(MK, G, false); } while (0)
    if (mask == 1) { if (gout) L(1, true); else L(1, false); }
    else if (mask == 2) { if (gout) L(2, true); else L(2, false); }
    else { if (gout) L(0, true); else L(0, false); }
#undef L
#undef L2
    return ddl_launch_status();
}

static int maxpool_fwd_launch(const void* x, const void* scale, const void* shift, void* out, void* idx, int B, int H, int W,
                              int C, hipStream_t stream) {
    CHK(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(x) && ddl_aligned16(out) && (!idx || (reinterpret_cast<uintptr_t>(idx) & 7) == 0), DDL_EALIGN);
    CHK((scale == nullptr) == (shift == nullptr) && (!scale || (ddl_aligned16(scale) && ddl_aligned16(shift))), DDL_EINVAL);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long long total = (long long)B * Ho * Wo * (C >> 3);
    CHK((long long)B * H * W * (C >> 3) < (1LL << 32), DDL_EINVAL);
    dim3 grid(ddl_stream_grid(total, 256)), block(256);
    if (scale) DDL_KLAUNCH(maxpool_fwd_kernel<true>, grid, block, 0, stream, (const bf16_t*)x, (const float*)scale,
                                  (const float*)shift, (bf16_t*)out, (unsigned char*)idx, B, H, W, C, Ho, Wo);
    else DDL_KLAUNCH(maxpool_fwd_kernel<false>, grid, block, 0, stream, (const bf16_t*)x, (const float*)nullptr,
                            (const float*)nullptr, (bf16_t*)out, (unsigned char*)idx, B, H, W, C, Ho, Wo);
    return ddl_launch_status();
}

extern "C" int ddl_maxpool3x3s2_fwd(const void* x, void* out, void* idx, int B, int H, int W, int C, hipStream_t stream) {
    return maxpool_fwd_launch(x, nullptr, nullptr, out, idx, B, H, W, C, stream);
}

// out = maxpool3x3s2(relu(x*scale + shift)) with x the pre-BatchNorm tensor (stem fusion)
extern "C" int ddl_bn_relu_maxpool3x3s2_fwd(const void* x, const void* scale, const void* shift, void* out, void* idx, int B,
                                            int H, int W, int C, hipStream_t stream) {
    CHK(scale && shift && idx, DDL_EINVAL);
    return maxpool_fwd_launch(x, scale, shift, out, idx, B, H, W, C, stream);
}

// Stem backward passes over the pre-BatchNorm tensor y [B,H,W,C]; dpool/idx are [B,Ho,Wo,C].
// apply = 0: partial [G][2][C] sums (G = ddl_bn_bwd_blocks(B*H*W)); apply = 1: dy from coef [3][C].
extern "C" int ddl_stem_pool_bn_bwd(const void* dpool, const void* idx, const void* y, const void* fcoef, const void* coef,
                                    void* partial, void* dy, int B, int H, int W, int C, int G, int apply, hipStream_t stream) {
    CHK(dpool && idx && y && fcoef && B > 0 && H > 0 && W > 0 && G > 0 && pow2_chunks_ok(C), DDL_EINVAL);
    CHK(apply ? (coef && dy) : (partial != nullptr), DDL_EINVAL);
    CHK(ddl_aligned16(dpool) && ddl_aligned16(y) && ddl_aligned16(fcoef) && (reinterpret_cast<uintptr_t>(idx) & 7) == 0 &&
        (!coef || ddl_aligned16(coef)) && (!dy || ddl_aligned16(dy)), DDL_EALIGN);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long long M = (long long)B * H * W;
    CHK(M < (1LL << 31), DDL_EINVAL);
    const long long NP = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);   // 2x2 pixel patches
    long long rpb = (NP + G - 1) / G;
    if (apply) DDL_KLAUNCH(stem_pool_bn_bwd_kernel<true>, dim3(G), dim3(256), 0, stream, (const bf16_t*)dpool,
                                  (const unsigned char*)idx, (const bf16_t*)y, (const float*)fcoef, (const float*)coef,
                                  (float*)nullptr, (bf16_t*)dy, B, H, W, C, Ho, Wo, rpb);
    else DDL_KLAUNCH(stem_pool_bn_bwd_kernel<false>, dim3(G), dim3(256), 0, stream, (const bf16_t*)dpool,
                            (const unsigned char*)idx, (const bf16_t*)y, (const float*)fcoef, (const float*)nullptr,
                            (float*)partial, (bf16_t*)nullptr, B, H, W, C, Ho, Wo, rpb);
    return ddl_launch_status();
}

extern "C" int ddl_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int B, int H, int W, int C, hipStream_t stream) {
    CHK(dy && idx && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(dy) && ddl_aligned16(dx) && (reinterpret_cast<uintptr_t>(idx) & 7) == 0, DDL_EALIGN);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long long total = (long long)B * H * W * (C >> 3);
    CHK(total < (1LL << 32), DDL_EINVAL);
    DDL_KLAUNCH(maxpool_bwd_kernel, dim3(ddl_stream_grid(total, 256)), dim3(256), 0, stream, (const bf16_t*)dy,
                       (const unsigned char*)idx, (bf16_t*)dx, B, H, W, C, Ho, Wo);
    return ddl_launch_status();
}

extern "C" int ddl_avgpool_fwd(const void* x, void* out, int B, int HW, int C, hipStream_t stream) {
    CHK(x && out && B > 0 && HW > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(x) && ddl_aligned16(out), DDL_EALIGN);
    long long total = (long long)B * (C >> 3);
    DDL_KLAUNCH(avgpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, B, HW, C);
    return ddl_launch_status();
}

extern "C" int ddl_avgpool_bwd(const void* dy, void* dx, int B, int HW, int C, hipStream_t stream) {
    CHK(dy && dx && B > 0 && HW > 0 && C > 0 && (C & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(dy) && ddl_aligned16(dx), DDL_EALIGN);
    long long total = (long long)B * HW * (C >> 3);
    DDL_KLAUNCH(avgpool_bwd_kernel, dim3(ddl_stream_grid(total, 256)), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, B, HW, C);
    return ddl_launch_status();
}

extern "C" int ddl_softmax_xent_ex(const void* logits, const void* labels, void* loss_rows, void* dlogits, int B, int N, int ldl,
                                   int ldd, float grad_scale, const void* go, void* loss_mean, void* ticket, hipStream_t stream) {
    CHK(logits && labels && B > 0 && N > 0 && ldl >= N && (!dlogits || ldd >= N), DDL_EINVAL);
    CHK(loss_rows || dlogits, DDL_EINVAL);
    CHK(!loss_mean || (ticket && loss_rows), DDL_EINVAL);
    DDL_KLAUNCH(softmax_xent_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, (const bf16_t*)logits,
                       (const long long*)labels, (float*)loss_rows, (bf16_t*)dlogits, B, N, ldl, ldd, grad_scale,
                       (const float*)go, (float*)loss_mean, (unsigned*)ticket);
    return ddl_launch_status();
}
extern "C" int ddl_softmax_xent(const void* logits, const void* labels, void* loss_rows, void* dlogits, int B, int N, int ldl,
                                int ldd, float grad_scale, hipStream_t stream) {
    CHK(loss_rows, DDL_EINVAL);
    return ddl_softmax_xent_ex(logits, labels, loss_rows, dlogits, B, N, ldl, ldd, grad_scale, nullptr, nullptr, nullptr, stream);
}
extern "C" int ddl_bias_add(void* y, const void* bias, long long M, int N, hipStream_t stream) {
    CHK(y && bias && M > 0 && N > 0 && (N & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(y) && ddl_aligned16(bias), DDL_EALIGN);
    long long nchunks = M * (N >> 3);
    DDL_KLAUNCH(bias_add_kernel, dim3(ddl_stream_grid(nchunks, 256)), dim3(256), 0, stream, (bf16_t*)y, (const float*)bias, nchunks, N >> 3);
    return ddl_launch_status();
}

extern "C" int ddl_colsum_ex(const void* a, void* out, int M, int N, int ld, int accumulate, hipStream_t stream) {
    CHK(a && out && M > 0 && N > 0 && ld >= N, DDL_EINVAL);
    DDL_KLAUNCH(colsum_kernel, dim3((N + 31) / 32), dim3(256), 0, stream, (const bf16_t*)a, (float*)out, M, N, ld, accumulate);
    return ddl_launch_status();
}
extern "C" int ddl_colsum(const void* a, void* out, int M, int N, int ld, hipStream_t stream) {
    return ddl_colsum_ex(a, out, M, N, ld, 0, stream);
}

// table entries are 6 x int64 + ... : see WtEntry; `table` is a device pointer prepared by the host
extern "C" int ddl_weight_flip_transpose(const void* base_src, void* base_dst, const void* table, int nentries,
                                         int total_blocks, hipStream_t stream) {
    CHK(base_src && base_dst && table && nentries > 0 && total_blocks > 0, DDL_EINVAL);
    DDL_KLAUNCH(weight_flip_transpose_kernel, dim3(total_blocks), dim3(256), 0, stream, (const bf16_t*)base_src,
                       (bf16_t*)base_dst, (const WtEntry*)table, nentries);
    return ddl_launch_status();
}

extern "C" int ddl_wt_entry_bytes() { return (int)sizeof(WtEntry); }

// Fast form for the usual input: 3 channels, NHWC-contiguous, W % 4 == 0.  A thread moves four pixels:
// three aligned 8-byte loads (4 x 3 bf16) become four 8-byte pixels; one extra thread per row end writes
// the zero pixels.  (The generic kernel issues three 2-byte loads and two 64-bit divisions per pixel.)
__global__ void __launch_bounds__(256) stem_pack3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long rows, int W) {
    const int G = (W >> 2) + 2;                       // groups per row: left padding, W / 4 pixel quads, right padding
    const long long total = rows * G;
    const long long stride = (long long)gridDim.x * blockDim.x;
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long row = i / G;
        const int g = (int)(i - row * G);
        u16x4* o = reinterpret_cast<u16x4*>(out + (row * (W + 8)) * 4);     // pixels of this row, 8 bytes each
        const u16x4 zero = {0, 0, 0, 0};
        if (g == 0) { o[0] = zero; o[1] = zero; o[2] = zero; continue; }
        if (g == G - 1) {
#pragma unroll
            for (int k = 0; k < 5; ++k) o[W + 3 + k] = zero;
            continue;
        }
        const int wi = (g - 1) << 2;
        const u16x4* src = reinterpret_cast<const u16x4*>(x + (row * W + wi) * 3);   // 24 bytes = 12 bf16, 8-byte aligned
        const u16x4 a = src[0], b = src[1], c = src[2];
        const u16x4 p0 = {a[0], a[1], a[2], 0}, p1 = {a[3], b[0], b[1], 0}, p2 = {b[2], b[3], c[0], 0}, p3 = {c[1], c[2], c[3], 0};
        o[wi + 3] = p0; o[wi + 4] = p1; o[wi + 5] = p2; o[wi + 6] = p3;
    }
}

// x: logical [B][C][H][W] bf16 with element strides (xs_b, xs_c, xs_h, xs_w); out: [B][H][W+8][4]
extern "C" int ddl_stem_pack(const void* x, void* out, int B, int H, int W, int C, long long xs_b, long long xs_h,
                             long long xs_w, long long xs_c, hipStream_t stream) {
    CHK(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, DDL_EINVAL);
    CHK((reinterpret_cast<uintptr_t>(out) & 15) == 0, DDL_EALIGN);
    if (C == 3 && (W & 3) == 0 && xs_c == 1 && xs_w == 3 && xs_h == 3LL * W && xs_b == 3LL * W * H &&
        (reinterpret_cast<uintptr_t>(x) & 7) == 0) {
        const long long rows = (long long)B * H;
        DDL_KLAUNCH(stem_pack3_kernel, dim3(ddl_stream_grid(rows * ((W >> 2) + 2), 256)), dim3(256), 0, stream,
                           (const bf16_t*)x, (bf16_t*)out, rows, W);
        return ddl_launch_status();
    }
    long long total = (long long)B * H * (W + 8);
    DDL_KLAUNCH(stem_pack_kernel, dim3(ddl_stream_grid(total, 256)), dim3(256), 0, stream, (const bf16_t*)x,
                       (bf16_t*)out, B, H, W, C, xs_b, xs_h, xs_w, xs_c);
    return ddl_launch_status();
}

// rank[row] = number of logits that precede the label's in top-k order; N for an invalid label
extern "C" int ddl_label_rank(const void* logits, const void* labels, void* rank, int B, int N, int ld, hipStream_t stream) {
    CHK(logits && labels && rank && B > 0 && N > 0 && ld >= N, DDL_EINVAL);
    DDL_KLAUNCH(label_rank_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, (const bf16_t*)logits,
                       (const long long*)labels, (int*)rank, B, N, ld);
    return ddl_launch_status();
}


extern "C" int ddl_bn_lin_prep(const void* wt, int ld, const void* coef, int C, int rows, void* b1, void* a2, void* bias,
                               hipStream_t stream) {
    CHK(wt && coef && b1 && a2 && bias && C > 0 && C <= 2048 && (C & 7) == 0 && rows > 0 && ld >= C && (ld & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(wt) && ddl_aligned16(coef) && ddl_aligned16(b1) && ddl_aligned16(a2), DDL_EALIGN);
    DDL_KLAUNCH(bn_lin_prep_kernel, dim3(rows), dim3(256), 0, stream, (const bf16_t*)wt, ld, (const float*)coef, C,
                       (bf16_t*)b1, (bf16_t*)a2, (float*)bias);
    return ddl_launch_status();
}

extern "C" int ddl_bn_lin_prep_q(const void* wt, int ld, const void* coef, int C, int rows, void* b1, void* q, void* bias,
                                 hipStream_t stream) {
    CHK(wt && coef && b1 && q && bias && C > 0 && C <= 2048 && (C & 31) == 0 && rows > 0 && (rows & 15) == 0 && ld >= C && (ld & 7) == 0, DDL_EINVAL);
    CHK(ddl_aligned16(wt) && ddl_aligned16(coef) && ddl_aligned16(b1) && ddl_aligned16(q), DDL_EALIGN);
    DDL_KLAUNCH(bn_lin_prep_q_kernel, dim3(rows / 16, rows / 16), dim3(256), 0, stream, (const bf16_t*)wt, ld, (const float*)coef, C, rows,
                       (bf16_t*)b1, (bf16_t*)q, (float*)bias);
    return ddl_launch_status();
}

// out[2][C] = sum over the T rows of partial[T][2][C] (one slice of the fold kernel)
extern "C" int ddl_bn_colsum_fold(const void* partial, int T, int C, void* out, hipStream_t stream) {
    CHK(partial && out && T > 0 && C > 0, DDL_EINVAL);
    DDL_KLAUNCH((bn_stats_fold_kernel<float>), dim3((C + 15) / 16, 1), dim3(256), 0, stream, (const float*)partial, T, C, 1,
                       (float*)out);
    return ddl_launch_status();
}
