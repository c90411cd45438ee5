// Streaming-kernel tuning probe for the BatchNorm apply kernels (memory-bound, 16-byte lanes):
// z = relu(y * scale[c] + shift[c]) over an [M, C] bf16 tensor, variants over grid size, loads in
// flight per thread and the non-temporal hint.  Build: hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) apply_kernel(const __bf16* __restrict__ y, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, __bf16* __restrict__ z,
                                                    long long nchunks, int cchunks) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nchunks; i += UNROLL * stride) {
        bf16x8 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const bf16x8* p = reinterpret_cast<const bf16x8*>(y) + i + u * stride;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int c = (int)((i + u * stride) % cchunks) * 8;
            f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(shift + c), b1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (__bf16)fmaxf(__builtin_fmaf((float)v[u][e], s0[e], b0[e]), 0.f);
                o[4 + e] = (__bf16)fmaxf(__builtin_fmaf((float)v[u][4 + e], s1[e], b1[e]), 0.f);
            }
            bf16x8* q = reinterpret_cast<bf16x8*>(z) + i + u * stride;
            if (NT) __builtin_nontemporal_store(o, q); else *q = o;
        }
    }
    for (; i < nchunks; i += stride) {
        bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
        int c = (int)(i % cchunks) * 8;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(__builtin_fmaf((float)v[e], scale[c + e], shift[c + e]), 0.f);
        reinterpret_cast<bf16x8*>(z)[i] = o;
    }
}

template <int U, bool NT>
float run(const __bf16* y, const float* sc, const float* sh, __bf16* z, long long nchunks, int cchunks, int grid, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((apply_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, y, sc, sh, z, nchunks, cchunks);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((apply_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, y, sc, sh, z, nchunks, cchunks);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const int Cs[2] = {256, 1024};
    const long long Ms[2] = {802816, 50176};
    for (int t = 0; t < 2; ++t) {
        const int C = Cs[t];
        const long long M = Ms[t], n = M * C, nchunks = n / 8;
        __bf16 *y, *z; float *sc, *sh;
        hipMalloc(&y, n * 2); hipMalloc(&z, n * 2); hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
        hipMemset(y, 0x3c, n * 2); hipMemset(sc, 0, C * 4); hipMemset(sh, 0, C * 4);
        printf("tensor [%lld, %d] bf16 = %.0f MB (read + write %.0f MB)\n", M, C, n * 2 / 1e6, n * 4 / 1e6);
        const int grids[6] = {1024, 2048, 4096, 8192, 16384, 65536};
        for (int gi = 0; gi < 6; ++gi) {
            int g = grids[gi];
            if ((long long)g * 256 > nchunks) g = (int)((nchunks + 255) / 256);
            float r[6];
            r[0] = run<1, false>(y, sc, sh, z, nchunks, C / 8, g, 20);
            r[1] = run<2, false>(y, sc, sh, z, nchunks, C / 8, g, 20);
            r[2] = run<4, false>(y, sc, sh, z, nchunks, C / 8, g, 20);
            r[3] = run<1, true>(y, sc, sh, z, nchunks, C / 8, g, 20);
            r[4] = run<2, true>(y, sc, sh, z, nchunks, C / 8, g, 20);
            r[5] = run<4, true>(y, sc, sh, z, nchunks, C / 8, g, 20);
            printf("  grid %6d: GB/s  u1 %5.0f  u2 %5.0f  u4 %5.0f | nt u1 %5.0f  u2 %5.0f  u4 %5.0f\n", g,
                   n * 4 / r[0] / 1e6, n * 4 / r[1] / 1e6, n * 4 / r[2] / 1e6, n * 4 / r[3] / 1e6, n * 4 / r[4] / 1e6, n * 4 / r[5] / 1e6);
        }
        hipFree(y); hipFree(z); hipFree(sc); hipFree(sh);
    }
    return 0;
}
