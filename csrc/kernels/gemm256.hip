// K1g: forward 1x1 / stride 1 convolution of the WIDE layers (stages 3-4 of a bottleneck net: 256..2048 channels
// on both sides) as a plain "NT" GEMM on a 256 x 256 tile with 8 waves and an LDS-DMA ring (gfx950):
//
//   y[m][n] = sum_k x[m][k] * w[n][k]          x: [M][K] pixels x input channels, w: [N][K], both k-contiguous
//
// The 4-wave 128 x 128 kernel (conv_igemm.hip) runs these layers at 500-650 TFLOP/s: a wave tile of 64 x 64 reads
// 512 bytes of LDS per MFMA (the CU's whole LDS bandwidth at full matrix rate), has one K tile in flight per
// workgroup and drains it (`vmcnt(0)`) at every barrier.  Built like the weight-gradient twin
// (conv_wgrad.hip::conv_wgrad256_kernel):
//   * 8 waves as 2 (pixel halves) x 4 (64-channel groups); a wave owns 128 pixels x 64 channels = 8 x 4 MFMA
//     fragments (128 accumulator registers): 32 MFMAs per 12 16-byte LDS reads (384 bytes per MFMA) and per barrier;
//   * NST LDS stages of 32 reduction elements (x [256][32] + w [256][32] = 32 KB), filled by LDS-DMA only (inline asm:
//     the compiler's wait-count pass never sees it), tile t+NST-1 issued before tile t is consumed, COUNTED
//     `s_waitcnt vmcnt` (never 0 inside the loop) + raw `s_barrier`;
//   * D rows = output channels, D columns = pixels (as in conv_igemm): a lane holds 4 consecutive channels of a
//     pixel per fragment, and the filter rows are dealt to the LDS image so that the two fragments of a pair give a
//     lane 8 CONSECUTIVE channels: the epilogue stores 16 bytes per lane straight from the rounded accumulators;
//   * BatchNorm statistics (sum y, sum y^2 of the bf16-rounded outputs) per 128-row half tile and channel: each wave
//     owns all 128 rows of its 64 channels, so a 16-lane fold finishes a table row - same [M / 128][2][N] table as
//     the tile kernel writes.
// XCD-aware tile order as everywhere (n fastest inside an XCD's run: the x panel is reused from its L2).
//
// MEASURED (profiles/r3_notes.md section 21) and therefore OFF by default (DDL_GEMM256=1 enables it): in isolation
// d1 (1024 -> 512 @14) 71.4 -> 67.1 us, c3 (1024 -> 256) 39.7 -> 40.3, c1 94 -> 101, c5 (256 -> 1024) 45 -> 58,
// d5 39 -> 43, d3 (98 tiles) 34 -> 54; hipBLASLt: c3 28.7, d1 52.1, d3 30.9, c5 48.9 us.  Unlike the weight gradients
// (reductions of 12 544-50 176 rows) these reductions are 8-32 iterations long: prologue and epilogue are a third of
// a tile, and with one workgroup per CU nothing overlaps them, while four 4-wave workgroups per CU overlap each
// other's phases for free.  An iteration takes ~2500 cycles against 1024 of matrix work per SIMD (DMA issue, fragment
// reads and multiplies of a wave are one serial chain with two waves per SIMD to hide it).
#include "common.h"
#include <cstdlib>

namespace {

struct Gemm256Params {
    const bf16_t* x;
    const bf16_t* w;
    bf16_t* y;
    float* stats;          // optional [stat_rows][2][N]
    long long x_bytes;
    unsigned w_bytes;
    int M, N, K;
    int tiles_m, tiles_n, stat_rows;
};

// byte offset of 16-byte chunk `ch` of row `row` in a [rows][32] bf16 LDS tile (the tile kernel's swizzle:
// 16 consecutive rows x one chunk column are conflict-free for ds_read_b128)
__device__ __forceinline__ int g256_off(int row, int ch) { return row * 64 + ((ch ^ ((0 - (row >> 2)) & 3)) << 4); }

template <bool STATS, int NST>
__global__ void __launch_bounds__(512) gemm256_fwd_kernel(const Gemm256Params p) {
    constexpr int BK = 32, HALF = 256 * BK * 2, STAGE = 2 * HALF;
    constexpr int FC = 4, FP = 8;                      // channel / pixel fragments of a wave
    constexpr unsigned INVALID = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fq = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int KT = p.K / BK;

    const long long x_first = (long long)m0 * p.K;
    long long x_left = p.x_bytes - x_first * 2;
    x_left = x_left < 0 ? 0 : (x_left > 0x7fffffffLL ? 0x7fffffffLL : x_left);
    const u32x4 xr = raw_rsrc(p.x + x_first, (unsigned)x_left), wr = raw_rsrc(p.w, p.w_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // DMA geometry: slot tid + 512 i of an operand image is row (slot / 4), physical chunk slot % 4; the swizzle is
    // applied on the source side.  LDS row rho of the filter image holds output channel perm(rho): inside a wave's
    // 64 rows, row 16 i + q <-> channel 32 (i / 2) + 8 (q / 4) + 4 (i % 2) + q % 4.
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = tid + i * 512, row = slot >> 2, chp = slot & 3;
        const int ch = chp ^ ((0 - (row >> 2)) & 3);
        a_voff[i] = (m0 + row < p.M) ? (unsigned)((row * p.K + ch * 8) * 2) : INVALID;
        const int rw = row & 63, i16 = rw >> 4, q = rw & 15;
        const int chan = (row & ~63) + 32 * (i16 >> 1) + 8 * (q >> 2) + 4 * (i16 & 1) + (q & 3);
        b_voff[i] = (n0 + chan < p.N) ? (unsigned)(((n0 + chan) * p.K + ch * 8) * 2) : INVALID;
    }
    // every tile is exactly 4 LDS-DMA instructions per thread (tiles past the end read as zeros): the counted waits rely on it
    auto fetch = [&](int kt, int stage) {
        const bool live = kt < KT;                                   // uniform
        const unsigned Ad = lds0 + (unsigned)(stage * STAGE + wid * 1024), Bd = Ad + HALF;
        const unsigned soff = live ? (unsigned)(kt * BK * 2) : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(xr, Ad + i * 8192, live ? a_voff[i] : INVALID, soff);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(wr, Bd + i * 8192, live ? b_voff[i] : INVALID, soff);
    };
    auto wait_tiles = [&](int tiles_in_flight) {
        if (tiles_in_flight >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (tiles_in_flight == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int xa_off[FP], wb_off[FC];
#pragma unroll
    for (int j = 0; j < FP; ++j) xa_off[j] = g256_off(wm * 128 + j * 16 + fr, fq);
#pragma unroll
    for (int i = 0; i < FC; ++i) wb_off[i] = HALF + g256_off(wn * 64 + i * 16 + fr, fq);

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) fetch(s0, s0);
    wait_tiles(NST - 2);
    __builtin_amdgcn_s_barrier();

    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < KT; ++kt) {
        fetch(kt + NST - 1, nxt);              // into the stage whose last reads retired before the previous barrier
        const unsigned char* St = smem + cur * STAGE;
        bf16x8 xa[FP], wb[FC];
#pragma unroll
        for (int i = 0; i < FC; ++i) wb[i] = *reinterpret_cast<const bf16x8*>(St + wb_off[i]);
#pragma unroll
        for (int j = 0; j < FP; ++j) xa[j] = *reinterpret_cast<const bf16x8*>(St + xa_off[j]);
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int i = 0; i < FC; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], xa[j], acc[i][j], 0, 0, 0);
        // tile kt+1 has landed (this wave's part; the barrier makes it everyone's) and every LDS read of this tile
        // has retired, so the next iteration may overwrite the stage
        wait_tiles(NST - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        nxt = cur;
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-filled tiles past the end

    // ---- epilogue: fragment pair (2a, 2a+1) of pixel fragment j = channels c8 .. c8+7 of pixel m0 + wm*128 + 16j + fr
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[a][e] = 0.f; s2[a][e] = 0.f; }
    const int c8 = n0 + wn * 64 + 8 * fq;
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int m = m0 + wm * 128 + j * 16 + fr;
        bf16_t* row = p.y + (long long)m * p.N + c8;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = f2bf(acc[2 * a][j][e]); v[4 + e] = f2bf(acc[2 * a + 1][j][e]); }
            if (STATS) {
                // (rows past the edge are exact zeros: their operand rows were zero-filled by the range check)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); s1[a][e] += f; s2[a][e] += f * f; }
            }
            if (m < p.M) *reinterpret_cast<bf16x8*>(row + 32 * a) = v;
        }
    }
    if (STATS) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[a][e] += __shfl_xor(s1[a][e], o, 64);
                    s2[a][e] += __shfl_xor(s2[a][e], o, 64);
                }
        const int srow = tile_m * 2 + wm;
        if (fr == 0 && srow < p.stat_rows) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float* t1 = p.stats + ((long long)srow * 2) * p.N + c8 + 32 * a;
                float* t2 = t1 + p.N;
                *reinterpret_cast<f32x4*>(t1) = f32x4{s1[a][0], s1[a][1], s1[a][2], s1[a][3]};
                *reinterpret_cast<f32x4*>(t1 + 4) = f32x4{s1[a][4], s1[a][5], s1[a][6], s1[a][7]};
                *reinterpret_cast<f32x4*>(t2) = f32x4{s2[a][0], s2[a][1], s2[a][2], s2[a][3]};
                *reinterpret_cast<f32x4*>(t2 + 4) = f32x4{s2[a][4], s2[a][5], s2[a][6], s2[a][7]};
            }
        }
    }
}

}  // namespace

// Shapes the kernel can compute: whole 256-channel tiles on the output side, 32-element reduction steps
extern "C" int ddl_gemm256_supported(long long M, int N, int K) {
    if (N <= 0 || N % 256 != 0 || K <= 0 || K % 32 != 0 || M <= 0 || M >= (1LL << 31) - 256) return 0;
    return (long long)N * K * 2 < 0x7fffffffLL && 256LL * K * 2 < 0x7fffffffLL;
}

// Shapes the conv dispatch hands to it (DDL_GEMM256=1; off by default, see the header): a reduction of at least
// DDL_GEMM256_MIN_K and at least DDL_GEMM256_MIN_TILES tiles (one workgroup per CU: fewer tiles leave CUs idle
// where the 128 x 128 kernel still fills them)
extern "C" int ddl_gemm256_ok(long long M, int N, int K) {
    static const int on = getenv("DDL_GEMM256") ? atoi(getenv("DDL_GEMM256")) : 0;                            // tuning knob
    static const int min_tiles = getenv("DDL_GEMM256_MIN_TILES") ? atoi(getenv("DDL_GEMM256_MIN_TILES")) : 160;  // tuning knob
    static const int min_k = getenv("DDL_GEMM256_MIN_K") ? atoi(getenv("DDL_GEMM256_MIN_K")) : 256;          // tuning knob
    if (!on || !ddl_gemm256_supported(M, N, K) || K < min_k) return 0;
    const long long tiles = ((M + 255) / 256) * (N / 256);
    return tiles >= min_tiles;
}

// y [M][N] = x [M][K] * w [N][K]^T (bf16, fp32 accumulation); `stats` ([stat_rows][2][N] fp32, optional): per 128
// output rows the sum and the sum of squares of the rounded outputs
extern "C" int ddl_gemm256_fwd(const void* x, const void* w, void* y, void* stats, long long M, int N, int K,
                               long long x_numel, int stat_rows, hipStream_t stream) {
    if (!x || !w || !y || !ddl_gemm256_supported(M, N, K)) return DDL_EINVAL;
    if (!ddl_aligned16(x) || !ddl_aligned16(w) || !ddl_aligned16(y) || (stats && !ddl_aligned16(stats))) return DDL_EALIGN;
    if (x_numel < M * K || M >= (1LL << 31) - 256) return DDL_EINVAL;
    if (stats && stat_rows < (int)((M + 127) / 128)) return DDL_EINVAL;
    Gemm256Params p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.stats = (float*)stats;
    p.x_bytes = x_numel * 2; p.w_bytes = (unsigned)((long long)N * K * 2);
    p.M = (int)M; p.N = N; p.K = K;
    p.tiles_m = (int)((M + 255) / 256); p.tiles_n = N / 256; p.stat_rows = stat_rows;
    static const int nst = getenv("DDL_GEMM256_NST") ? atoi(getenv("DDL_GEMM256_NST")) : 3;  // tuning knob
    dim3 grid(p.tiles_m * p.tiles_n), block(512);
    if (stats) {
        if (nst == 4) DDL_KLAUNCH((gemm256_fwd_kernel<true, 4>), grid, block, 0, stream, p);
        else DDL_KLAUNCH((gemm256_fwd_kernel<true, 3>), grid, block, 0, stream, p);
    } else {
        if (nst == 4) DDL_KLAUNCH((gemm256_fwd_kernel<false, 4>), grid, block, 0, stream, p);
        else DDL_KLAUNCH((gemm256_fwd_kernel<false, 3>), grid, block, 0, stream, p);
    }
    return ddl_launch_status();
}
