// K2w: input gradient AND raw weight-gradient product of a bottleneck's last 1x1 conv in ONE pass over the block's
// widest gradient (stage 1: 256 wide channels, 64 narrow ones, 56 x 56 pixels) - gfx950.
//
// With that BatchNorm's backward taken by linearity (hip_ops.py, "_bn3_linear"), two kernels read the masked
// output gradient g [M][256] (411 MB at batch 256): the two-segment GEMM that forms the input gradient
//     g2 = mask( g B1^T + z2 Q^T + bias )            [M][64]      (dependent chain)
// and the raw product of the weight gradient
//     P  = g^T z2                                    [256][64]    (weight-gradient stream).
// In the step they run next to each other, both bound by the same HBM (2.7 + 1.3 TB/s measured), and the second one
// only exists to read g and z2 again.  Here a workgroup owns a contiguous range of pixels and walks it in tiles of 64:
// the tile of g, z2 and y_prev goes global -> LDS once (LDS-DMA, double buffered, counted vmcnt), the input-gradient
// tile is formed from it against B1 / Q resident in LDS (MFMA, D rows = channels: 16-byte stores straight from the
// accumulators, masked by relu'(bn(y_prev)), with the sums of that BatchNorm's backward), and the SAME LDS image is
// then read through transposing loads (ds_read_tr16_b64) for g^T z2, accumulated in registers over the whole range
// (256 x 64 fp32 over 8 waves = 32 registers per lane).  A workgroup ends with one slab of P and one statistics row;
// slabs are summed by the ordinary split-K reduction in fixed order (deterministic, no atomics).
// Matrix work per tile is ~1150 cycles per SIMD against ~8000 cycles of HBM time: the kernel is a streaming kernel and
// is laid out for few, large memory instructions rather than for MFMA issue.
#include "common.h"
#include <cstdlib>

namespace {

struct LinFusedParams {
    const bf16_t* g;        // [M][256]
    const bf16_t* z2;       // [M][64]
    const bf16_t* b1;       // [64][256]  diag(ca) W^T rows
    const bf16_t* q;        // [64][64]
    const float* bias;      // [64]
    const bf16_t* y_prev;   // [M][64]   input of the BatchNorm + ReLU whose output gradient g2 is
    const float* fcoef;     // [2][64]   its forward scale | shift
    bf16_t* g2;             // [M][64]
    float* stats;           // [nwg][2][64]  sum g2 | sum g2 * y_prev
    float* partial;         // [nwg][256][64] fp32 slabs of g^T z2
    int M, tiles_per_wg;
};

typedef __attribute__((address_space(3))) bf16x4* lf_lds_bf16x4_ptr;

__device__ __forceinline__ void lf_store16(u32x4 rsrc, unsigned voff, unsigned soff, u32x4 v) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    // (wait states behind a store of more than 8 bytes are part of the asm: conv_rows.hip::gstore16)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff));
}

// MASK = false: plain store of g B1^T + z2 Q^T + bias (the projection shortcut's linear unit: the result is the addend
// of the block-input gradient), no y_prev, no statistics.
template <bool MASK>
__global__ void __launch_bounds__(512) lin_fused_kernel(const LinFusedParams p) {
    constexpr int C = 256, C2 = 64, BM = 64;
    constexpr int GROW = C * 2, ZROW = C2 * 2;                        // bytes per pixel row
    constexpr int WQ = C2 * GROW;                                     // b1 image [64][512 B], then q [64][128 B]
    constexpr int STG0 = WQ + C2 * ZROW;
    constexpr int GT = BM * GROW, ZT = BM * ZROW, STAGE = GT + 2 * ZT; // g tile | z2 tile | y_prev tile
    constexpr int FB = STG0 + 2 * STAGE;                              // bias[64] | scale[64] | shift[64] fp32
    constexpr unsigned INVALID = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FB + 768];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tile0 = blockIdx.x * p.tiles_per_wg;
    const int tiles_total = p.M / BM;
    int T = tiles_total - tile0;
    if (T > p.tiles_per_wg) T = p.tiles_per_wg;
    if (T < 0) T = 0;

    const unsigned g_bytes = (unsigned)((long long)p.M * GROW), z_bytes = (unsigned)((long long)p.M * ZROW);
    const u32x4 gr = raw_rsrc(p.g, g_bytes), zr = raw_rsrc(p.z2, z_bytes), yr = raw_rsrc(MASK ? p.y_prev : p.z2, z_bytes);
    const u32x4 or_ = raw_rsrc(p.g2, z_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- weights and coefficients into LDS, once.  Image row rho = 32 A + 16 I + Q holds output channel
    //      32 A + 8 (Q / 4) + 4 I + Q % 4: the two MFMA tiles (I = 0, 1) of a pair then give a lane 8 consecutive
    //      channels.  16-byte chunk c of row rho sits at chunk c ^ (rho & 31) (b1) / c ^ (rho & 7) (q).
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = tid + i * 512, rho = slot >> 5, ch = slot & 31;
        const int chan = (rho & 32) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.b1 + chan * C + ch * 8);
        *reinterpret_cast<u32x4*>(smem + rho * GROW + ((ch ^ (rho & 31)) << 4)) = v;
    }
    {
        const int rho = tid >> 3, ch = tid & 7;
        const int chan = (rho & 32) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.q + chan * C2 + ch * 8);
        *reinterpret_cast<u32x4*>(smem + WQ + rho * ZROW + ((ch ^ (rho & 7)) << 4)) = v;
    }
    if (tid < 64) reinterpret_cast<float*>(smem + FB)[tid] = p.bias[tid];
    else if (MASK && tid < 192) reinterpret_cast<float*>(smem + FB)[tid] = p.fcoef[tid - 64];

    // ---- DMA geometry (6 instructions per thread and tile; the swizzles are applied on the source side)
    //      g tile: rows of 512 B, 32-byte window w of row r at window w ^ (r & 7)   (conv_wgrad256's layout)
    //      z2 tile: rows of 128 B, window w of row r at window w ^ ((r >> 1) & 3)   (the row-streaming weight gradient's)
    //      y_prev tile: linear
    unsigned g_voff[4], z_voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = tid + i * 512, row = slot >> 5, chp = slot & 31;
        const int ch = (((chp >> 1) ^ (row & 7)) << 1) | (chp & 1);
        g_voff[i] = (unsigned)((row * C + ch * 8) * 2);
    }
    {
        const int m = tid >> 3, chp = tid & 7;
        const int ch = (((chp >> 1) ^ ((m >> 1) & 3)) << 1) | (chp & 1);
        z_voff = (unsigned)((m * C2 + ch * 8) * 2);
    }
    auto fetch = [&](int t, int stage) {
        const bool live = t < T;                                       // uniform
        const unsigned tile = (unsigned)(tile0 + (live ? t : 0));
        const unsigned St = lds0 + (unsigned)(STG0 + stage * STAGE + wid * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(gr, St + i * 8192, live ? g_voff[i] : INVALID, tile * (unsigned)GT);
        lds_dma16(zr, St + GT, live ? z_voff : INVALID, tile * (unsigned)ZT);
        lds_dma16(yr, St + GT + ZT, (live && MASK) ? (unsigned)(tid * 16) : INVALID, tile * (unsigned)ZT);   // (always issued: the counted wait)
    };

    // ---- input-gradient role: pixel fragment mf (pixels 16 mf + fr of the tile), channel pair a (channels 32 a ..)
    const int mf = wid & 3, pa = wid >> 2;
    const int mrow = 16 * mf + fr;
    int wa_off[2], wq_off[2];                 // weight fragments: row 32 a + 16 I + fr, chunk 4 ks + fq (xor applied per ks)
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int rho = 32 * pa + 16 * I + fr;
        wa_off[I] = rho * GROW;
        wq_off[I] = WQ + rho * ZROW;
    }
    const int rho_x[2] = {(32 * pa + fr) & 31, (32 * pa + 16 + fr) & 31};
    const int c8 = 32 * pa + 8 * fq;          // this lane's 8 output channels
    // ---- weight-gradient role: wide-channel fragments 2 wid, 2 wid + 1 against all four narrow fragments
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * tg + tq;
    int ga_off[2], zb_off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) ga_off[i] = tr_row * GROW + (((2 * wid + i) ^ (tr_row & 7)) << 5) + 8 * tp;
#pragma unroll
    for (int j = 0; j < 4; ++j) zb_off[j] = GT + tr_row * ZROW + ((j ^ ((tr_row >> 1) & 3)) << 5) + 8 * tp;

    f32x4 accw[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) accw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    fetch(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the weight image
    __builtin_amdgcn_s_barrier();
    float bia[8], fsc[8], fsh[8];
    load8f(reinterpret_cast<const float*>(smem + FB) + c8, bia);
    if (MASK) {
        load8f(reinterpret_cast<const float*>(smem + FB) + 64 + c8, fsc);
        load8f(reinterpret_cast<const float*>(smem + FB) + 128 + c8, fsh);
    }

    for (int t = 0; t < T; ++t) {
        const int stage = t & 1;
        fetch(t + 1, stage ^ 1);              // (its last readers passed the barrier that ended tile t-1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // tile t has landed everywhere
        const unsigned char* St = smem + STG0 + stage * STAGE;

        // ---- g2 tile: 8 reduction steps over g, 2 over z2 (the order of the two-segment tile kernel)
        f32x4 accd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int ch = 4 * ks + fq;
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(St + mrow * GROW + (((ch >> 1) ^ (mrow & 7)) << 5) + ((ch & 1) << 4));
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const bf16x8 wa = *reinterpret_cast<const bf16x8*>(smem + wa_off[I] + ((ch ^ rho_x[I]) << 4));
                accd[I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, accd[I], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 4 * ks + fq;
            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(St + GT + mrow * ZROW + (((ch >> 1) ^ ((mrow >> 1) & 3)) << 5) + ((ch & 1) << 4));
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const bf16x8 wa = *reinterpret_cast<const bf16x8*>(smem + wq_off[I] + ((ch ^ (rho_x[I] & 7)) << 4));
                accd[I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, accd[I], 0, 0, 0);
            }
        }
        {
            const bf16x8 yv = *reinterpret_cast<const bf16x8*>(St + GT + ZT + mrow * ZROW + c8 * 2);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (e < 4 ? accd[0][e & 3] : accd[1][e & 3]) + bia[e];
                if (MASK) {
                    const float yf = bf2f(yv[e]);
                    const bool on = bn_affine(yf, fsc[e], fsh[e]) > 0.f;
                    v[e] = on ? f2bf(a) : f2bf(0.f);
                    const float gg = bf2f(v[e]);
                    s1[e] += gg; s2[e] += gg * yf;
                } else {
                    v[e] = f2bf(a);
                }
            }
            lf_store16(or_, (unsigned)((mrow * C2 + c8) * 2), (unsigned)(tile0 + t) * (unsigned)ZT, __builtin_bit_cast(u32x4, v));
        }

        // ---- g^T z2 over the 64 pixels of the tile (two reduction steps of 32 rows)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* Sk = St + ks * 32 * GROW;
            const unsigned char* Zk = St + ks * 32 * ZROW;
            bf16x8 af[2], bf_[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Sk + ga_off[i]));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Sk + ga_off[i] + 16 * GROW));
                af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zb_off[j]));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zb_off[j] + 16 * ZROW));
                bf_[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf_[j], accw[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // everybody is done with this stage: the DMA of tile t+2 may overwrite it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the slab of this workgroup: rows = wide channel (4 per lane), columns = narrow channel
    float* out = p.partial + (long long)blockIdx.x * C * C2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[((2 * wid + i) * 16 + fq * 4 + e) * C2 + j * 16 + fr] = accw[i][j][e];

    if (!MASK) return;
    // ---- statistics row: fold the 16 pixel lanes, then the four pixel fragments (fixed order)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            s1[e] += __shfl_xor(s1[e], o, 64);
            s2[e] += __shfl_xor(s2[e], o, 64);
        }
    float* red = reinterpret_cast<float*>(smem + STG0);   // [2][4 fragments][64]  (all waves are past their last LDS read)
    if (fr == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[mf * 64 + c8 + e] = s1[e];
            red[256 + mf * 64 + c8 + e] = s2[e];
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        const float* r = red + which * 256 + ch;
        p.stats[((long long)blockIdx.x * 2 + which) * C2 + ch] = (r[0] + r[64]) + (r[128] + r[192]);
    }
}

}  // namespace

// Workgroups (= slabs of the raw product = statistics rows) the fused kernel uses for M pixels; 0 = not served
// (DDL_LIN_FUSED=0 switches it off; DDL_LIN_FUSED_WGS = workgroups to aim for, one per CU by default)
extern "C" int ddl_lin_fused_wgs(long long M, int C, int C2) {
    static const int on = getenv("DDL_LIN_FUSED") ? atoi(getenv("DDL_LIN_FUSED")) : 1;         // tuning knob
    static const int wgs = getenv("DDL_LIN_FUSED_WGS") ? atoi(getenv("DDL_LIN_FUSED_WGS")) : 256;  // tuning knob
    if (!on || C != 256 || C2 != 64 || M <= 0 || M % 64 != 0 || M * 512 >= 0x7fffffffLL) return 0;
    const long long tiles = M / 64;
    if (tiles < 512) return 0;                                  // small problems: the two plain kernels
    const long long per = (tiles + wgs - 1) / wgs;
    return (int)((tiles + per - 1) / per);
}

extern "C" int ddl_lin_fused(const void* g, const void* z2, const void* b1, const void* q, const void* bias,
                             const void* y_prev, const void* fcoef, void* g2, void* stats, void* partial,
                             long long M, int C, int C2, int nwg, hipStream_t stream) {
    if (!g || !z2 || !b1 || !q || !bias || !g2 || !partial) return DDL_EINVAL;
    if ((y_prev == nullptr) != (fcoef == nullptr) || (y_prev == nullptr) != (stats == nullptr)) return DDL_EINVAL;
    if (nwg <= 0 || nwg != ddl_lin_fused_wgs(M, C, C2)) return DDL_EINVAL;
    const void* al[] = {g, z2, b1, q, bias, y_prev, fcoef, g2, stats, partial};
    for (const void* a : al) if (a && !ddl_aligned16(a)) return DDL_EALIGN;
    LinFusedParams p;
    p.g = (const bf16_t*)g; p.z2 = (const bf16_t*)z2; p.b1 = (const bf16_t*)b1; p.q = (const bf16_t*)q;
    p.bias = (const float*)bias; p.y_prev = (const bf16_t*)y_prev; p.fcoef = (const float*)fcoef;
    p.g2 = (bf16_t*)g2; p.stats = (float*)stats; p.partial = (float*)partial;
    p.M = (int)M;
    const long long tiles = M / 64;
    p.tiles_per_wg = (int)((tiles + nwg - 1) / nwg);
    if (y_prev) DDL_KLAUNCH(lin_fused_kernel<true>, dim3(nwg), dim3(512), 0, stream, p);
    else DDL_KLAUNCH(lin_fused_kernel<false>, dim3(nwg), dim3(512), 0, stream, p);
    return ddl_launch_status();
}
