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
//
// Stage 2 (512 wide / 128 narrow channels): B1 alone is 128 KB and the 512 x 128 fp32 accumulator of the raw product is
// 128 registers per lane - too much for one workgroup.  TWO workgroups share a pixel range there, each owning 64 of the
// 128 narrow channels (its rows of B1 / Q, its columns of the raw product, its half of g2 and of the statistics); both
// read the whole g tile, and they are placed on the SAME XCD (block b and b + 8) and walk the range in step, so the
// second read of a tile is served by that XCD's L2 and HBM still sees g once.  Tiles are 32 pixels there (LDS).
#include "common.h"
#include <cstdlib>

namespace {

struct LinFusedParams {
    const bf16_t* g;        // [M][C]
    const bf16_t* z2;       // [M][K2]
    const bf16_t* b1;       // [K2][C]   diag(ca) W^T rows
    const bf16_t* q;        // [K2][K2]
    const float* bias;      // [K2]
    const bf16_t* y_prev;   // [M][K2]   input of the BatchNorm + ReLU whose output gradient g2 is
    const float* fcoef;     // [2][K2]   its forward scale | shift
    bf16_t* g2;             // [M][K2]
    float* stats;           // [ranges][2][K2]  sum g2 | sum g2 * y_prev
    float* partial;         // [ranges][C][K2] fp32 slabs of g^T z2
    float* gram;            // GRAM: [ranges][K2][K2] fp32 slabs of z2^T z2
    float* colsum;          // GRAM: [ranges][2][K2] column sums of z2 (both rows alike: a statistics-table layout)
    int M, tiles_per_wg, ranges;
};

typedef __attribute__((address_space(3))) bf16x4* lf_lds_bf16x4_ptr;

__device__ __forceinline__ void lf_store16(u32x4 rsrc, unsigned voff, unsigned soff, u32x4 v) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    // (wait states behind a store of more than 8 bytes are part of the asm: conv_rows.hip::gstore16)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff));
}

// A workgroup produces 64 output channels (all of them for K2 = 64, half h for K2 = 128 with HALVES = 2).
// MASK = false: plain store of g B1^T + z2 Q^T + bias (the projection shortcut's linear unit: the result is the addend
// of the block-input gradient), no y_prev, no statistics.
// GRAM: the Gram matrix z2^T z2 and the column sums of z2 (the forward-data part of the linear weight gradient) are
// accumulated from the z2 tile as well: their own launch - another pass over z2 on the weight-gradient stream - disappears.
template <int C, int K2, int BM, int HALVES, bool MASK, bool GRAM>
__global__ void __launch_bounds__(512) lin_fused_kernel(const LinFusedParams p) {
    static_assert(K2 == 64 * HALVES && (BM == 32 || BM == 64) && C % 128 == 0, "64 output channels per workgroup");
    constexpr int GROW = C * 2, ZROW = K2 * 2, YROW = 128;            // bytes per pixel row (y_prev tile: this half only)
    constexpr int CPRG = C / 8, CPRZ = K2 / 8;                        // 16-byte chunks per row
    constexpr int STG0 = 64 * GROW;                                   // b1 image [64][GROW]
    constexpr int GT = BM * GROW, ZT = BM * ZROW, YT = BM * YROW, STAGE = GT + ZT + YT;
    constexpr int FB = STG0 + 2 * STAGE;                              // bias[64] | scale[64] | shift[64] fp32
    constexpr int KSG = C / 32, KSZ = K2 / 32, KSW = BM / 32;         // reduction steps: g, z2 (input gradient); pixels (raw product)
    constexpr int MF = BM / 16, ACT = 2 * MF;                         // pixel fragments; waves with an input-gradient role
    constexpr int CF = C / 128;                                       // wide-channel fragments of a wave in the raw product
    constexpr unsigned INVALID = 0x80000000u;
    static_assert(GT == 32768 && ZT == 8192, "4 + 1 DMA instructions per thread for the g and z2 tiles");
    static_assert(FB + 768 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[FB + 768];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int range = blockIdx.x, h = 0;
    if (HALVES == 2) {      // blocks b and b + 8 (same XCD) share a range
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        h = idx & 1;
        range = (idx >> 1) * 8 + xcd;
    }
    const int tile0 = range * p.tiles_per_wg;
    const int tiles_total = p.M / BM;
    int T = tiles_total - tile0;
    if (T > p.tiles_per_wg) T = p.tiles_per_wg;
    if (T < 0) T = 0;
    auto zswz = [](int row) { return ZROW == 128 ? ((row >> 1) & 3) : (row & 7); };
    auto chan_of = [](int rho) { return (rho & 32) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3); };

    const unsigned g_bytes = (unsigned)((long long)p.M * GROW), z_bytes = (unsigned)((long long)p.M * ZROW);
    const u32x4 gr = raw_rsrc(p.g, g_bytes), zr = raw_rsrc(p.z2, z_bytes), yr = raw_rsrc(MASK ? p.y_prev : p.z2, z_bytes);
    const u32x4 or_ = raw_rsrc(p.g2, z_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- B1 rows of this half and the coefficients into LDS, once.  Image row rho = 32 A + 16 I + Q holds output
    //      channel 32 A + 8 (Q / 4) + 4 I + Q % 4 (of the half): the two MFMA tiles (I = 0, 1) of a pair then give a lane
    //      8 consecutive channels.  16-byte chunk c of row rho sits at chunk c ^ (rho & 31).
#pragma unroll
    for (int i = 0; i < 64 * CPRG / 512; ++i) {
        const int slot = tid + i * 512, rho = slot / CPRG, ch = slot % CPRG;
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.b1 + (long long)(h * 64 + chan_of(rho)) * C + ch * 8);
        *reinterpret_cast<u32x4*>(smem + rho * GROW + ((ch ^ (rho & 31)) << 4)) = v;
    }
    if (tid < 64) reinterpret_cast<float*>(smem + FB)[tid] = p.bias[h * 64 + tid];
    else if (MASK && tid < 192) reinterpret_cast<float*>(smem + FB)[tid] = p.fcoef[((tid - 64) >> 6) * K2 + h * 64 + (tid & 63)];

    // ---- DMA geometry (6 instructions per thread and tile; the swizzles are applied on the source side)
    //      g tile: 32-byte window w of row r at window w ^ (r & 7)                       (conv_wgrad256's layout)
    //      z2 tile: window w of row r at w ^ ((r >> 1) & 3) (128-byte rows) / w ^ (r & 7) (256-byte rows)
    //      y_prev tile (this half's 64 channels): linear
    unsigned g_voff[4], z_voff, y_voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = tid + i * 512, row = slot / CPRG, chp = slot % CPRG;
        const int ch = (((chp >> 1) ^ (row & 7)) << 1) | (chp & 1);
        g_voff[i] = (unsigned)((row * C + ch * 8) * 2);
    }
    {
        const int m = tid / CPRZ, chp = tid % CPRZ;
        const int ch = (((chp >> 1) ^ zswz(m)) << 1) | (chp & 1);
        z_voff = (unsigned)((m * K2 + ch * 8) * 2);
        // (BM = 32: the tile has 256 slots; waves 4-7 repeat the transfers of waves 0-3 so that every wave issues the same count)
        const int ys = tid & (BM * 8 - 1);
        y_voff = (unsigned)(((ys >> 3) * K2 + h * 64 + (ys & 7) * 8) * 2);
    }
    const unsigned y_dst = (unsigned)((wid & (BM / 8 - 1)) * 1024);
    auto fetch = [&](int t, int stage) {
        const bool live = t < T;                                       // uniform
        const unsigned tile = (unsigned)(tile0 + (live ? t : 0));
        const unsigned St = lds0 + (unsigned)(STG0 + stage * STAGE);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(gr, St + i * 8192 + wid * 1024, live ? g_voff[i] : INVALID, tile * (unsigned)GT);
        lds_dma16(zr, St + GT + wid * 1024, live ? z_voff : INVALID, tile * (unsigned)ZT);
        lds_dma16(yr, St + GT + ZT + y_dst, (live && MASK) ? y_voff : INVALID, tile * (unsigned)ZT);   // (always issued: the counted wait)
    };

    // ---- input-gradient role (waves 0 .. ACT-1): pixel fragment mf, channel pair pa (channels 32 pa .. of the half)
    const bool active = wid < ACT;
    const int mf = wid & (MF - 1), pa = (wid / MF) & 1;
    const int mrow = 16 * mf + fr;
    int wa_off[2];                            // B1 fragments: row 32 pa + 16 I + fr, chunk 4 ks + fq (xor applied per step)
    const int rho_x[2] = {fr, 16 + fr};       // (row & 31)
    bf16x8 qa[2][KSZ];                        // Q fragments live in registers (rows of this half, all K2 columns)
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int rho = 32 * pa + 16 * I + fr;
        wa_off[I] = rho * GROW;
#pragma unroll
        for (int ks = 0; ks < KSZ; ++ks)
            qa[I][ks] = *reinterpret_cast<const bf16x8*>(p.q + (long long)(h * 64 + chan_of(rho)) * K2 + (4 * ks + fq) * 8);
    }
    // (pinned: under register pressure hipcc otherwise re-loads these from global memory INSIDE the tile loop, and its
    // `s_waitcnt vmcnt(0)` in front of their use drains the LDS-DMA ring every tile - 187 -> 762 us)
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int ks = 0; ks < KSZ; ++ks) asm volatile("" : "+v"(qa[I][ks]));
    const int c8 = 32 * pa + 8 * fq;          // this lane's 8 output channels (inside the half)
    // ---- raw-product role: wide-channel fragments CF wid .. against the four narrow fragments of the half
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * tg + tq;
    int ga_off[CF], zb_off[4];
#pragma unroll
    for (int i = 0; i < CF; ++i) ga_off[i] = tr_row * GROW + (((CF * wid + i) ^ (tr_row & 7)) << 5) + 8 * tp;
#pragma unroll
    for (int j = 0; j < 4; ++j) zb_off[j] = GT + tr_row * ZROW + (((4 * h + j) ^ zswz(tr_row)) << 5) + 8 * tp;

    f32x4 accw[CF][4];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) accw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // ---- Gram role: K2 / 16 row windows x the 4 column windows of this half = GW fragments per wave
    constexpr int NWIN = K2 / 16, GW = NWIN * 4 / 8;          // 2 (K2 = 64) or 4 (K2 = 128)
    const int g_wi = GW == 2 ? (wid >> 1) : wid;              // row window of this wave
    const int g_j0 = GW == 2 ? 2 * (wid & 1) : 0;             // its first column window (inside the half)
    const int zg_off = GT + tr_row * ZROW + ((g_wi ^ zswz(tr_row)) << 5) + 8 * tp;
    int zgb_off[GW];      // (its column fragments are read again rather than picked out of bf_[] by a run-time index:
                          //  hipcc turns that into a thousand v_cndmask per tile)
#pragma unroll
    for (int i = 0; i < GW; ++i) zgb_off[i] = GT + tr_row * ZROW + (((4 * h + g_j0 + i) ^ zswz(tr_row)) << 5) + 8 * tp;
    f32x4 accg[GRAM ? GW : 1], accs[GRAM ? 4 : 1];
    bf16x8 ones;
    if (GRAM) {
#pragma unroll
        for (int i = 0; i < GW; ++i) accg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) accs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = f2bf(1.f);
    }

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

        if (active) {
            // ---- g2 tile: the reduction steps over g, then those over z2 (the order of the two-segment tile kernel)
            f32x4 accd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KSG; ++ks) {
                const int ch = 4 * ks + fq;
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(St + mrow * GROW + (((ch >> 1) ^ (mrow & 7)) << 5) + ((ch & 1) << 4));
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    const bf16x8 wa = *reinterpret_cast<const bf16x8*>(smem + wa_off[I] + ((ch ^ rho_x[I]) << 4));
                    accd[I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, accd[I], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSZ; ++ks) {
                const int ch = 4 * ks + fq;
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(St + GT + mrow * ZROW + (((ch >> 1) ^ zswz(mrow)) << 5) + ((ch & 1) << 4));
#pragma unroll
                for (int I = 0; I < 2; ++I) accd[I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[I][ks], xb, accd[I], 0, 0, 0);
            }
            const bf16x8 yv = *reinterpret_cast<const bf16x8*>(St + GT + ZT + mrow * YROW + c8 * 2);
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
            lf_store16(or_, (unsigned)((mrow * K2 + h * 64 + c8) * 2), (unsigned)(tile0 + t) * (unsigned)ZT, __builtin_bit_cast(u32x4, v));
        }

        // ---- g^T z2 over the pixels of the tile (reduction steps of 32 rows)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            const unsigned char* Sk = St + ks * 32 * GROW;
            const unsigned char* Zk = St + ks * 32 * ZROW;
            bf16x8 af[CF], bf_[4];
#pragma unroll
            for (int i = 0; i < CF; ++i) {
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
            for (int i = 0; i < CF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf_[j], accw[i][j], 0, 0, 0);
            if (GRAM) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zg_off));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zg_off + 16 * ZROW));
                const bf16x8 zg = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < GW; ++i) {
                    bf16x4 l2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zgb_off[i]));
                    bf16x4 h2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lf_lds_bf16x4_ptr)(Zk + zgb_off[i] + 16 * ZROW));
                    const bf16x8 gb = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
                    accg[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zg, gb, accg[i], 0, 0, 0);
                }
                if (wid == 0) {      // a row of ones through the matrix pipe: every D row holds the column sums
#pragma unroll
                    for (int j = 0; j < 4; ++j) accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf_[j], accs[j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // everybody is done with this stage: the DMA of tile t+2 may overwrite it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the slab of this range: rows = wide channel (4 per lane), columns = narrow channel (this half's)
    float* out = p.partial + (long long)range * C * K2;
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[((CF * wid + i) * 16 + fq * 4 + e) * K2 + h * 64 + j * 16 + fr] = accw[i][j][e];

    if (GRAM) {
        float* go = p.gram + (long long)range * K2 * K2;
#pragma unroll
        for (int i = 0; i < GW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                go[(g_wi * 16 + fq * 4 + e) * K2 + h * 64 + (g_j0 + i) * 16 + fr] = accg[i][e];
        if (wid == 0 && fq == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p.colsum[((long long)range * 2) * K2 + h * 64 + j * 16 + fr] = accs[j][0];
                p.colsum[((long long)range * 2 + 1) * K2 + h * 64 + j * 16 + fr] = accs[j][0];
            }
        }
    }
    if (!MASK) return;
    // ---- statistics row: fold the 16 pixel lanes, then the pixel fragments (fixed order)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            s1[e] += __shfl_xor(s1[e], o, 64);
            s2[e] += __shfl_xor(s2[e], o, 64);
        }
    float* red = reinterpret_cast<float*>(smem + STG0);   // [2][MF][64]  (all waves are past their last LDS read)
    if (active && fr == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[mf * 64 + c8 + e] = s1[e];
            red[MF * 64 + mf * 64 + c8 + e] = s2[e];
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        const float* r = red + which * MF * 64 + ch;
        float t = r[0] + r[64];
        if (MF == 4) t = t + (r[128] + r[192]);
        p.stats[((long long)range * 2 + which) * K2 + h * 64 + ch] = t;
    }
}

struct LinFusedShape { int BM, halves; };
static bool lin_fused_shape(int C, int C2, LinFusedShape& s) {
    if (C == 256 && C2 == 64) { s = {64, 1}; return true; }
    if (C == 512 && C2 == 128) { s = {32, 2}; return true; }
    return false;
}

}  // namespace

// Pixel ranges (= slabs of the raw product = statistics rows) the fused kernel uses for M pixels; 0 = not served
// (DDL_LIN_FUSED=0 switches it off, 1 = stage 1 only; DDL_LIN_FUSED_WGS / _WGS2 = ranges to aim for in the one- / two-workgroup form)
extern "C" int ddl_lin_fused_wgs(long long M, int C, int C2) {
    static const int on = getenv("DDL_LIN_FUSED") ? atoi(getenv("DDL_LIN_FUSED")) : 2;         // tuning knob
    // 128 ranges: half the CUs for this launch, the other half for the weight-gradient stream's kernels, which aim for 128
    // workgroups too (16.01 / 16.02 / 16.03 ms with 256 -> 15.94 / 15.93 / 15.95; 96 / 160 / 192: 16.13 / 16.15 / 16.03 -
    // profiles/r3_notes.md section 32); the two-workgroup form of stage 2 then runs 256 workgroups, one round
    static const int wgs1 = getenv("DDL_LIN_FUSED_WGS") ? atoi(getenv("DDL_LIN_FUSED_WGS")) : 128;  // tuning knob
    static const int wgs2 = getenv("DDL_LIN_FUSED_WGS2") ? atoi(getenv("DDL_LIN_FUSED_WGS2")) : wgs1;  // tuning knob (ranges of the two-workgroup form)
    LinFusedShape sh;
    if (!on || !lin_fused_shape(C, C2, sh) || (sh.halves == 2 && on < 2)) return 0;
    if (M <= 0 || M % sh.BM != 0 || M * C * 2 >= 0x7fffffffLL) return 0;
    const long long tiles = M / sh.BM;
    if (tiles < 512) return 0;                                  // small problems: the two plain kernels
    const int wgs = sh.halves == 2 ? wgs2 : wgs1;
    const long long per = (tiles + wgs - 1) / wgs;
    long long ranges = (tiles + per - 1) / per;
    if (sh.halves == 2) ranges = (ranges + 7) / 8 * 8;          // (pairs of workgroups are dealt to the XCDs in groups of 8)
    return (int)ranges;
}

// 1 when the fused launch of this shape can also accumulate the Gram matrix and the column sums of z2 (stage 1: the
// two-workgroup form of stage 2 has no registers left for it - 16 spilled with them)
extern "C" int ddl_lin_fused_gram_ok(int C, int C2) {
    LinFusedShape sh;
    return lin_fused_shape(C, C2, sh) && sh.halves == 1;
}

extern "C" int ddl_lin_fused(const void* g, const void* z2, const void* b1, const void* q, const void* bias,
                             const void* y_prev, const void* fcoef, void* g2, void* stats, void* partial,
                             void* gram, void* colsum, long long M, int C, int C2, int nwg, hipStream_t stream) {
    if (!g || !z2 || !b1 || !q || !bias || !g2 || !partial) return DDL_EINVAL;
    if ((gram == nullptr) != (colsum == nullptr) || (gram && (!ddl_aligned16(gram) || !ddl_aligned16(colsum)))) return DDL_EINVAL;
    if (gram && !ddl_lin_fused_gram_ok(C, C2)) return DDL_EINVAL;
    if ((y_prev == nullptr) != (fcoef == nullptr) || (y_prev == nullptr) != (stats == nullptr)) return DDL_EINVAL;
    if (nwg <= 0 || nwg != ddl_lin_fused_wgs(M, C, C2)) return DDL_EINVAL;
    const void* al[] = {g, z2, b1, q, bias, y_prev, fcoef, g2, stats, partial};
    for (const void* a : al) if (a && !ddl_aligned16(a)) return DDL_EALIGN;
    LinFusedShape sh;
    if (!lin_fused_shape(C, C2, sh)) return DDL_EINVAL;
    LinFusedParams p;
    p.g = (const bf16_t*)g; p.z2 = (const bf16_t*)z2; p.b1 = (const bf16_t*)b1; p.q = (const bf16_t*)q;
    p.bias = (const float*)bias; p.y_prev = (const bf16_t*)y_prev; p.fcoef = (const float*)fcoef;
    p.g2 = (bf16_t*)g2; p.stats = (float*)stats; p.partial = (float*)partial;
    p.gram = (float*)gram; p.colsum = (float*)colsum;
    p.M = (int)M; p.ranges = nwg;
    const long long tiles = M / sh.BM;
    p.tiles_per_wg = (int)((tiles + nwg - 1) / nwg);
    const dim3 grid(nwg * sh.halves), block(512);
#define LF(CC, KK, BB, HH) do { \
        if (y_prev) { if (gram) DDL_KLAUNCH((lin_fused_kernel<CC, KK, BB, HH, true, true>), grid, block, 0, stream, p); \
                      else DDL_KLAUNCH((lin_fused_kernel<CC, KK, BB, HH, true, false>), grid, block, 0, stream, p); } \
        else { if (gram) DDL_KLAUNCH((lin_fused_kernel<CC, KK, BB, HH, false, true>), grid, block, 0, stream, p); \
               else DDL_KLAUNCH((lin_fused_kernel<CC, KK, BB, HH, false, false>), grid, block, 0, stream, p); } } while (0)
    if (sh.halves == 1) LF(256, 64, 64, 1);
    else if (y_prev) DDL_KLAUNCH((lin_fused_kernel<512, 128, 32, 2, true, false>), grid, block, 0, stream, p);
    else DDL_KLAUNCH((lin_fused_kernel<512, 128, 32, 2, false, false>), grid, block, 0, stream, p);
#undef LF
    return ddl_launch_status();
}
