// K1r: 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels on 56-pixel rows (the
// middle conv of ResNet's first stage, forward and backward-data), as a ROW-STREAMING kernel (gfx950).
//
// In the tile kernel (conv_igemm.hip) this layer is nine 64-deep reduction steps per 128 x 64 tile: every
// step fetches a fresh copy of its input pixels (9x the tensor through L2 -> LDS) and one tap of the filter,
// does 16 MFMAs per wave and meets at a barrier; three workgroups per CU take turns waiting for DMA
// (114 us forward for 59 GFLOP and 206 MB; profiles/r3_conv_roofline_v2.txt).  Here
//   * a workgroup (8 waves) owns ONE IMAGE and walks down its rows, two output rows per step;
//   * the filter passes through LDS once per workgroup (72 KB by DMA, next to the first rows) and then lives in
//     registers: wave (nh, mg) keeps the [32 output channels][9 taps][64 channels] slice of its half nh in 144
//     registers for the whole image (MFMA A operand, D rows = output channels);
//   * input rows go global -> LDS exactly once, by LDS-DMA, into a ring of row slots (64 pixels x 128 bytes: pixel
//     0 and pixels 57.. are the zero halo, written by the DMA's range check; rows above / below the image
//     likewise), two steps ahead, waited for with counted `vmcnt` and one barrier per step;
//   * tap (r, s) of output pixel (row, w) is slot row + r, pixel w + s: the MFMA B operand is one 16-byte LDS
//     read per lane (16 pixels x 32 channels per fragment), each fragment feeds the two output rows that see its
//     input row and both 16-channel tiles: 72 MFMAs per 24 LDS reads per wave and step;
//   * D columns are pixels and D rows channels, so a lane holds 4 consecutive channels of one pixel: the epilogue
//     stores 16 bytes per lane (its two tiles' rows are dealt so that they are 8 consecutive channels) straight from
//     the (rounded) accumulators - no LDS staging of the output tile - and the epilogue of step t-1 sits between the
//     multiplies of step t.
// Every vector-memory instruction of the loop (DMA, output stores) is inline asm: hipcc's wait-count insertion
// never sees one and never drains the ring with `vmcnt(0)`; stores only make the counted waits stricter
// (loads return in order among themselves, stores do not return in order with loads).
//
// BWD = the backward-data form whose output is the gradient of relu(bn(y_prev)) (conv_igemm's EPI 2): the rows
// of y_prev ride in a second DMA ring, the epilogue masks with scale*y+shift > 0 and keeps sum g / sum g*y.
// Statistics come out as ONE row per image ([B][2][64]): the finalize kernels sum any number of rows.
#include "common.h"

namespace {

struct RowsParams {
    const bf16_t* x;        // [B][H][56][64]
    const bf16_t* w;        // [64][3][3][64]  (backward-data: the flipped / transposed filter, same layout)
    bf16_t* y;              // [B][H][56][64]
    float* stats;           // [B][2][64] or null
    const bf16_t* ep_y;     // BWD: [B][H][56][64]
    const float* ep_fcoef;  // BWD: [2][64] forward scale | shift
    const float* in_fcoef;  // XF: [2][64] scale | shift of the BatchNorm in FRONT of this conv: x is its raw input
    bf16_t* z_out;          // XF: [B][H][56][64] receives relu(scale * x + shift), the tensor the conv consumes
    const bf16_t* dy_y;     // DYF: [B][H][56][64] the raw output of the conv whose gradient x is to become
    const float* dy_coef;   // DYF: [3][64] ca | cb | cc of that BatchNorm's backward: the conv consumes ca*x + cb*dy_y + cc
    int B, H;
    unsigned bytes;         // of one activation tensor
};

__device__ __forceinline__ void gstore16(u32x4 rsrc, unsigned voff, unsigned soff, u32x4 v) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    // (no "memory" clobber: the kernel never reads its output back, and LDS reads may move across the store)
    // The wait states behind the store are part of the asm: a VMEM store of more than 8 bytes must not be followed at
    // once by a VALU write of its data registers (gfx9 store-data hazard).  hipcc pads its OWN stores but does not
    // look inside inline asm - without the s_nop the stored rows were occasionally wrong, and only in the training
    // step (the partner wave's issue timing decides what comes next), never in the isolated kernel.
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff));
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// XF (forward only): the input is the RAW output of the previous conv; BatchNorm apply + ReLU of the layer in between
// happen here, in LDS, on every row once after it has landed (each element is then read by nine taps) and the
// transformed rows are also written out (the weight gradient of this conv reads them): the separate apply pass - a
// read and a write of the whole tensor on the dependent chain - disappears.  One more ring stage: rows land two steps
// ahead, are transformed one step ahead and read from the next step on; halo pixels and rows stay zero.
//
// DYF (backward-data only): the input is the gradient g2 with respect to a BatchNorm's OUTPUT and `dy_y` that
// BatchNorm's input; the gradient the conv consumes, ca*g2 + cb*y2 + cc, is formed here in LDS (the backward apply
// pass - read two tensors, write one, on the dependent chain - disappears together with its output; the weight
// gradient of the same conv does the same on its own rows, conv_wgrad.hip).  The rows of y2 ride in a third ring next
// to their x rows (same pixel layout and swizzle: a thread's chunk has the same offset in both); a landed group is
// transformed in place behind the step's barrier and a second barrier follows (LDS has no room for the extra
// ring stage that lets the forward form hide its transform between the multiplies): x 64 + y2 32 + y_prev 48 KB.
template <bool BWD, bool XF = false, bool DYF = false>
__global__ void __launch_bounds__(512) conv_rows3x3_kernel(const RowsParams p) {
    static_assert(!(BWD && XF), "the input transform belongs to the forward form");
    static_assert(!DYF || BWD, "the gradient transform belongs to the backward-data form");
    constexpr int W = 56, C = 64;
    constexpr int LEAD = XF ? 3 : 2;                              // the DMA of group t + LEAD is issued in step t
    constexpr int NSX = 2 * LEAD + 4, NSY = DYF ? 3 : 4;          // x row slots; y_prev groups (two rows each)
    constexpr int YLEAD = DYF ? 1 : 2;                            // y_prev group t + YLEAD is issued in step t (read in step t + YLEAD + 1)
    constexpr int SLOT = 8192, XBYTES = NSX * SLOT + 256;         // + 256: the last tile's tap reads run 2 pixels past its slot
    constexpr int WBYTES = 64 * 9 * C * 2;                        // the whole filter, staged once
    constexpr int Y2BASE = XBYTES, Y2BYTES = DYF ? 4 * SLOT : 0;  // DYF: y2 ring, two groups
    constexpr int YBASE = Y2BASE + Y2BYTES;                       // y_prev ring (BWD)
    constexpr int RINGS = Y2BYTES + (BWD ? NSY * 2 * SLOT : 0);   // the filter's staging area before the loop lies over both
    constexpr int FBASE = Y2BASE + (RINGS > WBYTES ? RINGS : WBYTES), FBYTES = DYF ? 1280 : (BWD || XF) ? 512 : 0;
    constexpr int NPW = DYF ? 6 : BWD ? 4 : 2;                    // DMA instructions per wave and step
#ifndef DDL_CROWS_ISSUE_AT
#define DDL_CROWS_ISSUE_AT 2
#endif
#ifndef DDL_CROWS_PRIO
#define DDL_CROWS_PRIO 0
#endif
    // Wave priority experiments (profiles/r3_notes.md section 13): 1 = a wave runs the first half of its step at
    // raised priority (of the two waves of a SIMD the one that is BEHIND wins; by age waves 0-3 finish 800 cycles
    // early and wait at the barrier), 2 = waves 4-7 always high.  Cycles per step 3386 / 3338 / 3373 forward,
    // 3901 / 4184 / 3884 backward-data for 0 / 1 / 2: the SIMD's total is what counts, not who wins.
    constexpr int PRIO = DDL_CROWS_PRIO;
    constexpr int ISSUE_AT = DDL_CROWS_ISSUE_AT;                  // input row of the step behind whose reads the DMA is issued
    constexpr unsigned INVALID = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FBASE + FBYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wid >> 2, mg = wid & 3;
    const int col = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, H = p.H, T = H >> 1;
#ifdef DDL_CROWS_PROBE
    const unsigned long long cstart = __builtin_readcyclecounter();
#endif

    const u32x4 xr = raw_rsrc(p.x, p.bytes), yr = raw_rsrc(p.y, p.bytes);
    const u32x4 er = raw_rsrc(BWD ? p.ep_y : p.x, p.bytes), wr = raw_rsrc(p.w, WBYTES);
    const u32x4 zr = raw_rsrc(XF && p.z_out ? p.z_out : p.y, XF && p.z_out ? p.bytes : 0u);   // (no z_out: zero records)
    const u32x4 y2r = raw_rsrc(DYF ? p.dy_y : p.x, p.bytes);
    const bool zstore = XF && p.z_out != nullptr;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

    // ---- DMA geometry.  A wave moves pieces q0, q0+1 (8 pixels each) of row (wid >> 2) of the step's two rows.
    //      x: LDS pixel j = image column j - 1; 16-byte chunk c of pixel j sits at chunk c ^ ((j >> 1) & 7)
    //      (16 consecutive pixels of one fragment read: conflict-free), applied here on the source address.
    const int r2 = wid >> 2, q0 = 2 * (wid & 3);
    unsigned xv[2], yv_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 8 * (q0 + i) + (lane >> 3), cp = lane & 7;
        const int ch = cp ^ ((j >> 1) & 7), xcol = j - 1;
        xv[i] = (xcol >= 0 && xcol < W) ? (unsigned)(xcol * 128 + ch * 16) : INVALID;
        yv_off[i] = (q0 + i) < 7 ? (unsigned)((q0 + i) * 1024 + lane * 16) : INVALID;     // 56 pixels = 7 pieces, linear
    }
    // x rows v = 2k+2, 2k+3 (v = image row + 1; v = 0 and v = H+1 are zero rows)
    auto issue_x = [&](int k) {
        const int v = 2 * k + 2 + r2;
        const bool ok = v >= 1 && v <= H;                                   // uniform
        const unsigned soff = ok ? (unsigned)((b * H + v - 1) * W) * 128u : 0u;
        const unsigned dst = lds0 + (unsigned)(((v + NSX) % NSX) * SLOT + q0 * 1024);
        lds_dma16(xr, dst, ok ? xv[0] : INVALID, soff);
        lds_dma16(xr, dst + 1024, ok ? xv[1] : INVALID, soff);
    };
    // DYF: the y2 rows of the same group, laid out like their x rows; group k lives in ring half k & 1
    auto issue_y2 = [&](int k) {
        const int v = 2 * k + 2 + r2;
        const bool ok = v >= 1 && v <= H;
        const unsigned soff = ok ? (unsigned)((b * H + v - 1) * W) * 128u : 0u;
        const unsigned dst = lds0 + (unsigned)(Y2BASE + ((k & 1) * 2 + r2) * SLOT + q0 * 1024);
        lds_dma16(y2r, dst, ok ? xv[0] : INVALID, soff);
        lds_dma16(y2r, dst + 1024, ok ? xv[1] : INVALID, soff);
    };
    // BWD: the y_prev rows of step k (output rows 2k, 2k+1)
    auto issue_y = [&](int k) {
        const int row = 2 * k + r2;
        const bool yok = row >= 0 && row < H;
        const unsigned ysoff = yok ? (unsigned)((b * H + row) * W) * 128u : 0u;
        const unsigned yd = lds0 + (unsigned)(YBASE + ((k + NSY) % NSY) * 2 * SLOT + r2 * SLOT + q0 * 1024);
        lds_dma16(er, yd, yok ? yv_off[0] : INVALID, ysoff);
        lds_dma16(er, yd + 1024, yok ? yv_off[1] : INVALID, ysoff);
    };

    // ---- prologue: the filter (72 pieces of 1 KiB, 9 per wave) and the first six x rows are fetched together
#pragma unroll
    for (int i = 0; i < 9; ++i) lds_dma16(wr, lds0 + (unsigned)(Y2BASE + (wid + 8 * i) * 1024), (unsigned)((wid + 8 * i) * 1024 + lane * 16), 0u);
#pragma unroll
    for (int k = -1; k < LEAD; ++k) issue_x(k);
    if ((BWD || XF) && tid < 128) reinterpret_cast<float*>(smem + FBASE)[tid] = (XF ? p.in_fcoef : p.ep_fcoef)[tid];
    if (DYF && tid < 192) reinterpret_cast<float*>(smem + FBASE + 512)[tid] = p.dy_coef[tid];

    // ---- XF: a thread's chunk of a landed row: pixel tid / 8, physical chunk tid % 8 (1 KiB contiguous per wave)
    const int tj = tid >> 3, tlc = (tid & 7) ^ ((tj >> 1) & 7);             // logical chunk = channels 8 tlc ..
    const bool tvalid = tj >= 1 && tj <= W;
    auto transform = [&](int k, int r) {
        {
            const int v = 2 * k + 2 + r;
            if (v < 1 || v > H) return;                                     // uniform: zero rows stay zero
            if (tvalid) {
                unsigned char* a = smem + ((v + NSX) % NSX) * SLOT + tid * 16;
                bf16x8 d = *reinterpret_cast<const bf16x8*>(a);
                float sc[8], sh[8];
                load8f(reinterpret_cast<const float*>(smem + FBASE) + 8 * tlc, sc);
                load8f(reinterpret_cast<const float*>(smem + FBASE + 256) + 8 * tlc, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bn_affine(bf2f(d[e]), sc[e], sh[e]);     // (the forward apply kernel's rounding)
                    d[e] = f2bf(f > 0.f ? f : 0.f);
                }
                *reinterpret_cast<bf16x8*>(a) = d;
                if (zstore) gstore16(zr, (unsigned)(((tj - 1) * C + 8 * tlc) * 2), (unsigned)((b * H + v - 1) * W) * 128u, __builtin_bit_cast(u32x4, d));
            }
        }
    };

    // DYF: row r of landed group k becomes ca * g + cb * y2 + cc, in place (halo pixels and rows outside stay zero)
    auto transform_dy = [&](int k, int r) {
        const int v = 2 * k + 2 + r;
        if (v < 1 || v > H) return;                                         // uniform
        if (tvalid) {
            unsigned char* a = smem + ((v + NSX) % NSX) * SLOT + tid * 16;
            bf16x8 d = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 y2 = *reinterpret_cast<const bf16x8*>(smem + Y2BASE + ((k & 1) * 2 + r) * SLOT + tid * 16);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {        // (four channels at a time: the filter leaves few registers)
                const float* co = reinterpret_cast<const float*>(smem + FBASE + 512) + 8 * tlc + 4 * hh;
                const f32x4 ca = *reinterpret_cast<const f32x4*>(co), cb = *reinterpret_cast<const f32x4*>(co + 64),
                            cc = *reinterpret_cast<const f32x4*>(co + 128);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    d[4 * hh + e] = f2bf(bn_bwd_affine(bf2f(d[4 * hh + e]), bf2f(y2[4 * hh + e]), ca[e], cb[e], cc[e]));
            }
            *reinterpret_cast<bf16x8*>(a) = d;
        }
    };

    // ---- fragment read offsets inside a row slot: [s][channel half]
    int xo[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int j = 16 * mg + col + s;
#pragma unroll
        for (int h = 0; h < 2; ++h) xo[s][h] = j * 128 + (((4 * h + g) ^ ((j >> 1) & 7)) << 4);
    }
    // ---- epilogue geometry: this lane's pixel and its 4 channels per 16-channel tile
    const int px = 16 * mg + col;
    const bool pvalid = px < W;
    // Output channels are dealt to the two 16-row MFMA tiles of a wave so that a lane's 4 + 4 rows are 8 CONSECUTIVE
    // channels (tile j, row q  <->  channel 32*nh + 8*(q / 4) + 4*j + q % 4): one 16-byte store / LDS read per pixel
    const int ch8 = 32 * nh + 8 * g;
    const unsigned ov = pvalid ? (unsigned)((px * C + ch8) * 2) : INVALID;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- the filter slice of this wave: [tap][channel half][16-channel tile] fragments, A operand
    //      (lane: row = output channel col of the tile, 8 reduction channels 32h + 8g ..), kept for the whole image
    bf16x8 wreg[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wreg[tap][h][j] = *reinterpret_cast<const bf16x8*>(smem + Y2BASE + (((32 * nh + 8 * (col >> 2) + 4 * j + (col & 3)) * 9 + tap) * C + 32 * h + 8 * g) * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)   // (complete before the staging area is reused)
        asm volatile("" : "+v"(wreg[tap][0][0]), "+v"(wreg[tap][0][1]), "+v"(wreg[tap][1][0]), "+v"(wreg[tap][1][1]));
    if (BWD) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_y(0);
        if (DYF) issue_y(-1);       // (zero rows: the slot the idle epilogue of step 0 reads lies outside the staged filter)
        else issue_y(1);
    }
    if (DYF) {      // the rows of step 0 (groups -1 and 0); group t+1 follows behind the barrier of step t
        issue_y2(-1);
        issue_y2(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        transform_dy(-1, 0); transform_dy(-1, 1);
        transform_dy(0, 0); transform_dy(0, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_y2(1);
    }
    if (XF) {       // the rows of step 0 (groups -1 and 0); group t+1 follows inside step t
        transform(-1, 0); transform(-1, 1);
        transform(0, 0); transform(0, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    if (PRIO == 2 && nh == 1) __builtin_amdgcn_s_setprio(1);
    f32x4 acc[2][2];
    bf16x4 res[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[o][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            res[o][j] = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        }
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[j][e] = 0.f; s2[j][e] = 0.f; }

    // epilogue of output row 2*te + o (results in `res`): mask (BWD), statistics, one 16-byte store.  `live` = te >= 0.
    auto epilogue = [&](int te, int o, bool live) {
        const unsigned soff = live ? (unsigned)((b * H + 2 * te + o) * W) * 128u : 0u;
        bf16x8 v = __builtin_shufflevector(res[o][0], res[o][1], 0, 1, 2, 3, 4, 5, 6, 7);   // channels ch8 .. ch8 + 7
        if (BWD) {
            const bf16x8 yv = *reinterpret_cast<const bf16x8*>(smem + YBASE + ((te + NSY) % NSY) * 2 * SLOT + o * SLOT + px * 128 + ch8 * 2);
            float sc[8], sh[8];
            load8f(reinterpret_cast<const float*>(smem + FBASE) + ch8, sc);
            load8f(reinterpret_cast<const float*>(smem + FBASE + 256) + ch8, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // (`res` is zero for pad pixels and before the first step, and the y ring is finite there:
                // zero DMA fill, or the staged filter - no selects on the sums)
                const float yf = bf2f(yv[e]);
                const bool on = bn_affine(yf, sc[e], sh[e]) > 0.f;
                v[e] = on ? v[e] : f2bf(0.f);
                const float gg = bf2f(v[e]);
                s1[e >> 2][e & 3] += gg; s2[e >> 2][e & 3] += gg * yf;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(v[e]);
                s1[e >> 2][e & 3] += f; s2[e >> 2][e & 3] += f * f;
            }
        }
        gstore16(yr, live ? ov : INVALID, soff, __builtin_bit_cast(u32x4, v));
    };

#ifdef DDL_CROWS_PROBE
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, c0, c1;
#define CROWS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); c1 = __builtin_readcyclecounter(); ph[i] += c1 - c0; c0 = c1; __builtin_amdgcn_sched_barrier(0); } while (0)
    c0 = __builtin_readcyclecounter();
    ph[5] = c0 - cstart;                       // prologue: filter registers, first rows
#else
#define CROWS_STAMP(i) do { } while (0)
#endif
    // Step t: the multiplies of step t with, in between, the epilogue of step t-1 and the DMA of group t+2 (its
    // slots were last read in step t-1, before the barrier) - the matrix pipe works through the stores, the
    // statistics arithmetic and the DMA issue - then the counted wait for group t+1 and the barrier.
    for (int t = 0; t < T; ++t) {
        const int v0 = 2 * t;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (PRIO == 1 && i == 2) __builtin_amdgcn_s_setprio(0);
            const unsigned char* row = smem + ((v0 + i) % NSX) * SLOT;
            bf16x8 f[3][2];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) f[s][h] = *reinterpret_cast<const bf16x8*>(row + xo[s][h]);
            if (i == ISSUE_AT) {   // (behind this row's fragment reads: the DMA's "memory" clobber pins LDS reads)
                issue_x(t + LEAD);
                if (DYF) issue_y2(t + LEAD);
                if (BWD) issue_y(t + YLEAD);
            }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int r = i - o;
                if (r < 0 || r > 2) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[o][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[r * 3 + s][h][j], f[s][h], acc[o][j], 0, 0, 0);
            }
            if (i == 0) epilogue(t - 1, 0, t > 0);
            if (i == 1) epilogue(t - 1, 1, t > 0);
            // XF: the rows of group t+1 landed before the last barrier and are read from the next step on: their
            // transform sits between the multiplies like the epilogue
            if (XF && i == 1) transform(t + 1, 0);
            if (XF && i == 3) transform(t + 1, 1);
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) res[o][j][e] = pvalid ? f2bf(acc[o][j][e]) : f2bf(0.f);   // pad pixels (56..63) hold garbage
                acc[o][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        CROWS_STAMP(0);
        // ---- group t+1 has landed (everywhere, after the barrier) and everybody is done with rows 2t, 2t+1
        wait_vm<NPW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CROWS_STAMP(1);
        __builtin_amdgcn_s_barrier();
        if (DYF) {
            transform_dy(t + 1, 0); transform_dy(t + 1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        CROWS_STAMP(2);
    }
    epilogue(T - 1, 0, true);
    epilogue(T - 1, 1, true);
    wait_vm<0>();
    if (p.stats == nullptr) return;
#ifdef DDL_CROWS_PROBE
    // timing build: the table receives cycles per phase (multiply + epilogue, DMA wait, barrier, DMA issue) of
    // every wave instead of statistics
    __syncthreads();
    if (lane == 0) {
        ph[6] = __builtin_readcyclecounter() - cstart;   // whole kernel
        for (int i = 0; i < 7; ++i) p.stats[(long long)b * 128 + wid * 8 + i] = (float)ph[i];
    }
    return;
#endif

    // ---- statistics of the image: fold the 16 pixel lanes, then the four pixel groups (fixed order)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[j][e] += __shfl_xor(s1[j][e], o, 64);
                s2[j][e] += __shfl_xor(s2[j][e], o, 64);
            }
    __syncthreads();                              // all waves are past their last LDS read
    float* red = reinterpret_cast<float*>(smem);  // [2][4 pixel groups][64 channels]
    if (col == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = ch8 + 4 * j + e;
                red[mg * 64 + ch] = s1[j][e];
                red[256 + mg * 64 + ch] = s2[j][e];
            }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        const float* r = red + which * 256 + ch;
        p.stats[((long long)b * 2 + which) * 64 + ch] = (r[0] + r[64]) + (r[128] + r[192]);
    }
}

}  // namespace

// Shapes the row-streaming kernel serves (DDL_CONV_ROWS=0 switches it off: the tile kernel then takes them)
extern "C" int ddl_conv_rows3x3_ok(int C, int N, int R, int S, int stride, int pad, int H, int W) {
    static const int on = getenv("DDL_CONV_ROWS") ? atoi(getenv("DDL_CONV_ROWS")) : 1;  // tuning knob
    return on && C == 64 && N == 64 && R == 3 && S == 3 && stride == 1 && pad == 1 && W == 56 && H >= 2 && H % 2 == 0;
}

// y = conv3x3(x, w) for dense NHWC [B][H][56][64] tensors; `stats` ([B][2][64] fp32, may be null): per image
// sum y | sum y^2 (forward) or, with `ep_y` / `ep_fcoef`, sum g | sum g*ep_y of the masked gradient g that is stored
// `in_fcoef` (forward only): x is the raw input of a BatchNorm + ReLU in front of this conv ([2][64] scale | shift);
// the conv consumes relu(scale * x + shift) and, with `z_out`, also writes it out
// `dy_y` + `dy_coef` (backward-data only, i.e. with `ep_y`): x is the gradient with respect to a BatchNorm's output,
// dy_y that BatchNorm's input and dy_coef its backward coefficients [3][64] ca | cb | cc; the conv consumes
// ca * x + cb * dy_y + cc (rounded to bf16 like the separate apply pass)
extern "C" int ddl_conv_rows3x3(const void* x, const void* w, void* y, void* stats, const void* ep_y, const void* ep_fcoef,
                                const void* in_fcoef, void* z_out, const void* dy_y, const void* dy_coef,
                                int B, int H, int W, int C, int N, hipStream_t stream) {
    if (!x || !w || !y || B <= 0) return DDL_EINVAL;
    if ((z_out && !in_fcoef) || (in_fcoef && ep_y)) return DDL_EINVAL;
    if ((dy_y == nullptr) != (dy_coef == nullptr) || (dy_y && !ep_y)) return DDL_EINVAL;
    if (dy_y && (!ddl_aligned16(dy_y) || !ddl_aligned16(dy_coef))) return DDL_EALIGN;
    if (in_fcoef && (!ddl_aligned16(in_fcoef) || (z_out && !ddl_aligned16(z_out)))) return DDL_EALIGN;
    if (!ddl_conv_rows3x3_ok(C, N, 3, 3, 1, 1, H, W)) return DDL_EINVAL;
    if ((ep_y == nullptr) != (ep_fcoef == nullptr) || (ep_y && !stats)) return DDL_EINVAL;
    if (!ddl_aligned16(x) || !ddl_aligned16(w) || !ddl_aligned16(y) || (ep_y && !ddl_aligned16(ep_y)) ||
        (ep_fcoef && !ddl_aligned16(ep_fcoef)) || (stats && !ddl_aligned16(stats))) return DDL_EALIGN;
    const long long bytes = (long long)B * H * W * C * 2;
    if (bytes >= 0x7fffffffLL) return DDL_EINVAL;
    RowsParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.stats = (float*)stats;
    p.ep_y = (const bf16_t*)ep_y; p.ep_fcoef = (const float*)ep_fcoef;
    p.in_fcoef = (const float*)in_fcoef; p.z_out = (bf16_t*)z_out;
    p.dy_y = (const bf16_t*)dy_y; p.dy_coef = (const float*)dy_coef;
    p.B = B; p.H = H; p.bytes = (unsigned)bytes;
    if (in_fcoef) DDL_KLAUNCH((conv_rows3x3_kernel<false, true>), dim3(B), dim3(512), 0, stream, p);
    else if (dy_y) DDL_KLAUNCH((conv_rows3x3_kernel<true, false, true>), dim3(B), dim3(512), 0, stream, p);
    else if (ep_y) DDL_KLAUNCH((conv_rows3x3_kernel<true>), dim3(B), dim3(512), 0, stream, p);
    else DDL_KLAUNCH((conv_rows3x3_kernel<false>), dim3(B), dim3(512), 0, stream, p);
    return ddl_launch_status();
}
