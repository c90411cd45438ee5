// K1/K2/K3/K5: implicit-GEMM convolution forward and backward-data on bf16 MFMA (gfx950).
//
//   y[m][n] = sum_{r,s,c} x[b, ho*ostride - pad + r, wo*ostride - pad + s, c] * w[n][r][s][c]
//
// with m = (b, ho, wo) linearised, x NHWC bf16, w "KRSC" bf16 (n-major, k contiguous), y NHWC
// bf16, fp32 accumulation.  The same kernel serves
//   * forward conv (any RxS, stride, pad; Cin % 32 == 0),
//   * 1x1 convs and the dense head (R = S = H = W = 1)  -> plain GEMM,
//   * the stem 7x7/2 conv through the "row-of-taps" view (R = 7, S = 1, C = 32 over a W-padded
//     NHWC4 image: one tap row = 8 taps x 4 channels = 64 contiguous bytes),
//   * backward-data: dx = conv(dy dilated by the forward stride, flipped/transposed weights
//     "CRSK") with ostride = 1 and input dilation idil = forward stride.
//
// Structure (CDNA4): 256 threads = 4 waves (2 x 2), block tile BM x BN x BK, wave tile
// BM/2 x BN/2 of v_mfma_f32_16x16x32_bf16 fragments.  Operand tiles go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no register stop, out-of-image taps and M/N edges arrive as
// zeros through the buffer range check, no branches; the XOR swizzle of the 16-byte chunks is
// applied to the per-lane SOURCE address because the destination is lane-linear), two LDS
// buffers, one barrier per K tile, conflict-free ds_read_b128 fragment reads.  A register-staged
// twin of the loop (global -> VGPR -> ds_write_b128) is kept behind DDL_CONV_GLDS=0.  The product
// is computed swapped (D rows = output channels, D cols = pixels) so each lane owns 4 consecutive
// channels; the epilogue stages the bf16 tile through LDS and stores whole rows with 16-byte
// lanes, optionally adding a residual tensor (backward shortcut fusion) and emitting per-tile
// per-channel sum / sum-of-squares partials for BatchNorm (K6 fused: saves one full read of y).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct IGemmParams {
    const bf16_t* x;
    const bf16_t* w;
    bf16_t* y;
    const bf16_t* addend;  // optional [M][N], added in the epilogue
    float* stats;          // optional [tilesM][2][N] partial sums of y and y^2 (bf16-rounded y)
    long long x_bytes;     // whole input tensor; every workgroup addresses it relative to its first image (32-bit offsets)
    unsigned w_bytes;
    // DUAL (plain-GEMM launches only): a second reduction segment appended to the first,
    //   y[m][n] = sum_c x[m][c] w[n][c] + sum_j x2[m][j] w2[n][j] + bias[n]
    // (BatchNorm backward by linearity: the gradient of a 1x1 conv's input from g and z instead of dy)
    const bf16_t* x2;
    const bf16_t* w2;
    const float* bias;     // optional [N] fp32, added before the bf16 rounding
    long long x2_bytes;
    unsigned w2_bytes;
    int C2;
    int wt_rows;           // WT: reduction rows (forward output channels) the weight tensor really has (<= C)
    int wt;                // backward-data reads the FORWARD weights w[k][r][s][n] ("KRSC", n contiguous) through transposing LDS reads
    int Hin, Win, C;       // input image (logical, before dilation) and reduction channels
    long long xs_b, xs_h, xs_w;  // element strides of the input
    int Hout, Wout, N, M;
    int R, S, ostride, pad, pad_w, idil_log2;
    int tiles_m, tiles_n;
    unsigned long long* dbg;  // probe builds: [block][8] cycle stamps of wave 0
    int probe;             // timing-only probes (DDL_CONV_PROBE): 4 no MFMA, 8 no LDS stores, 16 no LDS fragment reads, 32 no barrier
    int stat_rows;         // rows of `stats`: one per 128 output rows (a 256-row tile writes two)
    int simple;       // 1x1, stride 1, unpadded, dense NHWC input: row m of the GEMM is pixel m (no (b, h, w) split)
    int addend_sub2;  // addend is [B][Hout/2][Wout/2][N]: added at even (h, w) pixels only
    int parity, tiles_per_class, Mc;  // stride-2 backward-data: M ordered by output-pixel parity class
    // fused backward epilogues (EPI 2 / 3), all tensors [M][N] like y:
    const bf16_t* ep_y;      // EPI 2: pre-BatchNorm activation of the layer whose output gradient this is
    const float* ep_fcoef;   // EPI 2: [2][N] forward scale | shift of that BatchNorm (ReLU mask = scale*y+shift > 0)
    const bf16_t* ep_mask;   // EPI 3: ReLU output; the result is zeroed where it is <= 0
    // EPI 4, optional: a second pre-BatchNorm tensor fed by the same gradient (the projection
    // shortcut's conv output); `stats2` receives [stat_rows][2][N] = sum g | sum g*ep_y2
    const bf16_t* ep_y2;
    float* stats2;
    // EPI 3 / 4: the ReLU mask as one bit per element (byte i = elements 8i .. 8i+7 of the [M][N]
    // tensor, written by the forward BatchNorm apply) instead of the activation itself
    const unsigned char* ep_maskbits;
};

// byte offset of 16-byte chunk `ch` of row `row` in a [rows][BK] bf16 LDS tile (swizzled)
template <int BK>
__device__ __forceinline__ int lds_off(int row, int ch) {
    if (BK == 64) return row * 128 + ((ch ^ (row & 7)) << 4);
    return row * 64 + ((ch ^ ((0 - (row >> 2)) & 3)) << 4);
}

// EPI: 0 plain store; 1 forward BatchNorm statistics (sum y, sum y^2 per tile and channel);
//      2 backward-data feeding a BatchNorm+ReLU backward: g = dz * relu'(bn(y)) is stored and the
//        tile's sum g / sum g*y partials are emitted (replaces the separate K8a reduce pass);
//      3 backward-data whose consumer is a ReLU: store dz * (mask > 0) so that the consumer's
//        backward needs neither the mask tensor nor a masked copy of the gradient;
//      4 = 3 plus the sum g / sum g*y partials of the BatchNorm that feeds that ReLU (ep_y), and
//        with ep_y2 also those of the projection shortcut's BatchNorm (same g, other y).
#ifndef DDL_CONV_STAGES
#define DDL_CONV_STAGES 2
#endif
#ifndef DDL_CONV_MIN_WAVES
#define DDL_CONV_MIN_WAVES 2   // waves per SIMD the register allocation must allow (3 and 4 measured: profiles/r1_notes.md)
#endif
// swizzle of the transposed weight image (WT): 32-byte window w of reduction row r is stored at
// window w ^ wt_swz(r), so that the 8 rows a half-wave's transposing read touches ({8g+q}, g in
// {0,1} or {2,3}, q = 0..3, and the same +4) fall into 8 distinct 32-byte windows of a bank row
template <int ROW_BYTES>
__device__ __forceinline__ int wt_swz(int row) {
    return ROW_BYTES == 256 ? ((row & 3) | ((row >> 1) & 4)) : (((row >> 1) & 1) | ((row >> 2) & 2));
}
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// LEAN (EPI 4 only): the ReLU mask arrives as bits and there is no second BatchNorm table - the common
// launch of the training step.  As compile-time facts they free the registers of the two alternatives
// (a bf16 mask tile, the second pre-BatchNorm tile, a third sum): 152 -> 128 VGPRs or fewer, i.e. 4
// workgroups per CU instead of 3 for a kernel that is sensitive to exactly that (profiles/r2_notes.md).
template <int BM, int BN, int BK, int EPI, bool ADDEND, bool GLDS, bool WT = false, bool DUAL = false, bool LEAN = false>
__global__ void __launch_bounds__(256, (BM == 128 && BK == 32) ? DDL_CONV_MIN_WAVES : 2) conv_igemm_kernel(const IGemmParams p) {
    static_assert(!DUAL || (GLDS && !WT && !ADDEND), "the two-segment reduction exists for LDS-DMA plain GEMMs");
    static_assert(BM % 128 == 0, "statistics rows are per 128 output rows");
    static_assert(!WT || (GLDS && EPI != 1), "the transposed weight image is an LDS-DMA image; forward statistics never use it");
    constexpr bool STATS = EPI == 1 || EPI == 2 || EPI == 4;
    constexpr int CPR = BK / 8;            // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;         // rows covered per pass of the 256 threads
    constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int OSTR = BN * 2 + 16;        // padded out-tile row stride (bytes)
    constexpr int SROWS = EPI == 4 ? 3 : 2;   // sums per channel kept by the epilogue
    constexpr int EPI_BYTES = BM * OSTR + 4 * SROWS * BN * 4;  // out tile + [4 waves][SROWS][BN] fp32 stats scratch
    // LDS-DMA ring: NSTG stages, NSTG-1 tiles in flight across the barrier (counted vmcnt, raw s_barrier);
    // 2 = one tile in flight, drained at every barrier (the round-1/2 loop)
    constexpr int NSTG = (GLDS && BK == 32) ? DDL_CONV_STAGES : 2;
    constexpr int LDS_BYTES = (NSTG * STAGE > EPI_BYTES) ? NSTG * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
#ifdef DDL_CONV_PROBES
    unsigned long long stamp[12];
    for (int i = 0; i < 12; ++i) stamp[i] = 0;
#define DDL_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
    DDL_STAMP(0);
#else
#define DDL_STAMP(i) do { } while (0)
#endif

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
    // contiguous run of logical tiles, n fastest, so the A panel of an M tile is reused in L2.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int n0 = tile_n * BN;
    // Parity mode (backward-data of a stride-2 conv): the input gradient is dy dilated by 2, so
    // an output pixel (hi, wi) only sees taps with r = (pad + hi) mod 2, s = (pad_w + wi) mod 2.
    // Rows are grouped by parity class (hi & 1, wi & 1): a tile has ONE class, walks only its
    // live taps (9 instead of 36 tap visits over the four classes of a 3x3) and skips the rest.
    int m0 = tile_m * BM, pcls = 0, ph = 0, pw = 0, m_lim = p.M;
    if (p.parity) {
        // classes interleaved over the tile index: a 3x3 tile visits 1, 2, 2 or 4 taps depending on
        // its class, and class-major order handed the XCDs that got the last class 4x the work of
        // the first (stride-2 backward-data ran at 0.65x the forward rate)
        pcls = tile_m & 3;
        m0 = (tile_m >> 2) * BM;
        ph = pcls >> 1; pw = pcls & 1;
        m_lim = p.Mc;
    }
    const int tstep = p.parity ? 2 : 1;
    const int r0 = p.parity ? ((p.pad + ph) & 1) : 0, s0 = p.parity ? ((p.pad_w + pw) & 1) : 0;

    // Tensors beyond 2 GiB: the buffer descriptor starts at the first image this tile touches
    // (plain GEMM: at the tile's first row), so the 32-bit offsets only span one tile's images.
    const int Wq = p.parity ? (p.Wout >> 1) : p.Wout;
    const int hw = p.parity ? (p.Hout >> 1) * Wq : p.Hout * p.Wout;
    const int b_lo = p.simple ? 0 : m0 / hw;
    const long long x_first = p.simple ? (long long)m0 * p.xs_w : (long long)b_lo * p.xs_b;   // elements
    long long x_left = p.x_bytes - x_first * 2;
    if (x_left < 0) x_left = 0;
    if (x_left > 0x7fffffffLL) x_left = 0x7fffffffLL;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + x_first), 0, (int)x_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    long long x2_left = DUAL ? p.x2_bytes - (long long)m0 * p.C2 * 2 : 0;
    x2_left = x2_left < 0 ? 0 : (x2_left > 0x7fffffffLL ? 0x7fffffffLL : x2_left);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.x2 + (long long)m0 * p.C2 : p.x), 0, (int)x2_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.w2 : p.w), 0, DUAL ? (int)p.w2_bytes : 0, 0x00020000);
    // raw twins for the LDS-DMA issued from inline asm (common.h::lds_dma16)
    const u32x4 xq = raw_rsrc(p.x + x_first, (unsigned)x_left), wq = raw_rsrc(p.w, p.w_bytes);
    const u32x4 xq2 = raw_rsrc(DUAL ? p.x2 + (long long)m0 * p.C2 : p.x, (unsigned)x2_left), wq2 = raw_rsrc(DUAL ? p.w2 : p.w, DUAL ? p.w2_bytes : 0u);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);

    // GLDS (operand tiles by LDS-DMA, `buffer_load_dwordx4 ... lds`): a wave-instruction writes
    // 64 x 16 bytes at a wave-uniform LDS base + lane * 16, so the swizzle moves to the SOURCE side:
    // the lane at physical chunk position `chp` of its row fetches logical chunk chp ^ f(row)
    // (f only depends on row bits that the per-slot row stride RPP leaves alone).
    const int chp = tid % CPR, row0 = tid / CPR;
    const int ch = !GLDS ? chp : (BK == 64 ? (chp ^ (row0 & 7)) : (chp ^ ((0 - (row0 >> 2)) & 3)));
    const int Kdim = p.R * p.S * p.C;
    // per-row A state
    int a_base[A_LOADS];   // element offset from x_first
    int a_h0[A_LOADS], a_w0[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        int m = m0 + row0 + i * RPP;
        if (p.simple) {  // plain GEMM: 36 of ResNet-50's 53 convs; saves the two integer divisions per row
            a_h0[i] = m < m_lim ? 0 : -(1 << 28);
            a_w0[i] = 0;
            a_base[i] = (row0 + i * RPP) * (int)p.xs_w + ch * 8;
        } else if (m < m_lim) {
            int b = m / hw, rem = m - b * hw;
            int ho = rem / Wq, wo = rem - ho * Wq;
            if (p.parity) { ho = 2 * ho + ph; wo = 2 * wo + pw; }
            a_h0[i] = ho * p.ostride - p.pad;
            a_w0[i] = wo * p.ostride - p.pad_w;
            a_base[i] = (b - b_lo) * (int)p.xs_b + ch * 8;
        } else {
            a_h0[i] = -(1 << 28); a_w0[i] = 0; a_base[i] = 0;  // never valid
        }
    }
    unsigned b_off[B_LOADS];
    unsigned b_off_last[WT ? B_LOADS : 1];   // WT: the same for the last reduction tile, rows past the weight's end masked
    constexpr int WROW = BN * 2;        // WT: bytes per reduction row of the [BK][BN] weight image
    constexpr int WCPR = BN / 8;        //     16-byte chunks per row
    const int RSN = p.R * p.S * p.N;    // WT: elements between consecutive reduction channels of w[k][r][s][n]
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        if (WT) {
            // slot idx of the dense image: row = reduction channel inside the tile, physical chunk
            // position -> logical chunk (the windows are swizzled on the SOURCE side, like the A tile)
            const int idx = tid + i * 256, row = idx / WCPR, chq = idx - row * WCPR;
            const int lch = (((chq >> 1) ^ wt_swz<WROW>(row)) << 1) | (chq & 1);
            const int n = n0 + lch * 8;
            b_off[i] = n < p.N ? (unsigned)((row * RSN + n) * 2) : 0x80000000u;
            // (the scalar offset that selects the reduction tile is not range-checked by the hardware)
            b_off_last[i] = (p.C - BK + row < p.wt_rows) ? b_off[i] : 0x80000000u;
        } else {
            int n = n0 + row0 + i * RPP;
            long long o = ((long long)n * Kdim + ch * 8) * 2;
            b_off[i] = (n < p.N && o < 0x7fffffffLL) ? (unsigned)o : 0x80000000u;
        }
    }

    const int idil_mask = (1 << p.idil_log2) - 1;
    const int cpt = p.C / BK;          // K tiles per filter tap
    const int nr = r0 < p.R ? (p.R - r0 + tstep - 1) / tstep : 0;
    const int ns = s0 < p.S ? (p.S - s0 + tstep - 1) / tstep : 0;
    const int KT1 = nr * ns * cpt;
    const int KT = KT1 + (DUAL ? p.C2 / BK : 0);
    unsigned a_voff2[DUAL ? A_LOADS : 1], b_off2[DUAL ? B_LOADS : 1];
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            a_voff2[DUAL ? i : 0] = (m0 + row0 + i * RPP < m_lim) ? (unsigned)(((row0 + i * RPP) * p.C2 + ch * 8) * 2) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int n = n0 + row0 + i * RPP;
            b_off2[DUAL ? i : 0] = n < p.N ? (unsigned)((n * p.C2 + ch * 8) * 2) : 0x80000000u;
        }
    }
    int kfetch = 0;   // DUAL: index of the K tile being fetched
    int kr = r0, ks = s0, kc = 0;      // tap / channel-block counters of the tile being fetched

    // Operand staging: the next tile's loads are issued before the current tile is consumed.
    // Tap geometry (bounds, dilation, pixel offset) is recomputed only when the filter tap
    // changes; the channel-block step rides in the buffer instruction's scalar offset.
    constexpr unsigned INVALID = 0x80000000u;
    const int xs_h2 = (int)(p.xs_h * 2), xs_w2 = (int)(p.xs_w * 2);
    unsigned a_voff[A_LOADS];
    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int th = a_h0[i] + kr, tw = a_w0[i] + ks;
            bool ok = (th >= 0) & (tw >= 0) & (((th | tw) & idil_mask) == 0);
            int hi = th >> p.idil_log2, wi = tw >> p.idil_log2;
            ok = ok & (hi < p.Hin) & (wi < p.Win);
            a_voff[i] = ok ? (unsigned)(a_base[i] * 2 + hi * xs_h2 + wi * xs_w2) : INVALID;
        }
    };
    auto fetch = [&](u32x4 (&ar)[A_LOADS], u32x4 (&br)[B_LOADS]) {
        if (kc == 0) retap();
        const int kco = kc * (BK * 2);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            ar[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)a_voff[i], kco, 0);
        const int wk = ((kr * p.S + ks) * p.C + kc * BK) * 2;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            br[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)b_off[i], wk, 0);
        if (++kc == cpt) { kc = 0; ks += tstep; if (ks >= p.S) { ks = s0; kr += tstep; } }
    };
    auto stash = [&](int stage, const u32x4 (&ar)[A_LOADS], const u32x4 (&br)[B_LOADS]) {
        unsigned char* As = smem + stage * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<u32x4*>(As + lds_off<BK>(row0 + i * RPP, ch)) = ar[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            *reinterpret_cast<u32x4*>(Bs + lds_off<BK>(row0 + i * RPP, ch)) = br[i];
    };

    typedef __attribute__((address_space(3))) void* lds_void_ptr;
    auto fetch_lds = [&](int stage) {   // GLDS: global -> LDS without a register stop
        const unsigned As = lds0 + (unsigned)(stage * STAGE + wid_u * (64 / CPR) * (BK * 2)), Bs = As + A_BYTES;
        if (DUAL && kfetch >= KT1) {        // second segment: plain rows of x2 / w2, C2 channels
            const int k2 = (kfetch - KT1) * (BK * 2);
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i)
                lds_dma16(xq2, As + i * RPP * (BK * 2), a_voff2[DUAL ? i : 0], (unsigned)k2);
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                lds_dma16(wq2, Bs + i * RPP * (BK * 2), b_off2[DUAL ? i : 0], (unsigned)k2);
            ++kfetch;
            return;
        }
        if (DUAL) ++kfetch;
        if (kc == 0) retap();
        const int kco = kc * (BK * 2);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            lds_dma16(xq, As + i * RPP * (BK * 2), a_voff[i], (unsigned)kco);
        // WT: tap (kr, ks) of the backward conv is tap (R-1-kr, S-1-ks) of the forward filter
        const int wk = WT ? (kc * BK * RSN + ((p.R - 1 - kr) * p.S + (p.S - 1 - ks)) * p.N) * 2
                          : ((kr * p.S + ks) * p.C + kc * BK) * 2;
        const bool wlast = WT && kc == cpt - 1;   // uniform
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            lds_dma16(wq, Bs + i * RPP * (BK * 2), wlast ? b_off_last[WT ? i : 0] : b_off[i], (unsigned)wk);
        if (++kc == cpt) { kc = 0; ks += tstep; if (ks >= p.S) { ks = s0; kr += tstep; } }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
#ifdef DDL_CONV_PROBES
    const __bf16 probe_seed = (__bf16)(float)(p.M & 3);
#endif
    auto compute = [&](int stage) {
        const unsigned char* As = smem + stage * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 xa[FM], wb[FN];
#ifdef DDL_CONV_PROBES
            if (p.probe & 16) {
#pragma unroll
                for (int j = 0; j < FM; ++j) xa[j] = (bf16x8)(probe_seed + j);
#pragma unroll
                for (int i = 0; i < FN; ++i) wb[i] = (bf16x8)(probe_seed + i);
            } else
#endif
            {
#pragma unroll
            for (int j = 0; j < FM; ++j)
                xa[j] = *reinterpret_cast<const bf16x8*>(As + lds_off<BK>(wm * WM + j * 16 + fr, kk * 4 + fq));
            if (WT) {
                // transposing read: lane 16g + 4q + pq supplies reduction row 8g + q (+4), columns 4pq..4pq+3 of
                // the 16-channel window; lane i of the group receives channel i with 4 consecutive reduction rows
                const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
                const int trow = kk * 32 + 8 * tg + tq;
                const int tswz = wt_swz<WROW>(trow);
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    unsigned char* b0 = const_cast<unsigned char*>(Bs) + trow * WROW + (((((wn * WN) >> 4) + i) ^ tswz) << 5) + 8 * tp;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(b0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(b0 + 4 * WROW));
                    wb[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            } else {
#pragma unroll
            for (int i = 0; i < FN; ++i)
                wb[i] = *reinterpret_cast<const bf16x8*>(Bs + lds_off<BK>(wn * WN + i * 16 + fr, kk * 4 + fq));
            }
            }
#ifdef DDL_CONV_PROBES
            if (p.probe & 4) {
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) { acc[i][j][0] += (float)wb[i][0] * (float)xa[j][0]; }
            } else
#endif
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], xa[j], acc[i][j], 0, 0, 0);
        }
    };

    // one tile of loads in flight per block (deeper prefetch measured twice - a second register
    // set, and a three-buffer LDS-DMA ring with counted vmcnt - and bought nothing at 2-3
    // workgroups per CU: profiles/r1_notes.md)
    u32x4 a0[A_LOADS], b0[B_LOADS];
    DDL_STAMP(1);
    if (GLDS && NSTG > 2) {
        // ring: NSTG-1 tiles in flight; tile kt+1 is waited for with a COUNTED vmcnt that leaves the younger
        // tiles in flight (every tile is A_LOADS + B_LOADS DMA instructions per thread), raw barrier; the last
        // NSTG-1 iterations issue nothing and drain
        constexpr int PRE = NSTG - 1, LPT = A_LOADS + B_LOADS, KEEP = (PRE - 1) * LPT;
        static_assert(KEEP < 64, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int s0_ = 0; s0_ < PRE; ++s0_)
            if (s0_ < KT) fetch_lds(s0_);
        if (KT >= PRE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0, nxt = PRE;
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = kt + PRE < KT;   // uniform
            if (more) fetch_lds(nxt);
            compute(cur);
            if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KEEP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            nxt = cur;
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    } else if (GLDS) {
        // two LDS buffers, the next tile's DMA in flight while this tile is consumed; the wait
        // before the barrier retires it (the barrier then publishes it to the other waves)
        if (KT > 0) fetch_lds(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) fetch_lds(cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
    if (KT > 0) {
        fetch(a0, b0);
        stash(0, a0, b0);
    }
    __syncthreads();
    DDL_STAMP(2);
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
#if defined(DDL_CONV_PROBES) && DDL_CONV_PROBES >= 2
        const bool mid = kt == (KT >> 1);
        if (mid) DDL_STAMP(5);
        if (kt + 1 < KT) fetch(a0, b0);
        if (mid) DDL_STAMP(6);
        compute(cur);
        if (mid) DDL_STAMP(7);
        if (kt + 1 < KT) stash(cur ^ 1, a0, b0);
        if (mid) DDL_STAMP(8);
        __syncthreads();
        if (mid) DDL_STAMP(9);
        continue;
#endif
        if (kt + 1 < KT) fetch(a0, b0);
        compute(cur);
#ifdef DDL_CONV_PROBES
        if (p.probe & 8) {  // keep the loads alive without the LDS stores
            if (kt + 1 < KT && a0[0][0] == 0x12345678u && b0[0][0] == 0x9abcdef0u) stash(cur ^ 1, a0, b0);
        } else
#endif
        if (kt + 1 < KT) stash(cur ^ 1, a0, b0);
#ifdef DDL_CONV_PROBES
        if (!(p.probe & 32))
#endif
        __syncthreads();
    }
    }

    DDL_STAMP(3);
    // ---- epilogue: acc -> bf16 tile in LDS ([BM][BN], row stride OSTR) -> coalesced stores ----
    unsigned char* Os = smem;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            bf16x4 v;
            int m = wm * WM + j * 16 + fr, n = wn * WN + i * 16 + fq * 4;
            if (DUAL && p.bias != nullptr) {   // uniform
                const int ng = n0 + n < p.N ? n0 + n : 0;   // N % 8 == 0: a group of 4 is inside or outside as a whole
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ng);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[i][j][e] + bv[e]);
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[i][j][e]);
            }
            *reinterpret_cast<bf16x4*>(Os + m * OSTR + n * 2) = v;
        }
    __syncthreads();
    constexpr int OCPR = BN / 8, ORPP = 256 / OCPR, OITER = BM / ORPP;
    const int oc = tid % OCPR, orow0 = tid / OCPR;
    const int n = n0 + oc * 8;
    float s1[8], s2[8], s3[8], fsc[8], fsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; s3[e] = 0.f; fsc[e] = 0.f; fsh[e] = 0.f; }
    static_assert(!LEAN || EPI == 4, "LEAN is a specialisation of the block-input-gradient epilogue");
    const bool y2on = !LEAN && EPI == 4 && p.ep_y2 != nullptr;  // uniform
    const bool maskbits = LEAN || ((EPI == 3 || EPI == 4) && p.ep_maskbits != nullptr);  // uniform
    if (EPI == 2 && n < p.N) { load8f(p.ep_fcoef + n, fsc); load8f(p.ep_fcoef + p.N + n, fsh); }
    // Rows are handled in batches: all global reads of a batch (shortcut gradient, ReLU mask, BN
    // input) are issued back to back with clamped addresses — no branch and no wait between them —
    // and only then consumed.  (One branchy row at a time made hipcc emit `s_waitcnt vmcnt(0)`
    // after every load: 2-3 exposed memory round trips per row, 3.3 TB/s on the layer-1 EPI-4
    // kernel against 4.5 TB/s for a plain streaming kernel — profiles/r1_notes.md.)
#ifndef DDL_EPI_BATCH
#define DDL_EPI_BATCH 4   // rows per thread whose epilogue reads are in flight together (8 measured: profiles/r1_notes.md)
#endif
    constexpr int BATCH = OITER < DDL_EPI_BATCH ? OITER : DDL_EPI_BATCH;
    const bool col_ok = n < p.N;
    constexpr int SUB = BM / 128, SUB_ITERS = OITER / SUB;  // a thread's rows of one 128-row sub-tile are consecutive iterations
    static_assert(SUB_ITERS % BATCH == 0, "a batch never straddles two sub-tiles");
    auto emit_stats = [&](int sub) {
        // lanes with equal (tid % OCPR) hold the same channels: fold the wave, then the 4 waves
        float* red = reinterpret_cast<float*>(smem + BM * OSTR);  // [4 waves][SROWS][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = OCPR; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
                if (EPI == 4) s3[e] += __shfl_xor(s3[e], o, 64);
            }
        }
        if (sub > 0) __syncthreads();  // the previous sub-tile's fold has been read
        float* r1 = red + wid * SROWS * BN;
        if (lane < OCPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                r1[lane * 8 + e] = s1[e]; r1[BN + lane * 8 + e] = s2[e];
                if (EPI == 4) r1[2 * BN + lane * 8 + e] = s3[e];
            }
        }
        __syncthreads();
        const int srow = tile_m * SUB + sub;
        if (tid < 2 * BN && srow < p.stat_rows) {
            float t = red[tid] + red[SROWS * BN + tid] + red[2 * SROWS * BN + tid] + red[3 * SROWS * BN + tid];
            int which = tid / BN, c = tid - which * BN;
            if (n0 + c < p.N) p.stats[((long long)srow * 2 + which) * p.N + n0 + c] = t;
            if (y2on && n0 + c < p.N) {  // second table: row 0 = the same sum g, row 1 = sum g*y2
                int src = which ? 2 * BN + c : c;
                float u = red[src] + red[SROWS * BN + src] + red[2 * SROWS * BN + src] + red[3 * SROWS * BN + src];
                p.stats2[((long long)srow * 2 + which) * p.N + n0 + c] = u;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; s3[e] = 0.f; }
    };
    // FULL: the tile lies inside the problem (every tile of the training step's large layers): no edge
    // predicates, no clamped addresses, no "live" multiplies in the sums - the epilogue of the short-reduction
    // layers is bound by vector-instruction issue (profiles/r2_notes.md section 19)
    auto rows = [&](auto full_c, int it0) {
        constexpr bool FULL = decltype(full_c)::value;
        long long off[BATCH], aoff[BATCH];
        bool ok[BATCH], has[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            int row = orow0 + (it0 + b) * ORPP, m = m0 + row;
            ok[b] = FULL ? true : ((m < m_lim) & col_ok);
            if (p.parity) {  // class-local row -> pixel row of the output image
                int bb = m / hw, rem = m - bb * hw;
                int hh = rem / Wq, wh = rem - hh * Wq;
                m = (bb * p.Hout + 2 * hh + ph) * p.Wout + 2 * wh + pw;
            }
            long long o = (long long)m * p.N + n;
            off[b] = (FULL || ok[b]) ? o : 0;
            has[b] = false; aoff[b] = 0;
            if (ADDEND) {
                long long ao = o;
                bool h = ok[b];
                if (!FULL && p.addend_sub2) {
                    int img = p.Hout * p.Wout;
                    int bb = m / img, rem = m - bb * img;
                    int hi = rem / p.Wout, wi = rem - hi * p.Wout;
                    h = h & (((hi | wi) & 1) == 0);
                    ao = ((long long)(bb * (p.Hout >> 1) + (hi >> 1)) * (p.Wout >> 1) + (wi >> 1)) * p.N + n;
                }
                has[b] = h; aoff[b] = h ? ao : 0;
                if (FULL) { has[b] = true; aoff[b] = o; }   // (FULL launches have no sub-sampled addend)
            }
        }
        bf16x8 av[BATCH], zm[BATCH], yv[BATCH], y2v[BATCH];
        if (ADDEND) {
#pragma unroll
            for (int b = 0; b < BATCH; ++b) av[b] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.addend + aoff[b]));
        }
        unsigned zb[BATCH];
        if (EPI == 3 || EPI == 4) {
            if (maskbits) {
#pragma unroll
                for (int b = 0; b < BATCH; ++b) zb[b] = p.ep_maskbits[off[b] >> 3];
            } else {
#pragma unroll
                for (int b = 0; b < BATCH; ++b) zm[b] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.ep_mask + off[b]));
            }
        }
        if (EPI == 2 || EPI == 4) {
#pragma unroll
            for (int b = 0; b < BATCH; ++b) yv[b] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.ep_y + off[b]));
        }
        if (y2on) {
#pragma unroll
            for (int b = 0; b < BATCH; ++b) y2v[b] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.ep_y2 + off[b]));
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            int row = orow0 + (it0 + b) * ORPP;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(Os + row * OSTR + oc * 16);
            if (ADDEND) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = has[b] ? f2bf(bf2f(v[e]) + bf2f(av[b][e])) : v[e];
            }
            if (EPI == 3 || EPI == 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bool keep = maskbits ? ((zb[b] >> e) & 1u) != 0 : bf2f(zm[b][e]) > 0.f;
                    v[e] = keep ? v[e] : f2bf(0.f);
                }
            }
            const float live = (FULL || ok[b]) ? 1.f : 0.f;   // rows past the edge add nothing to the sums
            if (EPI == 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float g = FULL ? bf2f(v[e]) : bf2f(v[e]) * live; s1[e] += g; s2[e] += g * bf2f(yv[b][e]); }
                if (y2on) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s3[e] += (FULL ? bf2f(v[e]) : bf2f(v[e]) * live) * bf2f(y2v[b][e]);
                }
            }
            if (EPI == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float yf = bf2f(yv[b][e]);
                    bool on = bn_affine(yf, fsc[e], fsh[e]) > 0.f;
                    v[e] = on ? v[e] : f2bf(0.f);
                    float g = bf2f(v[e]);
                    if (DUAL && !FULL) g *= live;   // (without the bias of the two-segment form rows past the edge are exact zeros)
                    s1[e] += g; s2[e] += g * yf;
                }
            }
            if (EPI == 1) {
                // (rows and columns past the edge are exact zeros here: their operand rows were
                // zero-filled by the buffer range check and the forward launch has no bias and no addend)
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = bf2f(v[e]); s1[e] += f; s2[e] += f * f; }
            }
            if (FULL || ok[b]) *reinterpret_cast<bf16x8*>(p.y + off[b]) = v;
        }
    };
    const bool full_tile = (m0 + BM <= m_lim) && (n0 + BN <= p.N) && !(ADDEND && p.addend_sub2);   // uniform
#pragma unroll
    for (int it0 = 0; it0 < OITER; it0 += BATCH) {
        if (full_tile) rows(std::true_type{}, it0);
        else rows(std::false_type{}, it0);
        if (STATS && (it0 + BATCH) % SUB_ITERS == 0) emit_stats((it0 + BATCH) / SUB_ITERS - 1);
    }
#ifdef DDL_CONV_PROBES
    DDL_STAMP(4);
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + (long long)blockIdx.x * 8;
        d[0] = stamp[1] - stamp[0];   // prologue (address setup)
        d[1] = stamp[2] - stamp[1];   // first tile fetch + stash + barrier
        d[2] = stamp[3] - stamp[2];   // main loop
        d[3] = stamp[4] - stamp[3];   // epilogue
        d[4] = stamp[6] - stamp[5];   // mid iteration: fetch issue
        d[5] = stamp[7] - stamp[6];   //                compute (LDS reads + MFMA issue)
        d[6] = stamp[8] - stamp[7];   //                wait for loads + LDS stores
        d[7] = stamp[9] - stamp[8];   //                barrier
    }
#endif
}

template <int BM, int BN, int BK, bool GLDS, bool WT>
int launch_cfg2(const IGemmParams& p, hipStream_t stream) {
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    static const int extra_lds = getenv("DDL_CONV_EXTRA_LDS") ? atoi(getenv("DDL_CONV_EXTRA_LDS")) : 0;  // occupancy probe
    const bool mask = p.ep_mask || p.ep_maskbits;
    if (p.stats && p.addend && !(p.ep_y && mask)) return DDL_EINVAL;
#define DDL_LAUNCH(E, A) DDL_KLAUNCH((conv_igemm_kernel<BM, BN, BK, E, A, GLDS, WT>), grid, block, extra_lds, stream, p)
    const bool lean = p.ep_maskbits && !p.ep_y2;
#define DDL_LAUNCH_LEAN(A) DDL_KLAUNCH((conv_igemm_kernel<BM, BN, BK, 4, A, GLDS, WT, false, true>), grid, block, extra_lds, stream, p)
    if (p.ep_y && mask && p.addend) { if (lean) DDL_LAUNCH_LEAN(true); else DDL_LAUNCH(4, true); }
    else if (p.ep_y && mask) { if (lean) DDL_LAUNCH_LEAN(false); else DDL_LAUNCH(4, false); }
    else if (p.ep_y) DDL_LAUNCH(2, false);
    else if (mask && p.addend) DDL_LAUNCH(3, true);
    else if (mask) DDL_LAUNCH(3, false);
    else if (p.stats) {
        if constexpr (WT) return DDL_EINVAL;   // forward statistics belong to the forward weight layout
        else DDL_LAUNCH(1, false);
    }
    else if (p.addend) DDL_LAUNCH(0, true);
    else DDL_LAUNCH(0, false);
#undef DDL_LAUNCH
#undef DDL_LAUNCH_LEAN
    return ddl_launch_status();
}

template <int BM, int BN, int BK>
int launch_cfg(const IGemmParams& p, hipStream_t stream) {
    // Operand tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR stop, no
    // ds_write pass (13 cycles per wave-instruction) - count-weighted fwd 3.74 -> 3.58 ms, dgrad
    // 3.47 -> 3.28 ms, the BK=64 tiles gain most (d4 3x3 512 @7: 78 -> 61 us = 974 TFLOP/s).
    // DDL_CONV_GLDS=0 selects the register-staged path (global -> VGPR -> ds_write_b128).
    static const int glds = getenv("DDL_CONV_GLDS") ? atoi(getenv("DDL_CONV_GLDS")) : 1;  // tuning knob
    if (p.x2) {   // two-segment reduction: plain store or the EPI-2 epilogue, LDS-DMA only
        dim3 grid(p.tiles_m * p.tiles_n), block(256);
        if (p.ep_y) DDL_KLAUNCH((conv_igemm_kernel<BM, BN, BK, 2, false, true, false, true>), grid, block, 0, stream, p);
        else DDL_KLAUNCH((conv_igemm_kernel<BM, BN, BK, 0, false, true, false, true>), grid, block, 0, stream, p);
        return ddl_launch_status();
    }
    if (p.wt) return launch_cfg2<BM, BN, BK, true, true>(p, stream);   // transposed weight image: LDS-DMA only
    return glds ? launch_cfg2<BM, BN, BK, true, false>(p, stream) : launch_cfg2<BM, BN, BK, false, false>(p, stream);
}

}  // namespace

// the 256 x 256 / 8-wave plain-GEMM kernel of the wide 1x1 layers (gemm256.hip)
extern "C" int ddl_gemm256_ok(long long M, int N, int K);
extern "C" int ddl_gemm256_fwd(const void* x, const void* w, void* y, void* stats, long long M, int N, int K,
                               long long x_numel, int stat_rows, hipStream_t stream);

static unsigned long long* g_conv_dbg = nullptr;
// probe builds (-DDDL_CONV_PROBES): every launch writes [block][8] cycle stamps of wave 0 here
extern "C" void ddl_conv_set_debug_buffer(void* p) { g_conv_dbg = (unsigned long long*)p; }

// Number of M tiles the forward kernel uses for a problem (so callers can size `stats`).
extern "C" int ddl_conv_igemm_tiles_m(long long M) { return (int)((M + 127) / 128); }

static bool igemm_parity(int idil, int ostride, int Hout, int Wout) {
    return idil == 2 && ostride == 1 && (Hout % 2) == 0 && (Wout % 2) == 0;
}

// M tiles of a backward-data launch (parity-class ordering pads every class to whole tiles).
extern "C" int ddl_conv_igemm_tiles_m_bwd(long long M, int Hout, int Wout, int idil, int ostride) {
    if (igemm_parity(idil, ostride, Hout, Wout)) return 4 * (int)((M / 4 + 127) / 128);
    return (int)((M + 127) / 128);
}

// `wt` = 1: `w` is the FORWARD filter w[k][r][s][n] of the convolution whose input gradient this
// launch computes (C = its output channels k, N = its input channels n); the kernel reads it through
// transposing LDS loads, so no flipped/transposed copy of the weights exists anywhere.
extern "C" int ddl_conv_igemm_ex4(const void* x, const void* w, void* y, const void* addend, void* stats,
                                  long long x_numel, long long w_numel,
                                  int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                  int Hout, int Wout, int N, int R, int S, int ostride, int pad, int pad_w, int idil,
                                  int addend_sub2, const void* ep_y, const void* ep_fcoef, const void* ep_mask,
                                  const void* ep_y2, void* stats2, const void* ep_maskbits, int wt, int wt_rows,
                                  const void* x2, const void* w2, long long x2_numel, long long w2_numel, int C2,
                                  const void* bias, hipStream_t stream) {
    if (!x || !w || !y || B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || N <= 0 || R <= 0 || S <= 0) return DDL_EINVAL;
    if (C % 32 != 0 || N % 8 != 0) return DDL_EINVAL;
    if (idil != 1 && idil != 2 && idil != 4) return DDL_EINVAL;
    if (ostride <= 0 || pad < 0 || pad_w < 0 || Hout <= 0 || Wout <= 0) return DDL_EINVAL;
    if (x_numel <= 0 || w_numel * 2 >= 0x7fffffffLL) return DDL_EINVAL;
    if (!wt && w_numel < (long long)N * R * S * C) return DDL_EINVAL;
    if (wt && (wt_rows <= 0 || wt_rows > C || wt_rows <= C - 32 || w_numel < (long long)N * R * S * wt_rows)) return DDL_EINVAL;
    if (wt && (stats && !ep_y)) return DDL_EINVAL;
    // 32-bit offsets are relative to the first image a tile touches: a tile of <= 256 output rows
    // spans at most 256 / (Hout*Wout) + 2 images (tensors themselves may exceed 2 GiB)
    {
        long long hw_min = (long long)Hout * Wout / 4;   // a parity class holds a quarter of the image's pixels
        if (hw_min < 1) hw_min = 1;
        if ((256 / hw_min + 3) * xs_b * 2 >= 0x7fffffffLL || 257 * xs_w * 2 >= 0x7fffffffLL) return DDL_EINVAL;
    }
    if (!ddl_aligned16(x) || !ddl_aligned16(w) || !ddl_aligned16(y) || (addend && !ddl_aligned16(addend))) return DDL_EALIGN;
    if ((ep_y && !ddl_aligned16(ep_y)) || (ep_fcoef && !ddl_aligned16(ep_fcoef)) || (ep_mask && !ddl_aligned16(ep_mask))) return DDL_EALIGN;
    // fused BatchNorm-backward reduce needs y, the forward coefficients and the partial buffer together
    // (EPI 2: ep_y + ep_fcoef, no addend; EPI 4: ep_y + ep_mask, addend allowed)
    if (ep_mask && ep_maskbits) return DDL_EINVAL;
    const bool has_mask = ep_mask || ep_maskbits;
    if (ep_fcoef && (!ep_y || has_mask || addend)) return DDL_EINVAL;
    if (ep_y && (!stats || (ep_fcoef == nullptr) == !has_mask)) return DDL_EINVAL;
    if (has_mask && stats && !ep_y) return DDL_EINVAL;
    // second BatchNorm of the same gradient: EPI 4 only, with a table of its own
    if ((ep_y2 == nullptr) != (stats2 == nullptr) || (ep_y2 && !(ep_y && has_mask)) || (ep_y2 && !ddl_aligned16(ep_y2))) return DDL_EINVAL;
    if ((xs_b % 8) || (xs_h % 8)) return DDL_EALIGN;
    // every tap address must be 16-byte aligned: either pixels are >= 8 elements apart, or (S == 1,
    // undilated) only multiples of ostride are visited and those are
    if (xs_w % 8) {
        if (S != 1 || idil != 1 || ((long long)ostride * xs_w) % 8 || ((long long)pad_w * xs_w) % 8) return DDL_EALIGN;
    }
    long long M = (long long)B * Hout * Wout;
    if (M >= (1LL << 31)) return DDL_EINVAL;
    // the farthest element any valid tap can touch must lie inside the buffer
    long long last = (long long)(B - 1) * xs_b + (long long)(Hin - 1) * xs_h + (long long)(Win - 1) * xs_w + C;
    if (last > x_numel) return DDL_EINVAL;
    IGemmParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y;
    p.addend = (const bf16_t*)addend; p.stats = (float*)stats;
    p.ep_y = (const bf16_t*)ep_y; p.ep_fcoef = (const float*)ep_fcoef; p.ep_mask = (const bf16_t*)ep_mask;
    p.ep_y2 = (const bf16_t*)ep_y2; p.stats2 = (float*)stats2;
    p.ep_maskbits = (const unsigned char*)ep_maskbits;
    p.x_bytes = x_numel * 2; p.w_bytes = (unsigned)(w_numel * 2);
    p.wt = wt ? 1 : 0;
    p.wt_rows = wt ? wt_rows : C;
    p.x2 = (const bf16_t*)x2; p.w2 = (const bf16_t*)w2; p.bias = (const float*)bias;
    p.x2_bytes = x2_numel * 2; p.w2_bytes = (unsigned)(w2_numel * 2); p.C2 = C2;
    if (x2) {
        // second reduction segment: plain GEMM launches only (rows of x2 are the output rows)
        const bool plain = R == 1 && S == 1 && ostride == 1 && pad == 0 && pad_w == 0 && idil == 1 && Hout == Hin && Wout == Win &&
                           xs_h == (long long)Win * xs_w && xs_b == (long long)Hin * xs_h;
        if (!plain || wt || addend || ep_mask || ep_maskbits || ep_y2 || !w2 || C2 <= 0 || C2 % 64 != 0) return DDL_EINVAL;
        if (x2_numel < (long long)B * Hout * Wout * C2 || w2_numel < (long long)N * C2 || w2_numel * 2 >= 0x7fffffffLL) return DDL_EINVAL;
        if (!ddl_aligned16(x2) || !ddl_aligned16(w2) || (bias && !ddl_aligned16(bias))) return DDL_EALIGN;
        if (stats && !ep_y) return DDL_EINVAL;
        if ((long long)257 * C2 * 2 >= 0x7fffffffLL) return DDL_EINVAL;
    } else if (w2 || bias) return DDL_EINVAL;
    // timing-only probe (results are wrong): a descriptor with zero records drops every load through
    // it while the instruction stream, waits and barriers stay - prices one operand's traffic
    static const int probe = getenv("DDL_CONV_PROBE") ? atoi(getenv("DDL_CONV_PROBE")) : 0;
    if (probe & 1) p.x_bytes = 0;
    if (probe & 2) p.w_bytes = 0;
    p.probe = probe;
    p.dbg = g_conv_dbg;
    p.Hin = Hin; p.Win = Win; p.C = C; p.xs_b = xs_b; p.xs_h = xs_h; p.xs_w = xs_w;
    p.Hout = Hout; p.Wout = Wout; p.N = N; p.M = (int)M;
    p.R = R; p.S = S; p.ostride = ostride; p.pad = pad; p.pad_w = pad_w;
    p.idil_log2 = idil == 1 ? 0 : (idil == 2 ? 1 : 2);
    p.simple = (R == 1 && S == 1 && ostride == 1 && pad == 0 && pad_w == 0 && idil == 1 && Hout == Hin && Wout == Win &&
                xs_h == (long long)Win * xs_w && xs_b == (long long)Hin * xs_h) ? 1 : 0;
    if (addend_sub2 && (!addend || (Hout & 1) || (Wout & 1))) return DDL_EINVAL;
    p.addend_sub2 = addend_sub2;
    // wide plain GEMMs without a fused backward epilogue (forward 1x1 convs of stages 3-4, with or without
    // statistics): the 8-wave ring kernel
    if (p.simple && xs_w == C && !addend && !ep_y && !has_mask && !x2 && !wt && !probe && !g_conv_dbg && ddl_gemm256_ok(M, N, C))
        return ddl_gemm256_fwd(x, w, y, stats, M, N, C, x_numel, (int)((M + 127) / 128), stream);
    p.tiles_m = (int)((M + 127) / 128);
    p.stat_rows = p.tiles_m;
    p.parity = 0; p.tiles_per_class = 0; p.Mc = 0;
    if (igemm_parity(idil, ostride, Hout, Wout) && (!stats || ep_y)) {
        p.parity = 1;
        p.Mc = (int)(M / 4);
        p.tiles_per_class = (p.Mc + 127) / 128;
        p.tiles_m = 4 * p.tiles_per_class;
        p.stat_rows = p.tiles_m;
    }
    // Tile per shape class, from benchmarks/kernel_bench.py sweeps with LDS-DMA staging on MI355X
    // (profiles/r1_notes.md, r1_kernel_bench_glds_sweep.log).  DDL_CONV_BK=32|64 and
    // DDL_CONV_BM=128|256 force a variant.
    static const int force_bk = getenv("DDL_CONV_BK") ? atoi(getenv("DDL_CONV_BK")) : 0;  // tuning knob
    static const int force_bm = getenv("DDL_CONV_BM") ? atoi(getenv("DDL_CONV_BM")) : 0;  // tuning knob
    const long long RSC = (long long)R * S * C;
    // BK = 64 (one barrier per 64 reduction elements, 64 KB of LDS per workgroup): every 3x3 (the
    // parity-ordered stride-2 backward only from 512 channels), 1x1 from 2048 channels, and the
    // narrow-output 1x1 (N <= 64) from 256
    static const int deep_min_c = getenv("DDL_CONV_DEEP_MIN_C") ? atoi(getenv("DDL_CONV_DEEP_MIN_C")) : 2048;  // tuning knob
    bool deep = (R * S == 1) ? (C >= deep_min_c || (N <= 64 && C >= 256)) : (p.parity ? C >= 512 : true);
    // 256x128 block tile, BK = 32 (wave tile 128x64: a quarter fewer LDS and L2->LDS bytes per FLOP)
    // for exactly two N tiles sharing the A panel, a long reduction and enough tiles to fill the
    // chip: c4 3x3 256->256 @14 fwd 71 (BK=64) -> 65 us, dgrad 68 -> 62 us.  With more N tiles, short
    // reductions or < 1.5 tiles per CU it loses (no stream-K here).
    const long long tiles256 = ((M + 255) / 256) * ((N + 127) / 128);
    bool big = !p.parity && N >= 256 && N < 512 && RSC >= 1024 && tiles256 >= 384;
    if (force_bm == 128) big = false;
    if (force_bm == 256 && !p.parity && N > 64) big = true;
    if (force_bk == 32) deep = false;
    if (force_bk == 64) deep = true;
    if (C % 64) deep = false;
    static const int force_bn = getenv("DDL_CONV_BN") ? atoi(getenv("DDL_CONV_BN")) : 0;  // tuning knob (64: narrow tiles everywhere)
    if (N <= 64 || force_bn == 64) {
        p.tiles_n = (N + 63) / 64;
        // (a 256x64 tile measured 5-12 % slower than 128x64 on the N = 64 layers: profiles/r1_notes.md)
        return deep ? launch_cfg<128, 64, 64>(p, stream) : launch_cfg<128, 64, 32>(p, stream);
    }
    p.tiles_n = (N + 127) / 128;
    if (big) {
        p.tiles_m = (int)((M + 255) / 256);
        return launch_cfg<256, 128, 32>(p, stream);  // (a BK=64 twin needs 96 KB of LDS: one workgroup per CU, 15-30 % slower)
    }
    return deep ? launch_cfg<128, 128, 64>(p, stream) : launch_cfg<128, 128, 32>(p, stream);
}
