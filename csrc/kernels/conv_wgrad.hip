// K4: convolution backward-weights as a split-K implicit GEMM on bf16 MFMA (gfx950).
//
//   dw[n][r][s][c] = sum_m dy[m][n] * x[b, ho*stride - pad + r, wo*stride - pad + s, c]
//
// GEMM view: rows = output channels n, cols = (r, s, c), reduction = m = (b, ho, wo) — the
// *slow* axis of both operands.  Tiles of dy [32 m][BMw n] and x [32 m][BNw c] are staged
// row-major (coalesced 16-byte buffer loads, out-of-image taps read as zero) into padded LDS
// rows, and the MFMA fragments — which need 8 consecutive reduction elements per lane — are read
// with the gfx950 transposing LDS load ds_read_b64_tr_b16 (no register shuffles, no transposed
// copy of the activations in memory).  The reduction index inside one 32-deep MFMA step is
// permuted identically for both operands (lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3)
// so that the 8 rows one half-wave touches land in 8 distinct 32-byte bank windows.
//
// The reduction (up to B*Ho*Wo = 3.2 M for the stem) is split over workgroups; each writes an
// fp32 partial slab and `ddl_splitk_reduce` sums the slabs in a fixed order (deterministic,
// no float atomics) into the gradient arena in bf16 or fp32.
#include "common.h"
#include <cstdlib>

namespace {

struct WgradParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* partial;  // [splits][K][RSC]
    long long dy_bytes, x_bytes;   // whole tensors (may exceed 2 GiB): every split addresses them relative to its first row / image
    int Hin, Win, C;
    long long xs_b, xs_h, xs_w;
    int Hout, Wout, K, M;
    int R, S, stride, pad, pad_w;
    int tiles_k, tiles_rsc, splits, m_per_split;  // tiles over K and over the flattened (r, s, c) axis
    int simple;  // x row of pixel m is x + m*xs_w (1x1, stride 1, no padding)
    // Interleaved reduction ranges (plain GEMMs, 0 = off): the reduction rows are cut into 8 groups of il_mg
    // rows (one per XCD, as the backward-data kernel's tile order cuts them) and the il_sg splits of a group
    // take its 32/64-row tiles round-robin, so the whole grid sweeps every group front to back - the
    // order in which the backward-data kernel running next to it reads the same two tensors
    int il_sg, il_mg;
    // COLSUM kernels (plain GEMMs): [splits][2][C] column sums of x over each split's rows, both rows alike (the
    // layout of a statistics table: bn_colsum_fold sums it) - the matrix pipe adds them up on the side (a row of
    // ones as a 17th "output channel" fragment), so the Gram launch of the linear path needs no extra pass over x
    float* colsum;
    // Row-streaming kernel, XF form: x is the RAW output of the conv in front; its BatchNorm apply + ReLU
    // ([2][C] scale | shift) happen on the staged rows (the activated tensor is never materialised)
    const float* x_fcoef;
    const bf16_t* dy_y;      // row-streaming kernel, K = 64: dy is the gradient w.r.t. a BatchNorm's OUTPUT, dy_y that
    const float* dy_coef;    // BatchNorm's input, dy_coef [3][64] ca | cb | cc: the kernel multiplies ca*dy + cb*dy_y + cc
};

__device__ __forceinline__ u32x4 wg_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
}


typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// swizzle of the LDS-DMA image: 32-byte window w of row r is stored at window w ^ lds_swz(r), so
// that the 8 rows one half-wave's transposing read touches fall into 8 distinct 32-byte windows of
// the 256-byte bank row (what the 32-byte row padding does for the register-staged image; an
// LDS-DMA image cannot be padded, its wave-instruction writes 1 KiB contiguously)
template <int ROW_BYTES>
__device__ __forceinline__ int lds_swz(int row) {
    return ROW_BYTES == 256 ? (row & 7) : ((row >> 1) & 3);
}

template <int BMw, int BNw, bool SIMPLE, bool UNITAP, bool GLDS, int BK = 32, bool COLSUM = false>
__global__ void __launch_bounds__(256, COLSUM ? 3 : 4) conv_wgrad_kernel(const WgradParams p) {
    static_assert(!GLDS || ((BMw == 64 || BMw == 128) && (BNw == 64 || BNw == 128)), "LDS-DMA image: 128- or 256-byte rows");
    static_assert(BK == 32 || (BK == 64 && GLDS), "64 reduction rows per tile only with LDS-DMA staging (no staging registers)");
    constexpr int A_STR = BMw * 2 + (GLDS ? 0 : 32), B_STR = BNw * 2 + (GLDS ? 0 : 32);  // row strides (bytes)
    constexpr int A_BYTES = BK * A_STR, B_BYTES = BK * B_STR;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_CPR = BMw / 8, B_CPR = BNw / 8;
    constexpr int A_LOADS = (BK * A_CPR + 255) / 256, B_LOADS = (BK * B_CPR + 255) / 256;
    constexpr int WM = BMw / 2, WN = BNw / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int ntile = p.tiles_k * p.tiles_rsc;
    const int nwg = ntile * p.splits;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / ntile;
    int t = bid - split * ntile;
    const int tile_k = t % p.tiles_k;
    const int tile_j = t / p.tiles_k;
    // columns of the tile = BNw consecutive entries of the flattened (r, s, c) axis: whole taps
    // when C divides BNw (several taps per tile: the dy tile is loaded once for all of them), a
    // channel slice of one tap when BNw divides C
    const int n0 = tile_k * BMw, j0 = tile_j * BNw;
    const int RSC = p.R * p.S * p.C;

    int m_begin = split * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    int kstep = 1;                       // tiles from one tile of this split to its next
    if (SIMPLE && p.il_sg > 0) {
        const int grp = split / p.il_sg, j = split - grp * p.il_sg;
        m_begin = grp * p.il_mg + j * BK;
        m_end = (grp + 1) * p.il_mg;
        kstep = p.il_sg;
    }
    if (m_end > p.M) m_end = p.M;
    const int KT = m_begin < m_end ? (m_end - m_begin + kstep * BK - 1) / (kstep * BK) : 0;

    // Tensors beyond 2 GiB: the descriptors start at this split's first reduction row (dy; x of a
    // plain GEMM) or first image (x), so 32-bit offsets only span one split.
    const int hw = p.Hout * p.Wout;
    const int b_first = m_begin / hw;
    const long long dy_first = (long long)m_begin * p.K;
    const long long x_first = SIMPLE ? (long long)m_begin * p.xs_w : (long long)b_first * p.xs_b;
    long long dy_left = p.dy_bytes - dy_first * 2, x_left = p.x_bytes - x_first * 2;
    dy_left = dy_left < 0 ? 0 : (dy_left > 0x7fffffffLL ? 0x7fffffffLL : dy_left);
    x_left = x_left < 0 ? 0 : (x_left > 0x7fffffffLL ? 0x7fffffffLL : x_left);
    const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + dy_first), 0, (int)dy_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + x_first), 0, (int)x_left, 0x00020000);
    // GLDS: the DMA is issued from inline asm (lds_dma16) - in its builtin form hipcc drains it (vmcnt(0)) in
    // front of the first transposing LDS read of every tile, i.e. the next tile's DMA never overlapped this
    // tile's MFMAs inside a workgroup
    const u32x4 dyq = raw_rsrc(p.dy + dy_first, (unsigned)dy_left), xq = raw_rsrc(p.x + x_first, (unsigned)x_left);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);

    // Per-thread load slots are fixed over the whole reduction: (row, chunk) never change and the
    // pixel index only advances by BK per tile.  dy offsets therefore ride in the buffer
    // instruction's scalar offset; the (b, ho, wo) decomposition of the x row is computed once
    // and then stepped (no divisions and no 64-bit arithmetic in the loop).
    u32x4 a_reg[A_LOADS], b_reg[B_LOADS];
    constexpr unsigned INVALID = 0x80000000u;
    unsigned a_voff[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        int idx = tid + i * 256;
        int row = idx / A_CPR, ch = idx - row * A_CPR;
        if (GLDS) ch = (((ch >> 1) ^ lds_swz<BMw * 2>(row)) << 1) | (ch & 1);  // physical position -> logical chunk
        int n = n0 + ch * 8;
        a_voff[i] = (row < BK && n < p.K) ? ((unsigned)row * (unsigned)p.K + (unsigned)n) * 2u : INVALID;
    }
    int b_wo[B_LOADS], b_ho[B_LOADS], b_dh[B_LOADS], b_dw[B_LOADS];
    unsigned b_base[B_LOADS];  // byte offset of (b, hi = ho*stride - pad + r, wi = 0, c)
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        int idx = tid + i * 256;
        int row = idx / B_CPR, ch = idx - row * B_CPR;
        if (GLDS) ch = (((ch >> 1) ^ lds_swz<BNw * 2>(row)) << 1) | (ch & 1);
        int j = j0 + ch * 8;                      // flattened (r, s, c) index of this 8-channel chunk
        // UNITAP (C is a multiple of BNw): the whole tile belongs to one filter tap
        int tap = (UNITAP ? j0 : j) / p.C, c = j - tap * p.C;
        int fr = tap / p.S, fs = tap - fr * p.S;
        const bool live = row < BK && j < RSC;
        b_dh[i] = fr - p.pad;
        b_dw[i] = fs - p.pad_w;
        int m = m_begin + row;
        if (SIMPLE) {
            b_base[i] = live ? ((unsigned)row * (unsigned)p.xs_w + (unsigned)c) * 2u : INVALID;
            b_wo[i] = 0; b_ho[i] = 0;
        } else {
            // once per thread and slot: exact integer divisions (an fp32 reciprocal is only exact below 2^24 rows)
            const int b = m / hw, rem = m - b * hw;
            const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
            b_wo[i] = wo;
            b_ho[i] = live ? ho : -(1 << 24);   // a dead slot never passes the hi >= 0 test
            b_base[i] = (unsigned)(((b - b_first) * (int)p.xs_b + (ho * p.stride + fr - p.pad) * (int)p.xs_h + c) * 2);
        }
    }
    const int step_w = BK % p.Wout, step_h = BK / p.Wout;         // BK pixels = step_h rows + step_w
    const int xs_w2 = (int)p.xs_w * 2;
    const int row_bytes = p.stride * (int)p.xs_h * 2;              // one output row down, in x bytes
    const int img_bytes = (int)p.xs_b * 2 - p.Hout * row_bytes;    // wrap to the next image
    typedef __attribute__((address_space(3))) void* lds_void_ptr;
    // GLDS: the tile goes global -> LDS by LDS-DMA; chunk idx = tid + i*256 of the dense image lands
    // at byte idx*16, i.e. wave base (wid*64 + i*256)*16 + lane*16
    auto fetch = [&](int kt, int stage) {
        kt *= kstep;
        const int rows_left = m_end - (m_begin + kt * BK);
        const int a_soff = kt * (BK * 2) * p.K;
        const unsigned Ad = lds0 + (unsigned)(stage * STAGE + wid_u * 1024), Bd = Ad + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = (tid + i * 256) / A_CPR;
            const int vo = (int)(row < rows_left ? a_voff[i] : INVALID);
            if (GLDS) lds_dma16(dyq, Ad + i * 4096, (unsigned)vo, (unsigned)a_soff);
            else a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(dyr, vo, a_soff, 0);
        }
        if (SIMPLE) {
            const int b_soff = kt * BK * xs_w2;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                const int row = (tid + i * 256) / B_CPR;
                const int vo = (int)(row < rows_left ? b_base[i] : INVALID);
                if (GLDS) lds_dma16(xq, Bd + i * 4096, (unsigned)vo, (unsigned)b_soff);
                else b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, b_soff, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                int hi = b_ho[i] * p.stride + b_dh[i], wi = b_wo[i] * p.stride + b_dw[i];
                const int row = (tid + i * 256) / B_CPR;
                bool ok = (row < rows_left) & (hi >= 0) & (hi < p.Hin) & (wi >= 0) & (wi < p.Win);
                const unsigned vo = ok ? b_base[i] + (unsigned)(wi * xs_w2) : INVALID;
                if (GLDS) lds_dma16(xq, Bd + i * 4096, vo, 0u);
                else b_reg[i] = wg_load16(xr, vo);
                // advance this slot by BK pixels
                int wo = b_wo[i] + step_w, ho = b_ho[i] + step_h;
                unsigned base = b_base[i] + (unsigned)(step_h * row_bytes);
                if (wo >= p.Wout) { wo -= p.Wout; ++ho; base += (unsigned)row_bytes; }
                while (ho >= p.Hout) { ho -= p.Hout; base += (unsigned)img_bytes; }
                b_wo[i] = wo; b_ho[i] = ho; b_base[i] = base;
            }
        }
    };
    auto stash = [&](int stage) {
        unsigned char* As = smem + stage * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int idx = tid + i * 256;
            int row = idx / A_CPR, ch = idx - row * A_CPR;
            if (row < BK) *reinterpret_cast<u32x4*>(As + row * A_STR + ch * 16) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int idx = tid + i * 256;
            int row = idx / B_CPR, ch = idx - row * B_CPR;
            if (row < BK) *reinterpret_cast<u32x4*>(Bs + row * B_STR + ch * 16) = b_reg[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cacc[COLSUM ? FN : 1];
    bf16x8 ones;
    if (COLSUM) {
#pragma unroll
        for (int j = 0; j < FN; ++j) cacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = f2bf(1.f);
    }

    // transposing-read addressing: lane 16g + 4q + pq supplies row (4g + q [+16]), cols 4pq..4pq+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int tr_row = 4 * g + q;

    const int swz_a = GLDS ? lds_swz<BMw * 2>(tr_row) : 0, swz_b = GLDS ? lds_swz<BNw * 2>(tr_row) : 0;
    if (KT > 0) {
        fetch(0, 0);
        if (!GLDS) stash(0);
    }
    if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) fetch(kt + 1, cur ^ 1);
        unsigned char* As = smem + cur * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {   // 32 reduction rows per MFMA step (rows r and r+32 share the XOR too)
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            // 16 columns = one 32-byte window; GLDS: windows XOR-swizzled per row (rows r and r+16 share the XOR)
            unsigned char* a0 = As + (tr_row + kk * 32) * A_STR + ((((wm * WM) >> 4) + i) ^ swz_a) * 32 + 8 * pq;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 16 * A_STR));
            af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            unsigned char* b0 = Bs + (tr_row + kk * 32) * B_STR + ((((wn * WN) >> 4) + j) ^ swz_b) * 32 + 8 * pq;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(b0));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(b0 + 16 * B_STR));
            bfr[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (COLSUM && wm == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j) cacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[j], cacc[j], 0, 0, 0);
        }
        }
        if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt + 1 < KT) stash(cur ^ 1);
        __syncthreads();
    }

    // D rows = n (4 per lane), cols = flattened (r, s, c) index (lane & 15)
    float* out = p.partial + (long long)split * p.K * RSC;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            int jc = j0 + wn * WN + j * 16 + fr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int n = n0 + wm * WM + i * 16 + fq * 4 + e;
                if (n < p.K && jc < RSC) out[(long long)n * RSC + jc] = acc[i][j][e];
            }
        }
    if (COLSUM && wm == 0 && tile_k == 0 && fq == 0) {   // every D row holds the column sums: lanes 0..15 have row 0
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int jc = j0 + wn * WN + j * 16 + fr;
            if (jc < RSC) {
                p.colsum[((long long)split * 2) * RSC + jc] = cacc[j][0];
                p.colsum[((long long)split * 2 + 1) * RSC + jc] = cacc[j][0];
            }
        }
    }
}

// ---- 256 x 256 tile, 8 waves, LDS-DMA ring with tiles in flight across the barrier ----------------
// The matrix-bound weight gradients (stages 3-4: 256..2048 channels on both sides; reduction 12 544 or
// 50 176 rows) ran at 600-730 TFLOP/s in the 128 x 128 / 4-wave kernel above: at the ceiling of a loop
// that drains its LDS-DMA (`vmcnt(0)`) and meets at a barrier every 32 reduction rows with one tile
// in flight.  This kernel is built the other way round (cdna_hip_programming.md section 5, "Pipelining
// across barriers"):
//   * one 256 x 256 fp32 tile per workgroup, 8 waves as 2 (n) x 4 (r,s,c); a wave owns 128 x 64 =
//     8 x 4 MFMA fragments (128 accumulator registers): 32 MFMAs per 24 transposing LDS reads and
//     per barrier instead of 16 per 16;
//   * NST LDS stages of 32 reduction rows (32 KB each: dy [32][256] + x [32][256]), filled by LDS-DMA
//     only; the DMA of tile t+NST-1 is issued before tile t is consumed and is waited for with a COUNTED
//     `s_waitcnt vmcnt(4 * (NST-2))` - never 0 inside the loop - followed by a raw `s_barrier`;
//   * a stage is overwritten one barrier after its last read retired (`lgkmcnt(0)` before that barrier).
// Rows are 512 bytes: window w (32 bytes) of row r sits at window w ^ (r & 7) as in the kernel above,
// applied on the SOURCE address of the DMA (the LDS image of one wave-instruction is linear) and on
// the read address.
template <bool SIMPLE, int NST, bool PIPE, bool COLSUM = false>
__global__ void __launch_bounds__(512) conv_wgrad256_kernel(const WgradParams p) {
    static_assert(!COLSUM || !PIPE, "column sums ride in the plain loop");
    constexpr int BK = 32, ROWB = 512, HALF = BK * ROWB, STAGE = 2 * HALF, CPR = 32, LOADS = 2;
    constexpr int FM = 8, FN = 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;

    const int ntile = p.tiles_k * p.tiles_rsc;
    const int nwg = ntile * p.splits;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / ntile;
    const int t = bid - split * ntile;
    const int tile_k = t % p.tiles_k, tile_j = t / p.tiles_k;
    const int n0 = tile_k * 256, j0 = tile_j * 256;
    const int RSC = p.R * p.S * p.C;

    const int m_begin = split * p.m_per_split;
    int m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;
    const int KT = (m_end - m_begin + BK - 1) / BK;

    const int hw = p.Hout * p.Wout;
    const int b_first = m_begin / hw;
    const long long dy_first = (long long)m_begin * p.K;
    const long long x_first = SIMPLE ? (long long)m_begin * p.xs_w : (long long)b_first * p.xs_b;
    long long dy_left = p.dy_bytes - dy_first * 2, x_left = p.x_bytes - x_first * 2;
    dy_left = dy_left < 0 ? 0 : (dy_left > 0x7fffffffLL ? 0x7fffffffLL : dy_left);
    x_left = x_left < 0 ? 0 : (x_left > 0x7fffffffLL ? 0x7fffffffLL : x_left);
    const u32x4 dyr = raw_rsrc(p.dy + dy_first, (unsigned)dy_left), xr = raw_rsrc(p.x + x_first, (unsigned)x_left);

    constexpr unsigned INVALID = 0x80000000u;
    unsigned a_voff[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int idx = tid + i * 512;
        const int row = idx / CPR;
        int ch = idx - row * CPR;
        ch = (((ch >> 1) ^ (row & 7)) << 1) | (ch & 1);          // physical position -> logical chunk
        const int n = n0 + ch * 8;
        a_voff[i] = (n < p.K) ? ((unsigned)row * (unsigned)p.K + (unsigned)n) * 2u : INVALID;
    }
    int b_wo[LOADS], b_ho[LOADS], b_dh[LOADS], b_dw[LOADS];
    unsigned b_base[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int idx = tid + i * 512;
        const int row = idx / CPR;
        int ch = idx - row * CPR;
        ch = (((ch >> 1) ^ (row & 7)) << 1) | (ch & 1);
        const int j = j0 + ch * 8;
        const int tap = j0 / p.C, c = j - tap * p.C;            // C is a multiple of 256: one tap per tile
        const int fr = tap / p.S, fs = tap - fr * p.S;
        const bool live = j < RSC;
        b_dh[i] = fr - p.pad;
        b_dw[i] = fs - p.pad_w;
        const int m = m_begin + row;
        if (SIMPLE) {
            b_base[i] = live ? ((unsigned)row * (unsigned)p.xs_w + (unsigned)c) * 2u : INVALID;
            b_wo[i] = 0; b_ho[i] = 0;
        } else {
            const int b = m / hw, rem = m - b * hw;
            const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
            b_wo[i] = wo;
            b_ho[i] = live ? ho : -(1 << 24);
            b_base[i] = (unsigned)(((b - b_first) * (int)p.xs_b + (ho * p.stride + fr - p.pad) * (int)p.xs_h + c) * 2);
        }
    }
    const int step_w = BK % p.Wout, step_h = BK / p.Wout;
    const int xs_w2 = (int)p.xs_w * 2;
    const int row_bytes = p.stride * (int)p.xs_h * 2;
    const int img_bytes = (int)p.xs_b * 2 - p.Hout * row_bytes;
    typedef __attribute__((address_space(3))) void* lds_void_ptr;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    // every tile issues exactly 4 LDS-DMA instructions per thread (rows past the end read as zeros): the
    // counted waits below rely on it.  piece 0, 1: dy rows; piece 2, 3: x rows.  Issuing one costs the wave
    // 60-180 cycles (MI355X_MICROARCH.md, "LDS-DMA piece"), so the pieces go between groups of MFMAs.
    auto fetch_piece = [&](int kt, int stage, int piece) {
        const int rows_left = m_end - (m_begin + kt * BK);
        const unsigned Ad = lds0 + (unsigned)(stage * STAGE + wid * 1024), Bd = Ad + HALF;
        const int i = piece & 1;
        const int row = (tid + i * 512) / CPR;
        if (piece < 2) {
            const int a_soff = kt * (BK * 2) * p.K;
            const int vo = (int)(row < rows_left ? a_voff[i] : INVALID);
            lds_dma16(dyr, Ad + i * 8192, (unsigned)vo, (unsigned)a_soff);
        } else if (SIMPLE) {
            const int b_soff = kt * BK * xs_w2;
            const int vo = (int)(row < rows_left ? b_base[i] : INVALID);
            lds_dma16(xr, Bd + i * 8192, (unsigned)vo, (unsigned)b_soff);
        } else {
            const int hi = b_ho[i] * p.stride + b_dh[i], wi = b_wo[i] * p.stride + b_dw[i];
            const bool ok = (row < rows_left) & (hi >= 0) & (hi < p.Hin) & (wi >= 0) & (wi < p.Win);
            const unsigned vo = ok ? b_base[i] + (unsigned)(wi * xs_w2) : INVALID;
            lds_dma16(xr, Bd + i * 8192, vo, 0u);
            int wo = b_wo[i] + step_w, ho = b_ho[i] + step_h;
            unsigned base = b_base[i] + (unsigned)(step_h * row_bytes);
            if (wo >= p.Wout) { wo -= p.Wout; ++ho; base += (unsigned)row_bytes; }
            while (ho >= p.Hout) { ho -= p.Hout; base += (unsigned)img_bytes; }
            b_wo[i] = wo; b_ho[i] = ho; b_base[i] = base;
        }
    };
    auto fetch = [&](int kt, int stage) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) fetch_piece(kt, stage, pc);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 cacc[COLSUM ? FN : 1];
    bf16x8 ones;
    if (COLSUM) {
#pragma unroll
        for (int j = 0; j < FN; ++j) cacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = f2bf(1.f);
    }

    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int tr_row = 4 * g + q, swz = tr_row & 7;
    // byte offsets of this lane's fragment reads inside a stage (row tr_row; the second half of a fragment is 16 rows down)
    int a_off[FM], b_off[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a_off[i] = tr_row * ROWB + (((wm * 8 + i) ^ swz) * 32) + 8 * pq;
#pragma unroll
    for (int j = 0; j < FN; ++j) b_off[j] = HALF + tr_row * ROWB + (((wn * 4 + j) ^ swz) * 32) + 8 * pq;

    auto wait_tiles = [&](int tiles_in_flight) {   // counted: 4 LDS-DMA instructions per tile and thread
        if (tiles_in_flight >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (tiles_in_flight == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto read_a = [&](const unsigned char* St, int i) {
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + a_off[i]));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + a_off[i] + 16 * ROWB));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_b = [&](const unsigned char* St, int j) {
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + b_off[j]));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + b_off[j] + 16 * ROWB));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    if (PIPE) {
        // Fragments of tile t+1 are read WHILE tile t is multiplied: the x fragments into a second register
        // set, each dy fragment into its own registers right after its last MFMA of tile t.  Stage of tile t
        // is free from the barrier that ends iteration t-1 on: the DMA of tile t+NST goes there.
#pragma unroll
        for (int s0 = 0; s0 < NST; ++s0) fetch(s0, s0);
        wait_tiles(NST - 1);
        __builtin_amdgcn_s_barrier();
        bf16x8 af[FM], b0[FN], b1[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) b0[j] = read_b(smem, j);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = read_a(smem, i);
        wait_tiles(NST - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;                               // stage of tile kt
        auto body = [&](int kt, bf16x8 (&bc)[FN], bf16x8 (&bn)[FN]) {
            const int ns = cur + 1 == NST ? 0 : cur + 1;
            const unsigned char* Sn = smem + ns * STAGE;
#pragma unroll
            for (int j = 0; j < FN; ++j) bn[j] = read_b(Sn, j);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bc[j], acc[i][j], 0, 0, 0);
                af[i] = read_a(Sn, i);
                if (i & 1) {                      // one DMA piece behind every 8 MFMAs
                    fetch_piece(kt + NST, cur, i >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wait_tiles(NST - 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = ns;
        };
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
            body(kt, b0, b1);
            body(kt + 1, b1, b0);
        }
        if (kt < KT) body(kt, b0, b1);
    } else {
    // prologue: NST-1 tiles in flight, the first one landed
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) fetch(s0, s0);
    wait_tiles(NST - 2);
    __builtin_amdgcn_s_barrier();

    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < KT; ++kt) {
        fetch(kt + NST - 1, nxt);              // into the stage whose last reads retired before the previous barrier
        unsigned char* St = smem + cur * STAGE;
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = read_b(St, j);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = read_a(St, i);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (COLSUM && wm == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j) cacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[j], cacc[j], 0, 0, 0);
        }
        // tile kt+1 has landed (this wave's part; the barrier makes it everyone's); every LDS read of this
        // tile has retired, so the next iteration may overwrite the stage
        wait_tiles(NST - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        nxt = cur;
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-filled tiles past the end

    float* out = p.partial + (long long)split * p.K * RSC;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int jc = j0 + wn * 64 + j * 16 + fr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wm * 128 + i * 16 + fq * 4 + e;
                if (n < p.K && jc < RSC) out[(long long)n * RSC + jc] = acc[i][j][e];
            }
        }
    if (COLSUM && wm == 0 && tile_k == 0 && fq == 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int jc = j0 + wn * 64 + j * 16 + fr;
            if (jc < RSC) {
                p.colsum[((long long)split * 2) * RSC + jc] = cacc[j][0];
                p.colsum[((long long)split * 2 + 1) * RSC + jc] = cacc[j][0];
            }
        }
    }
}

// ---- 3x3 / stride 1 / pad 1 layers with 64 or 128 channels: row-streaming kernel ------------------
// The 64- and 128-channel 3x3 weight gradients ran at 490-690 TFLOP/s in the tile kernel: one tap and 64 x 64
// outputs per workgroup means 4 MFMAs per wave between two DMA round trips, and every tap re-fetches dy and x.
// Here a workgroup (8 waves) owns ALL nine taps of 64 output channels: the whole [64][9*C] partial lives in its
// accumulators (72 / 144 registers per lane) and it streams its images row by row.  A step is RS output rows
// (RS * W = 56 pixels, zero-padded to 64 = two MFMA reduction steps); its LDS stage holds the RS + 2 input rows
// the nine taps read, each with a ZERO halo pixel on both sides and zero rows above / below the image (written by
// the DMA's range check: no masks, no bounds arithmetic in the loop), plus the dy rows.  Tap (r, s) of pixel
// (row rr, column w) is LDS row rr + r, pixel w + s: 36 / 72 MFMAs per wave and step against 5 / 6 DMA pieces.
// Stages form the same ring as above (DMA of step t+2 in flight while step t is multiplied; counted vmcnt).
// Wave grid 2 (halves of the 64 output channels) x 4 (16-channel groups of C): every wave sees all nine taps,
// so tap offsets are compile-time immediates.  Pad pixels (56..63) have dy = 0 (range check) and read a clamped,
// finite x pixel.  Swizzle: 32-byte window w of pixel P at window w ^ f(P), f = (P >> 1) & 3 for 128-byte
// pixels, P & 7 for 256-byte pixels (as in the tile kernels), applied to the DMA source and to the reads.
// DYF: the BatchNorm-backward apply of the layer behind this conv (dy = ca*g + cb*y + cc) happens on the landed dy
// tile, like XF on the x rows: the tile of y rides next to the tile of g and the materialised dy never exists (the
// backward-data kernel of the same conv does the same, conv_rows.hip).
template <int C, int RS, bool XF = false, bool DYF = false>
__global__ void __launch_bounds__(512) conv_wgrad_rows_kernel(const WgradParams p, int imgs_per_split, int B) {
    constexpr int XB = C * 2, XW = XB / 32, CF = C / 16, CL = CF / 4;
    constexpr int NST = 3, XROWS = RS + 2, DYOFF = XROWS * 8192, STAGE = DYOFF + 8192 + (DYF ? 8192 : 0), NP = XROWS + 1 + (DYF ? 1 : 0);
    static_assert(C == 64 || C == 128, "16-channel groups over four waves");
    static_assert(!(XF || DYF) || (C == 64 && RS == 1), "input transforms: the 64-channel, 56-pixel-row layers");
    constexpr int FBASE = NST * STAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE + (DYF ? 1280 : XF ? 512 : 0)];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid >> 2, wc = wid & 3;
    const int khalves = p.K / 64;
    const int nwg = p.splits * khalves;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / khalves, kh = bid - split * khalves;
    const int H = p.Hout, W = p.Wout, RSC = 9 * C;
    const int b_begin = split * imgs_per_split;
    int b_end = b_begin + imgs_per_split;
    if (b_end > B) b_end = B;
    const int SPI = H / RS;                       // steps per image
    const int T = (b_end - b_begin) * SPI;

    const u32x4 dyr = raw_rsrc(p.dy, (unsigned)p.dy_bytes), xr = raw_rsrc(p.x, (unsigned)p.x_bytes);
    const u32x4 dy2r = raw_rsrc(DYF ? p.dy_y : p.dy, (unsigned)p.dy_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    constexpr unsigned INVALID = 0x80000000u;

    // this lane's 16 bytes of its wave's 1 KiB DMA piece: an x row slot is 8 KiB (8192 / XB pixels, pixel 0 = the
    // left halo), the dy tile 64 pixels x 128 bytes (this workgroup's 64 output channels)
    unsigned x_lane, dy_lane;
    {
        const int byte = wid * 1024 + lane * 16;
        const int P = byte / XB, chp = (byte % XB) >> 4;
        const int fx = XB == 128 ? ((P >> 1) & 3) : (P & 7);
        const int lch = ((((chp >> 1) ^ fx)) << 1) | (chp & 1);
        const int w = P - 1;
        x_lane = (w >= 0 && w < W) ? (unsigned)((w * C + lch * 8) * 2) : INVALID;
        const int m = byte >> 7, dchp = (byte & 127) >> 4;
        const int dlch = ((((dchp >> 1) ^ ((m >> 1) & 3))) << 1) | (dchp & 1);
        dy_lane = (m < RS * W) ? (unsigned)((m * p.K + dlch * 8) * 2) : INVALID;
    }
    // every step is exactly NP DMA instructions per wave (steps past the end and rows outside the image read zeros)
    auto fetch = [&](int t, int stage) {
        const bool live = t < T;
        const int tt = live ? t : 0;
        const int b = b_begin + tt / SPI, h0 = (tt % SPI) * RS;
        const unsigned sb = lds0 + (unsigned)(stage * STAGE + wid * 1024);
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int hin = h0 - 1 + i;
            const bool ok = live && hin >= 0 && hin < H;     // uniform
            const unsigned soff = ok ? (unsigned)(((b * H + hin) * W) * C) * 2u : 0u;
            lds_dma16(xr, sb + i * 8192, ok ? x_lane : INVALID, soff);
        }
        const unsigned dsoff = (unsigned)(((b * H + h0) * W) * p.K + kh * 64) * 2u;
        lds_dma16(dyr, sb + DYOFF, live ? dy_lane : INVALID, dsoff);
        if (DYF) lds_dma16(dy2r, sb + DYOFF + 8192, live ? dy_lane : INVALID, dsoff);
    };
    auto wait_steps = [&](int steps_in_flight) {
        if (steps_in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP) : "memory");
        else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // transposing-read lane geometry: lane 16g + 4q + pq supplies reduction row 4g + q (+16), bytes 8pq..8pq+7 of a window
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int tr_row = 4 * g + q;
    // dy fragments (rows = pixels of the step, 128-byte rows): [reduction step][16-channel group of this wave's half]
    int a_off[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = ks * 32 + tr_row;
            a_off[ks][i] = DYOFF + row * 128 + (((wn * 2 + i) ^ ((row >> 1) & 3)) << 5) + 8 * pq;   // second half: + 16 rows (same swizzle)
        }
    // x fragments: [reduction step][half][s][channel group]: pixel w + s of row rr (tap row r adds r * 8192 as an immediate)
    int x_off[2][2][3][CL];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            int m = ks * 32 + tr_row + 16 * hf;
            if (m > RS * W - 1) m = RS * W - 1;          // pad pixels: dy is zero there, x only has to be finite
            const int rr = m / W, w = m - rr * W;
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const int P = w + s_;
                const int fx = XB == 128 ? ((P >> 1) & 3) : (P & 7);
#pragma unroll
                for (int cl = 0; cl < CL; ++cl)
                    x_off[ks][hf][s_][cl] = rr * 8192 + P * XB + ((((wc * CL + cl) ^ fx)) << 5) + 8 * pq;
            }
        }

    f32x4 acc[2][9][CL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int cl = 0; cl < CL; ++cl) acc[i][t9][cl] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments of half a step (32 pixels, all nine taps) live in registers, so that reading them and multiplying
    // them can be separate PHASES
    bf16x8 af[2], bfr[9][CL];
    auto read_frags = [&](const unsigned char* St, int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ao = ks ? a_off[1][i] : a_off[0][i];
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + ao));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + ao + 16 * 128));
            af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const int r_ = t9 / 3, s_ = t9 - 3 * r_;
#pragma unroll
            for (int cl = 0; cl < CL; ++cl) {
                const int o0 = ks ? x_off[1][0][s_][cl] : x_off[0][0][s_][cl], o1 = ks ? x_off[1][1][s_][cl] : x_off[0][1][s_][cl];
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + o0 + r_ * 8192));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(St + o1 + r_ * 8192));
                bfr[t9][cl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto mfma_all = [&]() {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int cl = 0; cl < CL; ++cl)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][t9][cl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[t9][cl], acc[i][t9][cl], 0, 0, 0);
    };

    // XF: scale * x + shift, ReLU on the x rows of a landed stage, in place, before anybody reads fragments from it
    // (one more barrier per step; halo pixels and rows outside the image stay zero).  A thread owns pixel tid / 8,
    // physical 16-byte chunk tid % 8 of each of the three rows.
    const int tP = tid >> 3, tchp = tid & 7;
    const int tlch = ((((tchp >> 1) ^ ((tP >> 1) & 3))) << 1) | (tchp & 1);          // logical chunk: channels 8 tlch ..
    const bool tvalid = tP >= 1 && tP <= W;
    if (XF && tid < 128) reinterpret_cast<float*>(smem + FBASE)[tid] = p.x_fcoef[tid];
    if (DYF && tid < 192) reinterpret_cast<float*>(smem + FBASE + 512)[tid] = p.dy_coef[tid];
    // DYF: the same thread owns pixel tid / 8, physical chunk tid % 8 of the dy tile (its swizzle: dy_lane above)
    const int dlch = ((((tchp >> 1) ^ ((tP >> 1) & 3))) << 1) | (tchp & 1);
    auto transform = [&](int t, int stage) {
        if (t >= T) return;
        if (XF) {
            const int h0 = (t % SPI) * RS;
            float sc[8], sh[8];
            load8f(reinterpret_cast<const float*>(smem + FBASE) + 8 * tlch, sc);
            load8f(reinterpret_cast<const float*>(smem + FBASE + 256) + 8 * tlch, sh);
#pragma unroll
            for (int i = 0; i < XROWS; ++i) {
                const int hin = h0 - 1 + i;
                if (hin < 0 || hin >= H || !tvalid) continue;
                unsigned char* a = smem + stage * STAGE + i * 8192 + tid * 16;
                bf16x8 d = *reinterpret_cast<const bf16x8*>(a);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bn_affine(bf2f(d[e]), sc[e], sh[e]);
                    d[e] = f2bf(f > 0.f ? f : 0.f);
                }
                *reinterpret_cast<bf16x8*>(a) = d;
            }
        }
        if (DYF && tP < RS * W) {                   // pad pixels keep their zero gradient
            unsigned char* a = smem + stage * STAGE + DYOFF + tid * 16;
            bf16x8 d = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 y2 = *reinterpret_cast<const bf16x8*>(a + 8192);
            float ca[8], cb[8], cc[8];
            load8f(reinterpret_cast<const float*>(smem + FBASE + 512) + 8 * dlch, ca);
            load8f(reinterpret_cast<const float*>(smem + FBASE + 768) + 8 * dlch, cb);
            load8f(reinterpret_cast<const float*>(smem + FBASE + 1024) + 8 * dlch, cc);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = f2bf(bn_bwd_affine(bf2f(d[e]), bf2f(y2[e]), ca[e], cb[e], cc[e]));
            *reinterpret_cast<bf16x8*>(a) = d;
        }
    };

    fetch(0, 0);
    fetch(1, 1);
    wait_steps(1);
    if (XF || DYF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the coefficient stores)
    __builtin_amdgcn_s_barrier();
    if (XF || DYF) {
        transform(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    int cur = 0, nxt = 2;
    // (Measured and dropped, profiles/r3_notes.md: a two-phase schedule with the two waves of a SIMD one phase
    // apart - one reads fragments while the other multiplies - 75 -> 95 us; the second wave group issuing its DMA
    // pieces between the two reduction halves instead of at the top of the step, 75.6 -> 78.5 us.)
    for (int t = 0; t < T; ++t) {
        fetch(t + 2, nxt);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            read_frags(smem + cur * STAGE, ks);
            mfma_all();
        }
        wait_steps(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        nxt = cur;
        cur = cur + 1 == NST ? 0 : cur + 1;
        if (XF || DYF) {                        // the stage of step t+1 has landed everywhere
            transform(t + 1, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // D rows = output channel (4 per lane), cols = (tap, channel) (lane & 15)
    float* out = p.partial + (long long)split * p.K * RSC;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int cl = 0; cl < CL; ++cl) {
                const int jc = t9 * C + (wc * CL + cl) * 16 + fr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = kh * 64 + wn * 32 + i * 16 + fq * 4 + e;
                    out[(long long)n * RSC + jc] = acc[i][t9][cl][e];
                }
            }
}

// out[i] = sum_s partial[s][i] (fixed order => deterministic), bf16 or fp32 output.
// Block = (256 / SL) float4 columns x SL split lanes, so that tall-and-narrow reductions (a 64x64
// 1x1 conv weight summed over 512 slabs) still spread over many CUs with short dependent chains.
template <typename OutT, bool ACCUM, int SL>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial, OutT* __restrict__ out,
                                                           long long n, int splits, long long slab) {
    constexpr int COLS = 256 / SL;
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const long long n4 = n >> 2;
    const long long i = (long long)blockIdx.x * COLS + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int k = sl;
        for (; k + 3 * SL < splits; k += 4 * SL) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + (long long)k * slab + (i << 2));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + SL) * slab + (i << 2));
            f32x4 v2 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + 2 * SL) * slab + (i << 2));
            f32x4 v3 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + 3 * SL) * slab + (i << 2));
            s += (v0 + v1) + (v2 + v3);
        }
        for (; k < splits; k += SL) s += *reinterpret_cast<const f32x4*>(partial + (long long)k * slab + (i << 2));
    }
    if (SL > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl != 0) return;
        for (int l = 1; l < SL; ++l) s += red[l * COLS + col];
    }
    if (i >= n4) return;
    if constexpr (sizeof(OutT) == 4) {
        f32x4* o = reinterpret_cast<f32x4*>(out + (i << 2));
        if (ACCUM) s += *o;
        *o = s;
    } else {
        bf16x4* o = reinterpret_cast<bf16x4*>(out + (i << 2));
        if (ACCUM) { bf16x4 old = *o; for (int e = 0; e < 4; ++e) s[e] += bf2f(old[e]); }
        bf16x4 v;
        for (int e = 0; e < 4; ++e) v[e] = f2bf(s[e]);
        *o = v;
    }
}

// BatchNorm backward by linearity (the BatchNorm that follows a 1x1 conv, dy = ca*g + cb*y + cc with
// y = z W^T): the weight gradient dy^T z from the RAW product g^T z (the slabs), the Gram term
// T = W (z^T z) and the column sums of z,
//   out[c][j] (+)= ca[c] * sum_s partial[s][c][j] + cb[c] * T[c][j] + cc[c] * colz[j]
// Same block shape and the same fixed summation order as splitk_reduce_kernel.
// CENTER = true is the companion that produces the Gram matrix itself: out[k][j] = sum_s partial -
// colz[k]*colz[j]/M, the CENTRED second moment (subtracted in fp32, before the bf16 rounding: with
// the raw Gram matrix the product W G and the mean term cc (x) colz cancel each other's large parts
// after rounding, which costs mean^2/variance in relative accuracy).  With the centred matrix the
// rank-1 coefficient becomes cc + cb*mean = -ca*sum(g)/M.
template <typename OutT, bool ACCUM, int SL, bool CENTER>
__global__ void __launch_bounds__(256) splitk_reduce_lin_kernel(const float* __restrict__ partial, OutT* __restrict__ out,
                                                               long long n, int splits, long long slab,
                                                               const float* __restrict__ coef, int C, const bf16_t* __restrict__ T,
                                                               const float* __restrict__ colz, int row_len,
                                                               const float* __restrict__ mean, float inv_m) {
    constexpr int COLS = 256 / SL;
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const long long n4 = n >> 2;
    const long long i = (long long)blockIdx.x * COLS + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int k = sl;
        for (; k + 3 * SL < splits; k += 4 * SL) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + (long long)k * slab + (i << 2));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + SL) * slab + (i << 2));
            f32x4 v2 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + 2 * SL) * slab + (i << 2));
            f32x4 v3 = *reinterpret_cast<const f32x4*>(partial + (long long)(k + 3 * SL) * slab + (i << 2));
            s += (v0 + v1) + (v2 + v3);
        }
        for (; k < splits; k += SL) s += *reinterpret_cast<const f32x4*>(partial + (long long)k * slab + (i << 2));
    }
    if (SL > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl != 0) return;
        for (int l = 1; l < SL; ++l) s += red[l * COLS + col];
    }
    if (i >= n4) return;
    const long long e0 = i << 2;
    const int c = (int)(e0 / row_len), j = (int)(e0 - (long long)c * row_len);
    const f32x4 cz = *reinterpret_cast<const f32x4*>(colz + j);
    if (CENTER) {
        const float zc = colz[c] * inv_m;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] -= zc * cz[e];
    } else {
        const float ca = coef[c], cb = coef[C + c], cc = coef[2 * C + c] + cb * mean[c];
        const bf16x4 tv = *reinterpret_cast<const bf16x4*>(T + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = ca * s[e] + cb * bf2f(tv[e]) + cc * cz[e];
    }
    if constexpr (sizeof(OutT) == 4) {
        f32x4* o = reinterpret_cast<f32x4*>(out + e0);
        if (ACCUM) s += *o;
        *o = s;
    } else {
        bf16x4* o = reinterpret_cast<bf16x4*>(out + e0);
        if (ACCUM) { bf16x4 old = *o; for (int e = 0; e < 4; ++e) s[e] += bf2f(old[e]); }
        bf16x4 v;
        for (int e = 0; e < 4; ++e) v[e] = f2bf(s[e]);
        *o = v;
    }
}

template <int BMw, int BNw>
int launch_wgrad(WgradParams& p, hipStream_t stream) {
    p.tiles_k = (p.K + BMw - 1) / BMw;
    p.tiles_rsc = (p.R * p.S * p.C + BNw - 1) / BNw;
    dim3 grid(p.tiles_k * p.tiles_rsc * p.splits), block(256);
    // Operand tiles by LDS-DMA (tile widths 64 / 128 only), per class (kernel_bench.py sweeps,
    // profiles/r1_notes.md): 3x3 filters with 32 reduction rows per tile (a3 130 -> 120 us, c4 90 -> 85);
    // 1x1 GEMMs with >= 128 channels on both sides with 64 rows per tile (half the barriers: c1
    // 100 -> 84 us, d1 70 -> 63, b5 57 -> 51; with 32 rows LDS-DMA loses 5-15 % there, with 64 rows the
    // 3x3 filters lose 10-15 % to the 64 KB of LDS).  The bandwidth-bound 64-channel layers keep the
    // register-staged loop.  IN THE TRAINING STEP the 1x1 GEMMs stay register-staged as well: their
    // LDS-DMA form wins in isolation but its 64 KB per workgroup (two resident per CU) leaves no room
    // for the dependent chain's conv workgroups next to it - 18.29 / 18.29 ms per step with it, 18.21 /
    // 18.19 without (profiles/r2_notes.md section 22).  DDL_WGRAD_GLDS=0|1|2 (2: the isolated optimum)
    // and DDL_WGRAD_BK=32|64 force.
    static const int glds_env = getenv("DDL_WGRAD_GLDS") ? atoi(getenv("DDL_WGRAD_GLDS")) : -1;  // tuning knob
    static const int bk_env = getenv("DDL_WGRAD_BK") ? atoi(getenv("DDL_WGRAD_BK")) : 0;         // tuning knob
    constexpr bool CAN = (BMw == 64 || BMw == 128) && (BNw == 64 || BNw == 128);
    if constexpr (CAN) {
        const bool one = p.R * p.S == 1;
        const bool glds = glds_env == 2 ? (!one || (p.K >= 128 && p.C >= 128)) : (glds_env > 0 || (glds_env < 0 && !one));   // 2: also the wide 1x1 GEMMs (the isolated optimum)
        const bool bk64 = glds && (bk_env == 64 || (bk_env == 0 && one));
        if (bk64) {
            if (p.simple) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, true, true, true, 64>), grid, block, 0, stream, p);
            else if (p.C % BNw == 0) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, true, true, 64>), grid, block, 0, stream, p);
            else DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, false, true, 64>), grid, block, 0, stream, p);
            return ddl_launch_status();
        }
        if (glds) {
            if (p.simple) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, true, true, true>), grid, block, 0, stream, p);
            else if (p.C % BNw == 0) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, true, true>), grid, block, 0, stream, p);
            else DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, false, true>), grid, block, 0, stream, p);
            return ddl_launch_status();
        }
    }
    if (p.colsum) {   // (the caller checked: plain GEMM, register-staged, a tile shape that has the instantiation)
        if constexpr ((BMw == 64 && BNw == 64) || (BMw == 128 && BNw == 128)) {
            DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, true, true, false, 32, true>), grid, block, 0, stream, p);
            return ddl_launch_status();
        }
        return DDL_EINVAL;
    }
    if (p.simple) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, true, true, false>), grid, block, 0, stream, p);
    else if (p.C % BNw == 0) DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, true, false>), grid, block, 0, stream, p);
    else DDL_KLAUNCH((conv_wgrad_kernel<BMw, BNw, false, false, false>), grid, block, 0, stream, p);
    return ddl_launch_status();
}

}  // namespace

// Split planning shared with the host wrapper: returns the number of reduction splits for a
// problem so the caller can size the partial workspace (splits * K * R*S*C floats).
// Tile shape per problem: rows = output channels, columns = flattened (r, s, c).
static void wgrad_tile(int K, int C, int R, int S, int& bm, int& bn) {
    const int RSC = R * S * C;
    bm = K <= 64 ? 64 : 128;
    if (K <= 64) {
        if (RSC <= 32) bn = 32;
        else if (RSC <= 64 || C == 64) bn = 64;   // 64-channel 3x3: one tap per tile measured faster than 3 (192)
        else bn = 128;                            // incl. the stem: 4 of its 32-wide taps per tile
    } else {
        bn = RSC <= 64 ? 64 : 128;
    }
}

// The 256 x 256 / 8-wave kernel serves problems whose output is made of whole 256 x 256 tiles with one
// filter tap per tile.  DDL_WGRAD_BIG=0|1 (default below), DDL_WGRAD_BIG_TARGET = workgroups to aim for
// (one per CU fits: 96-128 KB of LDS each), DDL_WGRAD_BIG_NST = LDS stages (3 or 4).
static bool wgrad_big(long long M, int K, int C, int R, int S) {
    static const int on = getenv("DDL_WGRAD_BIG") ? atoi(getenv("DDL_WGRAD_BIG")) : 1;  // tuning knob
    return on && K % 256 == 0 && C % 256 == 0 && M >= 4096;
}
static int wgrad_big_splits(long long M, int K, int C, int R, int S) {
    // In the training step fewer, longer workgroups win (each costs 256 KB of partial-sum traffic and one CU's
    // LDS): 64 / 96 / 128 / 160 / 192 / 256 workgroups -> 18.13 / 17.85 / 17.80 / 17.70 / 17.73 / 18.01 ms per step
    // against 18.18-18.24 ms with the 128 x 128 kernel (profiles/r3_notes.md).  Swept again on the final tree of the
    // round (row-streaming kernels next to it): 96 / 112 / 128 / 144 / 160 -> 16.79 / 16.78 / 16.65 / 16.81 / 16.69 ms,
    // ResNet-101 28.24 -> 28.08, ResNet-152 40.72 -> 40.54 for 160 -> 128 (not monotonic: the count is rounded per layer)
    static const int target = getenv("DDL_WGRAD_BIG_TARGET") ? atoi(getenv("DDL_WGRAD_BIG_TARGET")) : 128;  // tuning knob
    const long long tiles = (long long)(K / 256) * ((long long)R * S * C / 256);
    long long want = (target + tiles / 2) / tiles;
    const long long max_by_m = M / 512;                    // at least 16 tiles of 32 rows per split
    if (want > max_by_m) want = max_by_m;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    return (int)want;
}

// The row-streaming kernel: 3x3 / stride 1 / pad 1 on 56- or 28-pixel rows with 64 or 128 input channels.
// DDL_WGRAD_ROWS=0|1; DDL_WGRAD_ROWS_TARGET = workgroups to aim for (images are dealt out whole).
static bool wgrad_rows(int K, int C, int R, int S, int Hout, int Wout, int stride, int B) {
    static const int on = getenv("DDL_WGRAD_ROWS") ? atoi(getenv("DDL_WGRAD_ROWS")) : 1;  // tuning knob
    if (!on || R != 3 || S != 3 || stride != 1 || B <= 0) return false;
    if (!(C == 64 || C == 128) || K % 64 != 0) return false;
    if (!(Wout == 56 || Wout == 28)) return false;
    const int rs = 56 / Wout;
    return Hout % rs == 0;
}
static int wgrad_rows_ips(int K, int B) {
    // (re-swept on the final tree of round 3: 256 / 192 / 128 / 64 workgroups -> 16.30 / 16.27 / 16.23 / 16.24 ms per step,
    // ResNet-34 9.60 -> 9.44, ResNet-18 5.47 -> 5.40 for 256 -> 128: two images per workgroup leave half the CUs to the chain)
    static const int target = getenv("DDL_WGRAD_ROWS_TARGET") ? atoi(getenv("DDL_WGRAD_ROWS_TARGET")) : 128;  // tuning knob
    int nsplit = target / (K / 64);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > B) nsplit = B;
    return (B + nsplit - 1) / nsplit;
}

extern "C" int ddl_conv_wgrad_splits(long long M, int K, int C, int R, int S);
// `Hout`, `Wout`, `stride`, `B` (0 when the caller does not know them: a plain GEMM) select the row-streaming kernel
extern "C" int ddl_conv_wgrad_splits_ex(long long M, int K, int C, int R, int S, int Hout, int Wout, int stride, int B) {
    if (wgrad_rows(K, C, R, S, Hout, Wout, stride, B)) {
        const int ips = wgrad_rows_ips(K, B);
        return (B + ips - 1) / ips;
    }
    return ddl_conv_wgrad_splits(M, K, C, R, S);
}

extern "C" int ddl_conv_wgrad_splits(long long M, int K, int C, int R, int S) {
    if (wgrad_big(M, K, C, R, S)) return wgrad_big_splits(M, K, C, R, S);
    int bm, bn;
    wgrad_tile(K, C, R, S, bm, bn);
    long long tiles = (long long)((K + bm - 1) / bm) * ((R * S * C + bn - 1) / bn);
    // Workgroups to aim for, from the DDL_WGRAD_TARGET sweep of benchmarks/kernel_bench.py on
    // MI355X (profiles/r1_notes.md): every workgroup costs one fp32 tile of partial-sum traffic
    // (64 KB for 128x128), so the 1x1 GEMMs want just 2 per CU; 3x3 taps re-read their operands
    // through L2 and want more parallelism, the 64-channel 3x3 (tiny tiles) the most.
    static const int forced = getenv("DDL_WGRAD_TARGET") ? atoi(getenv("DDL_WGRAD_TARGET")) : 0;  // tuning knob
    int target;
    if (R * S == 1) target = (K <= 64 || C <= 64) ? 768 : 512;
    else if (R * S >= 9) target = (C <= 64) ? 3072 : 896;
    else target = 1024;
    if (forced > 0) target = forced;
    static const int scale_pct = getenv("DDL_WGRAD_TARGET_PCT") ? atoi(getenv("DDL_WGRAD_TARGET_PCT")) : 100;  // tuning knob
    if (scale_pct > 0 && scale_pct != 100) target = (int)((long long)target * scale_pct / 100);
    long long want = (target + tiles - 1) / tiles;
    long long max_by_m = (M + 255) / 256;                  // at least 256 reduction rows per split
    if (want > max_by_m) want = max_by_m;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    return (int)want;
}

// plain-GEMM weight gradients that can emit the column sums of x on the side
extern "C" int ddl_conv_wgrad_colsum_ok(long long M, int K, int C) {
    static const int on = getenv("DDL_WGRAD_COLSUM") ? atoi(getenv("DDL_WGRAD_COLSUM")) : 1;  // tuning knob
    if (!on) return 0;
    if (wgrad_big(M, K, C, 1, 1)) return 1;
    static const int glds_env = getenv("DDL_WGRAD_GLDS") ? atoi(getenv("DDL_WGRAD_GLDS")) : -1;
    if (glds_env > 0) return 0;            // (the LDS-DMA forms have no such instantiation)
    int bm, bn;
    wgrad_tile(K, C, 1, 1, bm, bn);
    return (bm == 64 && bn == 64) || (bm == 128 && bn == 128);
}

// `colsum` (optional; plain GEMMs whose tile shape has the instantiation - ddl_conv_wgrad_colsum_ok): [splits][2][C]
// fp32 column sums of x per split (a statistics-table layout, both rows alike)
// the row-streaming kernel will serve this weight gradient (what a caller that relies on `x_fcoef` has to know)
extern "C" int ddl_conv_wgrad_rows_ok(int K, int C, int R, int S, int Hout, int Wout, int stride, int B) {
    return wgrad_rows(K, C, R, S, Hout, Wout, stride, B) ? 1 : 0;
}

// `x_fcoef` (optional, [2][C] fp32 scale | shift; only where the row-streaming kernel with C = 64 on 56-pixel rows
// serves the launch - anything else is refused): x is the raw input of a BatchNorm + ReLU, applied on the staged rows
extern "C" int ddl_conv_wgrad_ex2(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                                  int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                  int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                                  void* colsum, const void* x_fcoef, hipStream_t stream);
// ... and `dy_y` + `dy_coef` ([3][K] fp32 ca | cb | cc; same kernel, K = 64): dy is the gradient with respect to a
// BatchNorm's output and dy_y that BatchNorm's input - the product uses ca*dy + cb*dy_y + cc (backward apply folded in)
extern "C" int ddl_conv_wgrad_ex3(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                                  int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                  int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                                  void* colsum, const void* x_fcoef, const void* dy_y, const void* dy_coef, hipStream_t stream);

extern "C" int ddl_conv_wgrad_ex(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                                 int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                 int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                                 void* colsum, hipStream_t stream) {
    return ddl_conv_wgrad_ex2(dy, x, partial, dy_numel, x_numel, B, Hin, Win, C, xs_b, xs_h, xs_w, Hout, Wout, K, R, S,
                              stride, pad, pad_w, splits, colsum, nullptr, stream);
}

extern "C" int ddl_conv_wgrad_ex2(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                                  int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                  int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                                  void* colsum, const void* x_fcoef, hipStream_t stream) {
    return ddl_conv_wgrad_ex3(dy, x, partial, dy_numel, x_numel, B, Hin, Win, C, xs_b, xs_h, xs_w, Hout, Wout, K, R, S,
                              stride, pad, pad_w, splits, colsum, x_fcoef, nullptr, nullptr, stream);
}

extern "C" int ddl_conv_wgrad_ex3(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                                  int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                                  int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                                  void* colsum, const void* x_fcoef, const void* dy_y, const void* dy_coef, hipStream_t stream) {
    if ((dy_y == nullptr) != (dy_coef == nullptr)) return DDL_EINVAL;
    if (dy_y && (!ddl_aligned16(dy_y) || !ddl_aligned16(dy_coef))) return DDL_EALIGN;
    if (!dy || !x || !partial || B <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || splits <= 0) return DDL_EINVAL;
    if (C % 8 != 0 || K % 8 != 0) return DDL_EINVAL;
    if (dy_numel <= 0 || x_numel <= 0) return DDL_EINVAL;
    if (!ddl_aligned16(dy) || !ddl_aligned16(x) || !ddl_aligned16(partial)) return DDL_EALIGN;
    if ((xs_b % 8) || (xs_h % 8)) return DDL_EALIGN;
    if (xs_w % 8) {
        if (S != 1 || ((long long)stride * xs_w) % 8 || ((long long)pad_w * xs_w) % 8) return DDL_EALIGN;
    }
    long long M = (long long)B * Hout * Wout;
    if (M >= (1LL << 31) - 4096 || dy_numel < M * K) return DDL_EINVAL;
    long long last = (long long)(B - 1) * xs_b + (long long)(Hin - 1) * xs_h + (long long)(Win - 1) * xs_w + C;
    if (last > x_numel) return DDL_EINVAL;
    WgradParams p;
    p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.partial = (float*)partial;
    p.dy_bytes = dy_numel * 2; p.x_bytes = x_numel * 2;
    p.Hin = Hin; p.Win = Win; p.C = C; p.xs_b = xs_b; p.xs_h = xs_h; p.xs_w = xs_w;
    p.Hout = Hout; p.Wout = Wout; p.K = K; p.M = (int)M;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.pad_w = pad_w;
    p.splits = splits;
    long long mps = (M + splits - 1) / splits;
    mps = (mps + 31) / 32 * 32;
    p.m_per_split = (int)mps;
    p.il_sg = p.il_mg = 0;
    p.colsum = (float*)colsum;
    p.x_fcoef = (const float*)x_fcoef;
    if (x_fcoef && !ddl_aligned16(x_fcoef)) return DDL_EALIGN;
    p.dy_y = (const bf16_t*)dy_y; p.dy_coef = (const float*)dy_coef;
    // 32-bit offsets inside one split: its dy rows, and the images its x rows come from
    if ((mps + 64) * K * 2 >= 0x7fffffffLL) return DDL_EINVAL;
    if ((mps / ((long long)Hout * Wout) + 3) * xs_b * 2 >= 0x7fffffffLL || (mps + 64) * xs_w * 2 >= 0x7fffffffLL) return DDL_EINVAL;
    p.simple = (R == 1 && S == 1 && stride == 1 && pad == 0 && pad_w == 0 && Hin == Hout && Win == Wout &&
                xs_h == (long long)Win * xs_w && xs_b == (long long)Hin * xs_h) ? 1 : 0;
    if (colsum && !(p.simple && ddl_conv_wgrad_colsum_ok(M, K, C))) return DDL_EINVAL;
    if (wgrad_rows(K, C, R, S, Hout, Wout, stride, B) && pad == 1 && pad_w == 1 && Hin == Hout && Win == Wout &&
        xs_w == C && xs_h == (long long)Win * C && xs_b == (long long)Hin * Win * C &&
        p.dy_bytes < 0x7fffffffLL && p.x_bytes < 0x7fffffffLL) {
        const int ips = wgrad_rows_ips(K, B);
        if (splits != (B + ips - 1) / ips) return DDL_EINVAL;   // the workspace was sized by ddl_conv_wgrad_splits_ex
        dim3 grid(splits * (K / 64)), block(512);
        if (p.x_fcoef || p.dy_y) {
            if (!(C == 64 && Wout == 56) || (p.dy_y && K != 64)) return DDL_EINVAL;
            if (p.x_fcoef && p.dy_y) DDL_KLAUNCH((conv_wgrad_rows_kernel<64, 1, true, true>), grid, block, 0, stream, p, ips, B);
            else if (p.dy_y) DDL_KLAUNCH((conv_wgrad_rows_kernel<64, 1, false, true>), grid, block, 0, stream, p, ips, B);
            else DDL_KLAUNCH((conv_wgrad_rows_kernel<64, 1, true>), grid, block, 0, stream, p, ips, B);
        }
        else if (C == 64 && Wout == 56) DDL_KLAUNCH((conv_wgrad_rows_kernel<64, 1>), grid, block, 0, stream, p, ips, B);
        else if (C == 64) DDL_KLAUNCH((conv_wgrad_rows_kernel<64, 2>), grid, block, 0, stream, p, ips, B);
        else if (Wout == 56) DDL_KLAUNCH((conv_wgrad_rows_kernel<128, 1>), grid, block, 0, stream, p, ips, B);
        else DDL_KLAUNCH((conv_wgrad_rows_kernel<128, 2>), grid, block, 0, stream, p, ips, B);
        return ddl_launch_status();
    }
    if (x_fcoef || dy_y) return DDL_EINVAL;    // only the row-streaming kernel transforms its inputs
    if (wgrad_big(M, K, C, R, S)) {
        static const int nst = getenv("DDL_WGRAD_BIG_NST") ? atoi(getenv("DDL_WGRAD_BIG_NST")) : 3;  // tuning knob
        p.tiles_k = K / 256;
        p.tiles_rsc = R * S * C / 256;
        dim3 grid(p.tiles_k * p.tiles_rsc * p.splits), block(512);
        // fragments of tile t+1 read under the MFMAs of tile t: +2-3 % alone; nothing in the step while the weight-gradient
        // stream had slack (17.70 / 17.73 vs 17.68 / 17.70), -0.12 ms now that both streams are critical (16.27 / 16.29 /
        // 16.24 -> 16.15 / 16.12 / 16.14, profiles/r3_notes.md section 31); same bits
        static const int pipe = getenv("DDL_WGRAD_BIG_PIPE") ? atoi(getenv("DDL_WGRAD_BIG_PIPE")) : 1;  // tuning knob
#define BIGL(SIMP, N, P) DDL_KLAUNCH((conv_wgrad256_kernel<SIMP, N, P>), grid, block, 0, stream, p)
        if (colsum) {
            DDL_KLAUNCH((conv_wgrad256_kernel<true, 3, false, true>), grid, block, 0, stream, p);
            return ddl_launch_status();
        }
        if (pipe) {
            if (nst == 4) { if (p.simple) BIGL(true, 4, true); else BIGL(false, 4, true); }
            else { if (p.simple) BIGL(true, 3, true); else BIGL(false, 3, true); }
        } else {
            if (nst == 4) { if (p.simple) BIGL(true, 4, false); else BIGL(false, 4, false); }
            else { if (p.simple) BIGL(true, 3, false); else BIGL(false, 3, false); }
        }
#undef BIGL
        return ddl_launch_status();
    }
    int bm, bn;
    wgrad_tile(K, C, R, S, bm, bn);
    {
        // DDL_WGRAD_INTERLEAVE = smallest reduction length (rows) that gets the interleaved order; 0 = never.
        // Off by default: with it the stage 1-2 weight gradients sweep their operands in the backward-data
        // kernel's order (hoping for Infinity Cache hits on the shared tensors) - 17.13 / 17.14 ms per step
        // against 17.10 / 17.12 without (profiles/r3_notes.md section 12)
        static const long long il_min = getenv("DDL_WGRAD_INTERLEAVE") ? atoll(getenv("DDL_WGRAD_INTERLEAVE")) : 0;  // tuning knob
        const long long mg = ((M + 7) / 8 + 63) / 64 * 64;
        if (il_min > 0 && p.simple && M >= il_min && splits % 8 == 0 && (mg + 128) * K * 2 < 0x7fffffffLL &&
            (mg + 128) * xs_w * 2 < 0x7fffffffLL) {
            p.il_sg = splits / 8;
            p.il_mg = (int)mg;
        }
    }
    if (bm == 64) {
        if (bn == 32) return launch_wgrad<64, 32>(p, stream);
        if (bn == 64) return launch_wgrad<64, 64>(p, stream);
        return launch_wgrad<64, 128>(p, stream);
    }
    if (bn == 64) return launch_wgrad<128, 64>(p, stream);
    return launch_wgrad<128, 128>(p, stream);
}

extern "C" int ddl_conv_wgrad(const void* dy, const void* x, void* partial, long long dy_numel, long long x_numel,
                              int B, int Hin, int Win, int C, long long xs_b, long long xs_h, long long xs_w,
                              int Hout, int Wout, int K, int R, int S, int stride, int pad, int pad_w, int splits,
                              hipStream_t stream) {
    return ddl_conv_wgrad_ex(dy, x, partial, dy_numel, x_numel, B, Hin, Win, C, xs_b, xs_h, xs_w, Hout, Wout, K, R, S,
                             stride, pad, pad_w, splits, nullptr, stream);
}

extern "C" int ddl_splitk_reduce(const void* partial, void* out, long long n, int splits, long long slab,
                                 int out_is_bf16, int accumulate, hipStream_t stream) {
    if (!partial || !out || n <= 0 || splits <= 0 || (n & 3) || slab < n || (slab & 3)) return DDL_EINVAL;
    if (!ddl_aligned16(partial) || (reinterpret_cast<uintptr_t>(out) & 7)) return DDL_EALIGN;
    if (!out_is_bf16 && !ddl_aligned16(out)) return DDL_EALIGN;
    const long long n4 = n / 4;
    const int sl = splits >= 64 ? 16 : (splits >= 8 ? 4 : 1);
    dim3 grid((unsigned)((n4 + 256 / sl - 1) / (256 / sl))), block(256);
#define RL3(T, A, S) DDL_KLAUNCH((splitk_reduce_kernel<T, A, S>), grid, block, 0, stream, (const float*)partial, (T*)out, n, splits, slab)
#define RL(T, A) do { if (sl == 16) RL3(T, A, 16); else if (sl == 4) RL3(T, A, 4); else RL3(T, A, 1); } while (0)
    if (out_is_bf16) { if (accumulate) RL(bf16_t, true); else RL(bf16_t, false); }
    else { if (accumulate) RL(float, true); else RL(float, false); }
#undef RL3
#undef RL
    return ddl_launch_status();
}

extern "C" int ddl_splitk_reduce_lin(const void* partial, void* out, long long n, int splits, long long slab,
                                     int out_is_bf16, int accumulate, const void* coef, int C, const void* T,
                                     const void* colz, int row_len, const void* mean, hipStream_t stream) {
    if (!partial || !out || !coef || !T || !colz || !mean || n <= 0 || splits <= 0 || (n & 3) || slab < n || (slab & 3)) return DDL_EINVAL;
    if (C <= 0 || row_len <= 0 || (row_len & 3) || n != (long long)C * row_len) return DDL_EINVAL;
    if (!ddl_aligned16(partial) || (reinterpret_cast<uintptr_t>(out) & 7) || (reinterpret_cast<uintptr_t>(T) & 7) || !ddl_aligned16(colz)) return DDL_EALIGN;
    if (!out_is_bf16 && !ddl_aligned16(out)) return DDL_EALIGN;
    const long long n4 = n / 4;
    const int sl = splits >= 64 ? 16 : (splits >= 8 ? 4 : 1);
    dim3 grid((unsigned)((n4 + 256 / sl - 1) / (256 / sl))), block(256);
#define RL3(TT, A, S) DDL_KLAUNCH((splitk_reduce_lin_kernel<TT, A, S, false>), grid, block, 0, stream, (const float*)partial, (TT*)out, n, splits, slab, \
                                        (const float*)coef, C, (const bf16_t*)T, (const float*)colz, row_len, (const float*)mean, 0.f)
#define RL(TT, A) do { if (sl == 16) RL3(TT, A, 16); else if (sl == 4) RL3(TT, A, 4); else RL3(TT, A, 1); } while (0)
    if (out_is_bf16) { if (accumulate) RL(bf16_t, true); else RL(bf16_t, false); }
    else { if (accumulate) RL(float, true); else RL(float, false); }
#undef RL3
#undef RL
    return ddl_launch_status();
}

// out (bf16 [C][C]) = sum_s partial[s] - colz (x) colz / M: the centred Gram matrix of a [M][C] tensor
extern "C" int ddl_splitk_reduce_centered(const void* partial, void* out, int C, int splits, const void* colz, long long M,
                                          hipStream_t stream) {
    if (!partial || !out || !colz || C <= 0 || (C & 3) || splits <= 0 || M <= 0) return DDL_EINVAL;
    if (!ddl_aligned16(partial) || (reinterpret_cast<uintptr_t>(out) & 7) || !ddl_aligned16(colz)) return DDL_EALIGN;
    const long long n = (long long)C * C, n4 = n / 4;
    const int sl = splits >= 64 ? 16 : (splits >= 8 ? 4 : 1);
    dim3 grid((unsigned)((n4 + 256 / sl - 1) / (256 / sl))), block(256);
#define RC(S) DDL_KLAUNCH((splitk_reduce_lin_kernel<bf16_t, false, S, true>), grid, block, 0, stream, (const float*)partial, (bf16_t*)out, n, splits, n, \
                                 (const float*)nullptr, C, (const bf16_t*)nullptr, (const float*)colz, C, (const float*)nullptr, 1.0f / (float)M)
    if (sl == 16) RC(16); else if (sl == 4) RC(4); else RC(1);
#undef RC
    return ddl_launch_status();
}
