// 3x3 stride-1 convolution over an LDS-resident 16-channel patch, TWO BLOCKS PER CU (gfx950 / CDNA4).
//
// The 3x3 sibling of conv_col_kernel (conv_col.hip): the same static structure — 16-channel chunks (one MFMA K slice),
// 32-byte patch pixels, a four-stage weight ring that runs one stage ahead of the consumer, every fragment read = a
// per-step base VGPR + an immediate, the next step's first fragments read across the barrier — sized so that two blocks
// share a CU (<= 80 KiB of LDS, <= 128 VGPRs): a 3x3 K loop is short (cin / 16 x 3 steps), and only a second resident
// block hides a block's prologue (first DMA round trip) and epilogue (stores) behind matrix work.
//
//   tile   = (2 RW) rows x (32 CW) pixels x 64 couts, RW x CW = 8 waves: 16 x 32, 8 x 64 or 4 x 128 — chosen per map so that
//            tall detector maps and the 12- / 6-row recogniser maps both tile well; wave (rw, cw) owns rows 2rw, 2rw+1 of pixel
//            columns 32cw .. 32cw+31 and all 64 couts (2 x 2 accumulator tiles); waves outside the map idle.
//   step   = one filter column dx of one chunk: 3 taps x 4 MFMAs per wave; ring stage = [3 dy][64 couts][16 ch] = 6 KiB.
//   patch  = (2 RW + 2) rows, row stride PW = 32 CW + 8 pixels (an odd multiple of 8): the bank swizzle
//            slot = k-half ^ ((pixel >> 3) & 1) = k-half ^ ((row + (col >> 3)) & 1) flips with the row parity only, so a step
//            needs two base VGPRs (even / odd rows: base ^ 16) and immediates.
//   sync   = as conv_col_kernel: one raw s_barrier per step, vmcnt(1) (+ the patch DMAs at a chunk's first step).
//   couts  = ceil(Np / 64) cout tiles per pixel tile (innermost in the block order: the tiles that share a patch run together).
//   weights packed [cinp/16][3 dx][3 dy][Np][16] + 3 zero stages (compiler.col_weights, F_COL).
#include <stdlib.h>
#include "conv_common.h"
#ifdef VSE_TRACE
#include <stdio.h>
#include <vector>
#define TR_STAMP(i) do { if (tid == 0) tr[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TR_STAMP(i) do { } while (0)
#endif

#ifndef VSE_C3_XPRE
#define VSE_C3_XPRE 0     // 1: ring one stage ahead of the consumer + the next step's first fragments read across the barrier (as
                          // conv_col_kernel); 0: plain look-ahead of three stages.  With two blocks per CU the partner block covers
                          // the barrier bubble, and a stage's DMA round trip (~5k cycles under load) needs the third step of slack.
#endif
// (Measured and removed: a PERSISTENT form — a block walking tiles lb, lb + 512, ... with the DMA ring running across tile
// boundaries, so that only a block's first tile pays the prologue's DMA round trip.  Correct, but 9-25 % SLOWER on every
// layer (128->128 0.687 -> 0.751 ms, 64->64 0.76 -> 0.96 ms): the epilogue's stores sit in the same in-order vmcnt queue as
// the LDS-DMAs, so the first counted waits of the next tile also wait for the stores' HBM round trip.)
#define C3RING 4
#define C3BN 64
// Timing-only ablations for the Winograd F(2x2, 3x3) go / no-go (EXPERIMENTS G; results are WRONG, never in a product build):
//   VSE_C3_ABL = 1: 16 of the 36 MFMAs per chunk and wave (the MFMA count of the 16 frequency GEMMs over the same outputs), every DMA,
//                   fragment read, wait and barrier unchanged — an UPPER bound for a Winograd form inside this skeleton (its 16 / 9 x
//                   larger weight stream, input transform and output transform all cost extra);
//   VSE_C3_ABL = 2: + 128 v_pk_add_f16 per chunk and wave (the packed adds of B^T d B on 8-channel fragments).
#ifndef VSE_C3_ABL
#define VSE_C3_ABL 0
#endif
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
#if VSE_C3_ABL && !defined(VSE_DEV_BUILD)
#error "VSE_C3_ABL is a development-build ablation"
#endif

// BN = 64 couts per block (TN = 2 accumulator tiles per row) or, for layers with <= 32 couts (the mobile detectors' 96 -> 24 neck convs,
// the server detector's 32 -> 32), BN = 32: half the MFMAs, half the weight stage; everything else is the same code.
template <int RW, int CW, int BN>
__device__ __forceinline__ void conv_c3_body(const ConvParams& p) {
    constexpr int TN = BN / 32;
    constexpr int TH = 2 * RW, TW = 32 * CW;
    constexpr int PW = TW + 8, PH = TH + 2;
    constexpr int PPIX = (PH * PW + 31) / 32 * 32;       // whole wave instructions
    constexpr int PINSTR = PPIX / 32;
    constexpr int PNPL = (PINSTR + 7) / 8;
    constexpr int WROWS = 3 * BN;
    constexpr int WINSTR = WROWS / 32;                   // 6: waves 0..5
    constexpr int PATCH_HALFS = PPIX * 16, WSTAGE_HALFS = WROWS * 16;
    constexpr int PATCH_BYTES = PATCH_HALFS * 2, WSTAGE_BYTES = WSTAGE_HALFS * 2;
    constexpr int ROWB = PW * 32;
    static_assert(RW * CW == 8 && (PW / 8) % 2 == 1, "tile shapes");
    static_assert(2 * PATCH_BYTES + C3RING * WSTAGE_BYTES + 1024 + 4 * BN <= 81920, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) half_t lds[2 * PATCH_HALFS + C3RING * WSTAGE_HALFS + 512 + 2 * BN];   // the ONLY LDS object
    half_t* const patch0 = lds;
    half_t* const ring0 = lds + 2 * PATCH_HALFS;
    half_t* const dummy0 = ring0 + C3RING * WSTAGE_HALFS;
    float* const sbias = reinterpret_cast<float*>(dummy0 + 512);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave / CW, cw = wave % CW;
#ifdef VSE_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_sync = 0;
#endif
    TR_STAMP(0);

    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int nt = t % p.ntn;  t /= p.ntn;
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nt * BN;
    if (conv_tile_right_of_sample<TH, TW>(p, img, oy0, ox0, n0, BN)) return;      // ragged batch: nothing to compute here
    const int nch1 = p.cinp >> 4;                        // 16-channel chunks of one pass over the input
    // F_HILO: fp16 hi + lo weight pairs — the lo stream follows the hi stream, the patch chunks are walked a second time into
    // the same accumulators (the implicit-GEMM kernels' two-pass K walk)
    const int nchunks = (p.flags & F_HILO) ? 2 * nch1 : nch1;

    // ---- DMA source state: 32-bit element offsets from the tensor bases -----------------------------------------------
    int poff[PNPL];                                        // < 0: zero page
#pragma unroll
    for (int j = 0; j < PNPL; ++j) {
        const int q = 32 * (wave + 8 * j) + (lane >> 1);
        const int kh_ = (lane & 1) ^ ((q >> 3) & 1);
        const int py = q / PW, px = q - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = (py < PH) && (px < TW + 2) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        const long o = ((img * p.Hs + (iy >> p.inshift)) * p.Ws + (ix >> p.inshift)) * (long)p.in_ld + kh_ * 8;
        poff[j] = ok ? (int)(o - img * (long)p.Hs * p.Ws * p.in_ld) : -1;
    }
    const half_t* const in_img = p.in + img * (long)p.Hs * p.Ws * p.in_ld;
    const half_t* wptr;
    bool wok;
    const int wnp = p.wnp;                                 // weight rows per tap (= Np; F_HLSUM: hi 32 | lo 32)
    const int winc = 3 * wnp * 16;                         // elements per stage (wave-uniform; masked per lane at issue)
    {
        const int row = 32 * wave + (lane >> 1);
        const int kh_ = (lane & 1) ^ ((row >> 3) & 1);
        const int dy = row / BN, r = row - dy * BN;
        wok = (row < WROWS) && (n0 + r < wnp);
        wptr = wok ? p.w + ((long)dy * wnp + n0 + r) * 16 + kh_ * 8 : p.zero;
    }
    auto issue_patch = [&](int cc) {
        half_t* base = patch0 + (cc & 1) * PATCH_HALFS;
        const bool live = cc < nchunks;
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const int i = wave + 8 * j;
            const half_t* src = (live && poff[j] >= 0) ? in_img + poff[j] + (cc >= nch1 ? cc - nch1 : cc) * 16 : p.zero;
            half_t* dst = base + i * 512;
            if (i >= PINSTR) { src = p.zero; dst = dummy0; }
            glds16_asm(src, dst);
        }
    };
    auto issue_w = [&](int s) {
        half_t* st = ring0 + (s & (C3RING - 1)) * WSTAGE_HALFS;
        glds16_asm(wptr, wave < WINSTR ? st + wave * 512 : dummy0);
        wptr += wok ? winc : 0;
    };

    // ---- fragment addressing (bytes) ------------------------------------------------------------------------------------
    const int fx = lane & 31, fj = lane >> 5;
    // weight row of cout tile j = row of tile 0 + 32 rows (+1024 bytes; the swizzle bit (r >> 3) & 1 is the same)
    const int wr0 = conv_wrow(fx);
    const unsigned woffb = (unsigned)(2 * PATCH_BYTES + wr0 * 32 + ((fj ^ ((wr0 >> 3) & 1)) << 4));
    const unsigned xrow0 = (unsigned)(2 * rw * ROWB);
    auto xcol = [&](int dx, int buf) -> unsigned {       // even-row base of the wave's fragments under column dx (odd rows: ^ 16)
        const unsigned c = (unsigned)(32 * cw + fx + dx);
        return (unsigned)buf * PATCH_BYTES + xrow0 + c * 32 + ((fj ^ ((c >> 3) & 1)) << 4);
    };
    const char* const ldsb = reinterpret_cast<const char*>(lds);
    const bool wave_live = (oy0 + 2 * rw) < p.OH && (ox0 + 32 * cw) < p.OW;

    float16v acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    conv_stage_consts<true>(sbias, p.bias, p.zero, n0, BN, p.Np, wave, lane);      // wave 0
    issue_patch(0);
    issue_w(0);
    issue_w(1);
    issue_w(2);
#if VSE_C3_XPRE
    wait_vm<1>();                                       // constants, patch 0, stages 0 and 1
#else
    wait_vm<2>();                                       // constants, patch 0, stage 0
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TR_STAMP(1);

    auto wbase = [&](int s_) __attribute__((always_inline)) -> unsigned {
        unsigned v = (unsigned)(s_ & (C3RING - 1)) * WSTAGE_BYTES + woffb;
        asm volatile("" : "+v"(v));
        return v;
    };
    auto close_step = [&](int dx) __attribute__((always_inline)) {
#ifdef VSE_TRACE
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
#if VSE_C3_XPRE
        // open step s+1: own DMAs of stage s+2 landed (+ the next chunk's patch unless it was issued in this step)
        if (dx == 0) wait_vm<1 + PNPL>();
        else wait_vm<1>();
#else
        // open step s+1: own DMAs of stage s+1 landed; two younger stages (and a patch issued after stage s+1) may fly
        if (dx == 2) wait_vm<2>();
        else wait_vm<2 + PNPL>();
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
#ifdef VSE_TRACE
        t_sync += __builtin_amdgcn_s_memtime() - tw0;
#endif
    };
    if (!wave_live) {
        // a wave outside the map: DMA issue and barriers only (its partner on the SIMD gets the matrix pipe), no epilogue
        int s = 0;
        for (int cc = 0; cc < nchunks; ++cc) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx, ++s) {
                if (dx == 0) issue_patch(cc + 1);
                issue_w(s + 3);
                close_step(dx);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    half8 X0, X1, Wc[TN];
#if VSE_C3_XPRE
    {
        const unsigned xe = xcol(0, 0), wv0 = wbase(0);
        X0 = *reinterpret_cast<const half8*>(ldsb + xe);
        X1 = *reinterpret_cast<const half8*>(ldsb + (xe ^ 16u) + ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j) Wc[j] = *reinterpret_cast<const half8*>(ldsb + wv0 + j * 1024);
    }
#endif
    int s = 0;
#pragma unroll 1
    for (int cc = 0; cc < nchunks; ++cc) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx, ++s) {
            if (dx == 0) issue_patch(cc + 1);
            issue_w(s + 3);
            unsigned xe = xcol(dx, cc & 1);
            unsigned xne = dx == 2 ? xcol(0, (cc + 1) & 1) : xcol(dx + 1, cc & 1);
            unsigned xo = xe ^ 16u, xno = xne ^ 16u;
            asm volatile("" : "+v"(xe), "+v"(xo), "+v"(xne), "+v"(xno));
            const unsigned wv = wbase(s), wvn = wbase(s + 1);
            half8 Wn[TN], Xn, Xn0;
#if !VSE_C3_XPRE
            X0 = *reinterpret_cast<const half8*>(ldsb + xe);
            X1 = *reinterpret_cast<const half8*>(ldsb + xo + ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) Wc[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024);
            (void)wvn; (void)xne; (void)xno;
#endif
            // tap dy = 0: rows k = 0, 1 (held); fetch W[1], row k = 2 (even)
#pragma unroll
            for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024 + BN * 32);
            Xn = *reinterpret_cast<const half8*>(ldsb + xe + 2 * ROWB);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X1, acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
            X0 = X1; X1 = Xn;
#pragma unroll
            for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
            // tap dy = 1: rows 1, 2; fetch W[2], row k = 3 (odd)
#pragma unroll
            for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024 + 2 * BN * 32);
            Xn = *reinterpret_cast<const half8*>(ldsb + xo + 3 * ROWB);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);
            if (!VSE_C3_ABL || dx == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X1, acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
            } else {                                     // ablation: the fragments stay read (kept alive), no matrix work
                asm volatile("" :: "v"(X0), "v"(Wc[0]));
            }
#if VSE_C3_ABL == 2
            {
                unsigned t0 = __builtin_bit_cast(uint4v, X0)[0], t1 = __builtin_bit_cast(uint4v, X0)[1], t2 = __builtin_bit_cast(uint4v, X1)[2], t3 = __builtin_bit_cast(uint4v, X1)[3];
#pragma unroll
                for (int r = 0; r < (dx == 0 ? 11 : 10); ++r) {          // 4 x (11 + 10 + 10) + ... ~ 128 packed adds per chunk
                    asm volatile("v_pk_add_f16 %0, %0, %1\n\tv_pk_add_f16 %1, %1, %2\n\tv_pk_add_f16 %2, %2, %3\n\tv_pk_add_f16 %3, %3, %0"
                                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                }
                asm volatile("" :: "v"(t0), "v"(t1), "v"(t2), "v"(t3));
            }
#endif
            X0 = X1; X1 = Xn;
#pragma unroll
            for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
            // tap dy = 2: rows 2, 3
#if VSE_C3_XPRE
            // fetch the first fragments of step s+1 (visible since the barrier that opened this step)
#pragma unroll
            for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wvn + j * 1024);
            Xn0 = *reinterpret_cast<const half8*>(ldsb + xne);
            Xn = *reinterpret_cast<const half8*>(ldsb + xno + ROWB);
            __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#endif
            if (!VSE_C3_ABL) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X1, acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
            } else {
                asm volatile("" :: "v"(X0), "v"(X1), "v"(Wc[0]));
            }
#if VSE_C3_XPRE
            X0 = Xn0; X1 = Xn;
#pragma unroll
            for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
#endif
            close_step(dx);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_STAMP(2);

    // ---- epilogue ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * rw + i, ox = ox0 + 32 * cw + fx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const long m = (img * p.OH + oy) * p.OW + ox;
        if constexpr (TN == 2) {
            if (p.flags & F_HLSUM) {                     // tile 0 = W_hi x, tile 1 = W_lo x of the same 32 couts
                float bias[16];
                conv_epilogue_consts(sbias, 0, lane, bias);
                float16v sum;
#pragma unroll
                for (int e = 0; e < 16; ++e) sum[e] = acc[i][0][e] + acc[i][1][e];
                conv_epilogue_tile(p, sum, bias, m, img, oy, ox, n0, lane);
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, img, oy, ox, n0 + j * 32, lane);
        }
    }
#ifdef VSE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && p.trace) {
        tr[3] = __builtin_amdgcn_s_memtime();
        tr[4] = t_sync;
        for (int i = 0; i < 8; ++i) p.trace[(unsigned long long)blockIdx.x * 8 + i] = tr[i];
    }
#endif
}

template <int RW, int CW>
__global__ __launch_bounds__(512, 4) void conv_c3_kernel(const ConvParams p) { conv_c3_body<RW, CW, C3BN>(p); }
template <int RW, int CW>
__global__ __launch_bounds__(512, 4) void conv_c3n32_kernel(const ConvParams p) { conv_c3_body<RW, CW, 32>(p); }

// Tile shape per map: estimated cost (in full tiles) of covering OH x OW with (2 RW) x (32 CW) tiles when waves outside the
// map idle (a partial tile costs ~0.35 + 0.65 * live waves / 8 of a full one).  Mirrored by compiler.py (c3_tile_eff).
static double c3_axis_cost(int n, int unit, int waves) {      // n pixels along an axis covered by tiles of `waves` x `unit`
    const int tile = unit * waves, full = n / tile, rem = n - full * tile;
    return full + (rem ? 0.35 + 0.65 * ((rem + unit - 1) / unit) / (double)waves : 0.0);
}
double conv_c3_plan(int OH, int OW, int* rw_out) {
    double best = 0;
    int brw = 8;
    for (int rw = 8; rw >= 2; rw >>= 1) {
        const int cw = 8 / rw;
        // partial tiles in both directions: live fraction multiplies; approximate by the product of the axis costs
        const double cost = c3_axis_cost(OH, 2, rw) * c3_axis_cost(OW, 32, cw) * 512.0;
        const double eff = (double)OH * OW / cost;
        if (eff > best + 1e-9) { best = eff; brw = rw; }
    }
    if (rw_out) *rw_out = brw;
    return best;
}
bool conv_c3_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int flags) {
    return kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && (cinp & 15) == 0
           && !(flags & (F_SRC2 | F_PIXSHUF | F_DOT1));
}

int launch_conv_c3(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (!conv_c3_ok(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.cinp, p.flags)) return VSE_E_UNSUPPORTED;
    if ((double)p.Hs * p.Ws * p.in_ld > 2.0e9) return VSE_E_UNSUPPORTED;      // 32-bit in-image offsets
#ifdef VSE_DEV_BUILD      // kernel experiments (conv_c3w.hip, forced tile shapes): compiled only into development builds
    static const int wide = [] { const char* e = getenv("VSE_C3_WIDE"); return e && e[0] ? atoi(e) : 0; }();
    if (wide == 1 && conv_c3w_ok(p)) return launch_conv_c3w(pin, n_img, st);
    if (wide == 2 && p.Np == 128 && !(p.flags & F_HILO)) return launch_conv_col3w(pin, n_img, st);
#endif
    int rw;
    conv_c3_plan(p.OH, p.OW, &rw);
#ifdef VSE_DEV_BUILD
    static const int force = [] { const char* e = getenv("VSE_C3_RW"); return e && e[0] ? atoi(e) : 0; }();
    if (force == 8 || force == 4 || force == 2) rw = force;
#endif
    const int cw = 8 / rw;
    const bool hlsum = (p.flags & F_HLSUM) != 0;
    if (hlsum && (p.Np > 32 || (p.flags & F_HILO))) return VSE_E_INVAL;
    const int bn = (p.Np <= 32 && !hlsum) ? 32 : C3BN;
    p.wnp = hlsum ? 64 : p.Np;
    p.ntn = hlsum ? 1u : (unsigned)((p.Np + bn - 1) / bn);
    p.tiles_h = (p.OH + 2 * rw - 1) / (2 * rw);
    p.tiles_w = (p.OW + 32 * cw - 1) / (32 * cw);
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w * p.ntn;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    const dim3 grid((unsigned)blocks), block(512);
#ifdef VSE_TRACE
    static unsigned long long* trace_dev = nullptr;
    static size_t trace_cap = 0;
    if (trace_cap < blocks * 8) {
        if (trace_dev) (void)hipFree(trace_dev);
        (void)hipMalloc(&trace_dev, blocks * 8 * sizeof(unsigned long long));
        trace_cap = blocks * 8;
    }
    (void)hipMemsetAsync(trace_dev, 0, blocks * 8 * sizeof(unsigned long long), st);
    p.trace = trace_dev;
#endif
    if (bn == 32) {
        if (rw == 8) hipLaunchKernelGGL((conv_c3n32_kernel<8, 1>), grid, block, 0, st, p);
        else if (rw == 4) hipLaunchKernelGGL((conv_c3n32_kernel<4, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_c3n32_kernel<2, 4>), grid, block, 0, st, p);
    } else if (rw == 8) hipLaunchKernelGGL((conv_c3_kernel<8, 1>), grid, block, 0, st, p);
    else if (rw == 4) hipLaunchKernelGGL((conv_c3_kernel<4, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_c3_kernel<2, 4>), grid, block, 0, st, p);
#ifdef VSE_TRACE
    {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
        double d[4] = {0, 0, 0, 0};
        size_t nb = 0;
        for (size_t b = 0; b < blocks; ++b) {
            const unsigned long long* t = &h[b * 8];
            if (!t[3]) continue;           // wave 0 of the block was outside the map
            d[0] += (double)(t[1] - t[0]); d[1] += (double)(t[2] - t[1]); d[2] += (double)(t[3] - t[2]); d[3] += (double)t[4];
            ++nb;
        }
        fprintf(stderr, "[c3 trace] cin%d N%d %dx%d rw%d blocks %llu: per block (s_memtime ticks) prologue %.0f, loop %.0f (of which wait+barrier %.0f), "
                "epilogue %.0f; steps %d\n", p.cinp, p.Np, p.OH, p.OW, rw, blocks, d[0] / nb, d[1] / nb, d[3] / nb, d[2] / nb, (p.cinp >> 4) * 3);
    }
#endif
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
