// 3x3 stride-1 convolution, ALL couts of a pixel tile in ONE PERSISTENT block per CU (gfx950 / CDNA4).
//
// STATUS: EXPERIMENT, NOT ON THE DEFAULT ROUTE (VSE_C3_WIDE=1 sends the 3x3 layers with 128 couts here; results are bit-identical
// to conv_c3_kernel's, tests/test_gpu_nets.py::test_wide_3x3_route_gives_identical_bits).  Round 3's structural attempt at the
// 3x3 class.  Measured on the detector's 128 -> 128 layers @136x240 x 64: 0.74-0.76 ms against conv_c3_kernel's
// 0.64-0.65 — it loses, and the ablations / s_memtime traces (tools/ablate_c3w.sh, tools/trace_c3w.sh) say why:
//   * every global_load_lds costs the ISSUING wave ~180-200 cycles here (whatever the number of waves issuing at that moment:
//     two, four or eight gave the same per-instruction cost), 3.7 of them per wave and step = ~700 cycles during which the wave
//     issues no MFMA; with two waves per SIMD the partner has ~770 cycles of MFMA work per step to cover that — in practice the
//     DMA-issue + sync skeleton (0.24 ms with the MFMAs compiled out) and the MFMA time (0.33 ms) ADD UP to the layer time;
//     conv_c3_kernel's four waves per SIMD from two independent blocks are what hides DMA issue on this chip;
//   * a ping-pong form (LOAD / COMPUTE segments, waves 4..7 one segment behind waves 0..3, two barriers per step; git history)
//     removed the waits for data entirely (four steps of look-ahead) and was slower still (0.79 ms): the half-step overhead
//     (barrier + first-fragment latency, ~600 cycles) is paid twice;
//   * not persistent (VSE_C3W_GRID=0) +13 %; staging patches once instead of twice, the deep stream and the cross-tile prefetch
//     all work as designed (waits for data: 10 % of the loop with a 3-step look-ahead, ~0 with 4) — they are not what limits it.
//
// conv_c3_kernel (conv_c3.hip) gives a block 512 pixels x 64 couts and 80 KiB of LDS so that two blocks share a CU: the
// partner covers a block's prologue and epilogue, but a layer with 128 couts then stages every input patch twice, the
// 80 KiB buy a look-ahead of three short steps, and the counters say the waves wait for data (DESIGN 3.2 log).  This
// kernel takes the other road for the layers with 128 couts:
//
//   block   = one per CU (~150 KiB of LDS, 256 VGPRs), PERSISTENT: it walks pixel tiles t, t + nb, ... of its XCD's
//             contiguous tile range; the DMA stream (patches and weight stages) runs ACROSS tile boundaries, so only a
//             block's first tile pays a DMA round trip in front of its first MFMA.
//   tile    = (2 RW) x (32 CW) pixels x all BN = 32 TN couts; wave (rw, cw) owns 2 rows x 32 pixels x BN couts
//             (2 x TN accumulator tiles): an input patch is staged once, 6 TN MFMAs per wave and step.
//   step    = one filter column dx of one 16-channel chunk (as conv_c3_kernel / conv_col_kernel: fragment reads are
//             per-step base VGPRs + immediates); weight stage = [3 dy][BN][16] = 12 KiB for 128 couts.
//   stream  = SIX weight stages (stage s+5 is issued in step s) and THREE patch buffers (chunk c+2 is issued in the first
//             step of chunk c).  The wait in front of the barrier that opens step s+1 covers stage s+2 (issued three steps,
//             ~4.6k cycles of MFMA work, earlier), so stage s+1 is visible during step s and the first fragments of step
//             s+1 are read at the tail of step s, across the barrier (as conv_col_kernel); a step's DMAs are issued behind
//             its first eight MFMAs.
//   sync    = one raw s_barrier per step; vmcnt(3 WNPL + PNPL) in front of it: every wave issues WNPL weight DMAs per step
//             and PNPL patch DMAs per chunk, dummies included.  The epilogue's stores
//             (and residual loads) enter the same in-order queue behind the DMAs the next tile's first steps wait for, and
//             any extra entry only makes a counted wait stricter, never weaker.
//   K order = chunk-major, then dx, then dy — the order of conv_c3_kernel, so the two kernels give identical bits.
//   weights packed [cinp/16][3 dx][3 dy][Np][16] (compiler.col_weights, F_COL).
#include <stdlib.h>
#include "conv_common.h"

#define W3R 6
#define W3PB 3
#ifndef VSE_W3_ABL
#define VSE_W3_ABL 0      // timing-only ablations (tools/ablate_c3w.sh; results are garbage): 1 patch DMAs read the zero page,
#endif                    // 2 weight DMAs read the zero page, 4 no MFMAs, 8 no epilogue
#ifndef VSE_W3_STORECREDIT
#define VSE_W3_STORECREDIT 0   // 1: the counted waits of a tile's first three steps let the previous epilogue's stores fly (measured: no gain)
#endif
#ifndef VSE_W3_STAGGER
#define VSE_W3_STAGGER 1  // DMA issue slots of a step: 4 = two waves per slot (step start / behind tap 0 / 1 / 2; SIMD partners w, w+4 apart), 1 = behind tap 0 (waves 0..3) / tap 1, 0 = all behind tap 0
#endif
#ifdef VSE_TRACE
#include <stdio.h>
#include <vector>
#define TRT(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TRACC(acc, a, b) acc += (b) - (a)
#else
#define TRT(v) do { } while (0)
#define TRACC(acc, a, b) do { } while (0)
#endif

#if VSE_W3_ABL & 4
#define W3MFMA(a, b, c) (c)
#else
#define W3MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

template <int N> __device__ __forceinline__ void w3_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt literal");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <int RW, int CW, int TN>
__global__ __launch_bounds__(512, 2) void conv_c3w_kernel(const ConvParams p) {
    constexpr int BN = 32 * TN;
    constexpr int TH = 2 * RW, TW = 32 * CW;
    constexpr int PW = TW + 8, PH = TH + 2;
    constexpr int PPIX = (PH * PW + 31) / 32 * 32;       // whole wave instructions
    constexpr int PINSTR = PPIX / 32;
    constexpr int PNPL = (PINSTR + 7) / 8;
    constexpr int WROWS = 3 * BN;
    constexpr int WINSTR = WROWS / 32;
    constexpr int WNPL = (WINSTR + 7) / 8;
    constexpr int PATCH_HALFS = PPIX * 16, WSTAGE_HALFS = WROWS * 16;
    constexpr int PATCH_BYTES = PATCH_HALFS * 2, WSTAGE_BYTES = WSTAGE_HALFS * 2;
    constexpr int ROWB = PW * 32;
    constexpr int RING_BYTES0 = W3PB * PATCH_BYTES;
    static_assert(RW * CW == 8 && (PW / 8) % 2 == 1, "tile shapes");
    static_assert(W3PB * PATCH_BYTES + W3R * WSTAGE_BYTES + 1024 + 4 * BN <= 160 * 1024, "LDS");
    static_assert(3 * WNPL + PNPL <= 16, "vmcnt literal");
    __shared__ __attribute__((aligned(16))) half_t lds[W3PB * PATCH_HALFS + W3R * WSTAGE_HALFS + 512 + 2 * BN];   // the ONLY LDS object
    half_t* const ring0 = lds + W3PB * PATCH_HALFS;
    half_t* const dummy0 = ring0 + W3R * WSTAGE_HALFS;
    float* const sbias = reinterpret_cast<float*>(dummy0 + 512);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave / CW, cw = wave % CW;

    // ---- this block's tiles: XCD x (blockIdx & 7) owns a contiguous range of the tile sequence, its blocks interleave ----
    const unsigned G = gridDim.x, bid = blockIdx.x, xcd = bid & 7, bslot = bid >> 3;
    const unsigned T = p.ntiles, q8 = T >> 3, r8 = T & 7;
    const unsigned x0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned tend = x0 + q8 + (xcd < r8 ? 1u : 0u);
    const unsigned nbx = (G >> 3) + (xcd < (G & 7) ? 1u : 0u);
    const long img_halfs = (long)p.Hs * p.Ws * p.in_ld;
    int g_oy0 = 0, g_ox0 = 0;                              // geometry of the tile last decoded
    long g_img = 0;
    auto decode = [&](unsigned t) {
        const unsigned tx = t % (unsigned)p.tiles_w, r = t / (unsigned)p.tiles_w;
        g_ox0 = (int)tx * TW;
        g_oy0 = (int)(r % (unsigned)p.tiles_h) * TH;
        g_img = (long)(r / (unsigned)p.tiles_h);
    };
    // first tile at or behind t (in this block's sequence) with work in it; a ragged batch's tiles right of their sample are
    // written as zeros on the way
    auto next_live = [&](unsigned t) -> unsigned {
        while (t < tend) {
            decode(t);
            if (!conv_tile_right_of_sample<TH, TW>(p, g_img, g_oy0, g_ox0, 0, BN)) break;
            t += nbx;
        }
        return t;
    };
    auto offsets = [&](int (&poff)[PNPL]) {                // 32-bit element offsets inside the image, < 0: zero page
        int ln = lane;
        asm volatile("" : "+v"(ln));                       // recomputed per tile: hoisted, the per-lane parts would live (spilled) across the K loop
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const int q = 32 * (wave + 8 * j) + (ln >> 1);
            const int kh_ = (ln & 1) ^ ((q >> 3) & 1);
            const int py = q / PW, px = q - py * PW;
            const int iy = g_oy0 - 1 + py, ix = g_ox0 - 1 + px;
            const bool ok = (py < PH) && (px < TW + 2) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
            poff[j] = ok ? ((iy >> p.inshift) * p.Ws + (ix >> p.inshift)) * p.in_ld + kh_ * 8 : -1;
        }
    };
    unsigned t = bslot < nbx ? next_live(x0 + bslot) : tend;
    if (t >= tend) return;
    const int nchunks = p.cinp >> 4;
    const int nsteps = 3 * nchunks;

    int poffc[PNPL], poffn[PNPL];
    offsets(poffc);
#pragma unroll
    for (int j = 0; j < PNPL; ++j) poffn[j] = -1;
    int c_oy0 = g_oy0, c_ox0 = g_ox0;
    long c_img = g_img;
    const half_t* in_c = p.in + c_img * img_halfs;
    const half_t* in_n = in_c;
    unsigned tn = tend;
    bool have_next = false;

    auto issue_patch = [&](const half_t* img_base, const int (&poff)[PNPL], int chunk, int pb, bool live) {
        half_t* base = lds + pb * PATCH_HALFS;
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const int i = wave + 8 * j;
            const half_t* src = (live && poff[j] >= 0) ? img_base + poff[j] + chunk * 16 : p.zero;
            half_t* dst = base + i * 512;
            if (i >= PINSTR) { src = p.zero; dst = dummy0; }
            glds16_asm(src, dst);
        }
    };
    const half_t* wl[WNPL];
    bool wok[WNPL];
#pragma unroll
    for (int j = 0; j < WNPL; ++j) {
        const int row = 32 * (wave + 8 * j) + (lane >> 1);
        const int kh_ = (lane & 1) ^ ((row >> 3) & 1);
        const int dy = row / BN, r = row - dy * BN;
        wok[j] = (row < WROWS) && (r < p.Np);
        wl[j] = wok[j] ? p.w + ((long)dy * p.Np + r) * 16 + kh_ * 8 : p.zero;
    }
    const long winc = 3L * p.Np * 16;                     // elements per stage
    int wst = 0, wslot = 0;                                // tile-local index / ring slot of the next stage to issue
    auto issue_w = [&]() {
        half_t* st = ring0 + wslot * WSTAGE_HALFS;
        const long so = wst * winc;
#pragma unroll
        for (int j = 0; j < WNPL; ++j) {
            const int i = wave + 8 * j;
            glds16_asm(wok[j] && !(VSE_W3_ABL & 2) ? wl[j] + so : p.zero, i < WINSTR ? st + i * 512 : dummy0);
        }
        wst = wst + 1 == nsteps ? 0 : wst + 1;           // behind a tile's last stage: the next tile's first
        wslot = wslot + 1 == W3R ? 0 : wslot + 1;
    };

    // ---- fragment addressing (bytes) ------------------------------------------------------------------------------------
    const int fx = lane & 31, fj = lane >> 5;
    const int wr0 = conv_wrow(fx);
    const unsigned woffb = (unsigned)(RING_BYTES0 + wr0 * 32 + ((fj ^ ((wr0 >> 3) & 1)) << 4));
    const unsigned xrow0 = (unsigned)(2 * rw * ROWB);
    auto xcol = [&](int dx, int pb) -> unsigned {        // even-row base of the wave's fragments under column dx (odd rows: ^ 16)
        const unsigned c = (unsigned)(32 * cw + fx + dx);
        return (unsigned)pb * PATCH_BYTES + xrow0 + c * 32 + ((fj ^ ((c >> 3) & 1)) << 4);
    };
    const char* const ldsb = reinterpret_cast<const char*>(lds);

    // ---- prologue: the queue a steady-state step expects — P(0) W0 W1 P(1) W2 W3 W4 ------------------------------------
    conv_stage_consts<true>(sbias, p.bias, p.zero, 0, BN, p.Np, wave, lane);
    issue_patch(in_c, poffc, 0, 0, true);
    issue_w();
    issue_w();
    issue_patch(in_c, poffc, 1, 1, true);
    issue_w();
    issue_w();
    issue_w();
    wait_vm<3 * WNPL + PNPL>();                           // constants, patch 0, stages 0 and 1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#ifdef VSE_TRACE
    unsigned long long tr_wait = 0, tr_bar = 0, tr_dma = 0, tr_loop = 0, tr_epi = 0, tr_tiles = 0;
    const unsigned long long tr_begin = __builtin_amdgcn_s_memtime();
#endif
    // `stores`: a lower bound (0, 2 TN or 4 TN) on the VMEM instructions the wave's last epilogue issued, passed during the
    // three steps whose stage was issued BEFORE those stores: they are younger than what the wait is for, so they may fly too —
    // without this the first waits of a tile sit out the HBM round trip of the previous tile's stores (all blocks store at once)
    auto close_step = [&](int stores) __attribute__((always_inline)) {
        TRT(w0);
        // open step s+1: own DMAs of stage s+2 (issued in step s-3) and everything older landed — stage s+1 became visible one
        // barrier ago, which is what lets a step read the next step's first fragments at its tail; three younger stages and
        // one patch may fly
        if (stores == 0) w3_wait_vm<3 * WNPL + PNPL>();
        else if (stores == 2 * TN) w3_wait_vm<3 * WNPL + PNPL + 2 * TN>();
        else w3_wait_vm<3 * WNPL + PNPL + 4 * TN>();
        __builtin_amdgcn_sched_barrier(0);
        TRT(w1);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        TRT(w2);
        TRACC(tr_wait, w0, w1);
        TRACC(tr_bar, w1, w2);
    };

    int pbc = 0, cslot = 0;                                // patch buffer / ring slot being consumed
    int cc = 0;                                            // chunk of the current tile
    // a chunk's first step starts the patch two chunks ahead — in the next tile (found when the stream first reaches it) for a
    // tile's last two chunks
    auto stream_patch = [&]() __attribute__((always_inline)) {
        const int c2 = cc + 2, pb2 = pbc == 0 ? 2 : pbc - 1;
        if (c2 == nchunks) {
            tn = next_live(t + nbx);
            have_next = tn < tend;
            if (have_next) {
                offsets(poffn);
                in_n = p.in + g_img * img_halfs;
            }
        }
        const bool nx = c2 >= nchunks;
        const half_t* base = nx ? in_n : in_c;
        const int chunk = nx ? c2 - nchunks : c2;
        const bool live = nx ? have_next : true;
        half_t* dstb = lds + pb2 * PATCH_HALFS;
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const int i = wave + 8 * j;
            const int po = nx ? poffn[j] : poffc[j];
            const half_t* src = (live && po >= 0 && !(VSE_W3_ABL & 1)) ? base + po + chunk * 16 : p.zero;
            half_t* dst = dstb + i * 512;
            if (i >= PINSTR) { src = p.zero; dst = dummy0; }
            glds16_asm(src, dst);
        }
    };
    // the wave's DMA issue slot inside a step: a global_load_lds blocks its wave until the CU's one vector-memory pipe takes it,
    // so eight waves issuing at the same point queue behind each other while the matrix pipe starves
    const int dslot = VSE_W3_STAGGER == 4 ? ((wave < 4 ? wave : wave + 2) & 3) : VSE_W3_STAGGER == 1 ? (wave < 4 ? 1 : 2) : 1;
#define DMA_SLOT(k)                                                                                  \
    do {                                                                                             \
        if (dslot == (k)) {                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            TRT(d0_);                                                                                \
            if (dx == 0) stream_patch();                                                             \
            issue_w();                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            TRT(d1_);                                                                                \
            TRACC(tr_dma, d0_, d1_);                                                                 \
        }                                                                                            \
    } while (0)
    half8 X0, X1, Wc[TN];                                  // first fragments of the coming step (read across the barrier)
    bool pre_ok = false;
    int st_prev = 0;                                       // see close_step
    for (;;) {
        const bool wave_live = (c_oy0 + 2 * rw) < p.OH && (c_ox0 + 32 * cw) < p.OW;
        float16v acc[2][TN];
        if (!wave_live) {
            // a wave outside the map: DMA issue and barriers only (its partner on the SIMD gets the matrix pipe), no epilogue
            for (cc = 0; cc < nchunks; ++cc) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dx == 0) stream_patch();
                    issue_w();
                    close_step(cc == 0 ? st_prev : 0);
                    cslot = cslot + 1 == W3R ? 0 : cslot + 1;
                }
                pbc = pbc == 2 ? 0 : pbc + 1;
            }
            pre_ok = false;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            if (!pre_ok) {                                  // the wave sat the previous tile out: fetch what its tail would have
                const unsigned xe = xcol(0, pbc), wv = (unsigned)cslot * WSTAGE_BYTES + woffb;
                X0 = *reinterpret_cast<const half8*>(ldsb + xe);
                X1 = *reinterpret_cast<const half8*>(ldsb + (xe ^ 16u) + ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) Wc[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024);
                pre_ok = true;
            }
            TRT(l0);
#pragma unroll 1
            for (cc = 0; cc < nchunks; ++cc) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int nslot = cslot + 1 == W3R ? 0 : cslot + 1;
                    const int npb = pbc == 2 ? 0 : pbc + 1;
                    unsigned xe = xcol(dx, pbc);
                    unsigned xo = xe ^ 16u;
                    unsigned wv = (unsigned)cslot * WSTAGE_BYTES + woffb;
                    unsigned xne = dx == 2 ? xcol(0, npb) : xcol(dx + 1, pbc);
                    unsigned xno = xne ^ 16u;
                    unsigned wvn = (unsigned)nslot * WSTAGE_BYTES + woffb;
                    asm volatile("" : "+v"(xe), "+v"(xo), "+v"(wv), "+v"(xne), "+v"(xno), "+v"(wvn));
                    half8 Xn, Wn[TN];
                    DMA_SLOT(0);
                    // tap dy = 0: rows 0, 1 (held); fetch W[1], row 2 (even)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024 + BN * 32);
                    Xn = *reinterpret_cast<const half8*>(ldsb + xe + 2 * ROWB);
                    __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[0][j] = W3MFMA(Wc[j], X0, acc[0][j]);
                        acc[1][j] = W3MFMA(Wc[j], X1, acc[1][j]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the step's DMAs, behind the first MFMAs: the matrix pipe works them off while the wave issues
                    DMA_SLOT(1);
                    X0 = X1; X1 = Xn;
#pragma unroll
                    for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
                    // tap dy = 1: rows 1, 2; fetch W[2], row 3 (odd)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wv + j * 1024 + 2 * BN * 32);
                    Xn = *reinterpret_cast<const half8*>(ldsb + xo + 3 * ROWB);
                    __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[0][j] = W3MFMA(Wc[j], X0, acc[0][j]);
                        acc[1][j] = W3MFMA(Wc[j], X1, acc[1][j]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                    DMA_SLOT(2);
                    // tap dy = 2: rows 2, 3; fetch the first fragments of step s+1 (stage s+1 and a next chunk's patch are visible
                    // since the barrier that opened this step)
                    half8 Y0, Y1;
#pragma unroll
                    for (int j = 0; j < TN; ++j) Wc[j] = *reinterpret_cast<const half8*>(ldsb + wvn + j * 1024);
                    Y0 = *reinterpret_cast<const half8*>(ldsb + xne);
                    Y1 = *reinterpret_cast<const half8*>(ldsb + xno + ROWB);
                    __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[0][j] = W3MFMA(Wn[j], X1, acc[0][j]);
                        acc[1][j] = W3MFMA(Wn[j], Xn, acc[1][j]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                    X0 = Y0; X1 = Y1;
                    DMA_SLOT(3);
                    close_step(cc == 0 ? st_prev : 0);
                    cslot = nslot;
                }
                pbc = pbc == 2 ? 0 : pbc + 1;
            }
            TRT(l1);
            TRACC(tr_loop, l0, l1);
#ifdef VSE_TRACE
            ++tr_tiles;
#endif
        }

        TRT(e0);
        // ---- epilogue of tile t (the stream of the next tile is already in flight) ------------------------------------
        if (wave_live && !(VSE_W3_ABL & 8)) {
            int ln = lane;
            asm volatile("" : "+v"(ln));                   // as in offsets()
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = c_oy0 + 2 * rw + i, ox = c_ox0 + 32 * cw + (ln & 31);
                if (oy >= p.OH || ox >= p.OW) continue;
                const long m = (c_img * p.OH + oy) * p.OW + ox;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float bias[16];
                    conv_epilogue_consts(sbias, j * 32, ln, bias);
                    conv_epilogue_tile(p, acc[i][j], bias, m, c_img, oy, ox, j * 32, ln);
                }
            }
        }
        TRT(e1);
        TRACC(tr_epi, e0, e1);
#if VSE_W3_STORECREDIT
        st_prev = wave_live && !(VSE_W3_ABL & 8) ? ((c_oy0 + 2 * rw + 1) < p.OH ? 4 * TN : 2 * TN) : 0;
#endif
        if (!have_next) break;
        t = tn;
        c_oy0 = g_oy0; c_ox0 = g_ox0; c_img = g_img;
        in_c = in_n;
#pragma unroll
        for (int j = 0; j < PNPL; ++j) poffc[j] = poffn[j];
        have_next = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the look-ahead / dummy DMAs before LDS is released
#ifdef VSE_TRACE
    if (lane == 0 && p.trace) {
        unsigned long long* o = p.trace + ((unsigned long long)blockIdx.x * 8 + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - tr_begin; o[1] = tr_loop; o[2] = tr_wait; o[3] = tr_bar; o[4] = tr_dma; o[5] = tr_epi; o[6] = tr_tiles;
    }
#endif
}

// Layers this kernel serves (mirrored by vse_plan_op_kernel_name): conv_c3_kernel's, with all 128 couts in one block and at
// least two 16-channel chunks; in-image offsets are 32-bit.
bool conv_c3w_ok(const ConvParams& p) {
    return conv_c3_ok(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.cinp, p.flags) && !(p.flags & F_HILO) && p.Np == 128 && p.cinp >= 32
           && (double)p.Hs * p.Ws * p.in_ld < 2.0e9;
}

int launch_conv_c3w(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (!conv_c3w_ok(p)) return VSE_E_UNSUPPORTED;
    int rw;
    conv_c3_plan(p.OH, p.OW, &rw);
    const int cw = 8 / rw;
    p.ntn = 1;
    p.tiles_h = (p.OH + 2 * rw - 1) / (2 * rw);
    p.tiles_w = (p.OW + 32 * cw - 1) / (32 * cw);
    const unsigned long long tiles = (unsigned long long)n_img * p.tiles_h * p.tiles_w;
    if (tiles == 0 || tiles > 0x7fffffffull) return VSE_E_INVAL;
    p.ntiles = (unsigned)tiles;
    static const int cus = [] {
        const char* e = getenv("VSE_C3W_GRID");              // experiments: 0 = one block per tile (not persistent)
        if (e && e[0]) return atoi(e);
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount;
    }();
    const unsigned grid = cus > 0 && (unsigned long long)cus < tiles ? (unsigned)cus : (unsigned)tiles;
#ifdef VSE_TRACE
    static unsigned long long* trace_dev = nullptr;
    if (!trace_dev) (void)hipMalloc(&trace_dev, 1024 * 64 * sizeof(unsigned long long));
    (void)hipMemsetAsync(trace_dev, 0, (size_t)1024 * 64 * sizeof(unsigned long long), st);
    p.trace = grid <= 1024 ? trace_dev : nullptr;
#endif
    if (rw == 8) hipLaunchKernelGGL((conv_c3w_kernel<8, 1, 4>), dim3(grid), dim3(512), 0, st, p);
    else if (rw == 4) hipLaunchKernelGGL((conv_c3w_kernel<4, 2, 4>), dim3(grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_c3w_kernel<2, 4, 4>), dim3(grid), dim3(512), 0, st, p);
#ifdef VSE_TRACE
    if (p.trace) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)grid * 64);
        (void)hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            double d[7] = {0, 0, 0, 0, 0, 0, 0};
            size_t nb = 0;
            for (size_t b = 0; b < grid; ++b) {
                const unsigned long long* t = &h[(b * 8 + w) * 8];
                if (!t[6]) continue;                   // the wave sat every tile out
                for (int i = 0; i < 7; ++i) d[i] += (double)t[i];
                ++nb;
            }
            if (nb) fprintf(stderr, "[c3w trace] cin%d %dx%d rw%d grid %u wave %d: per block (s_memtime ticks) total %.0f, K loops %.0f (wait %.0f, barrier %.0f, DMA issue %.0f), "
                            "epilogues %.0f; tiles %.1f\n", p.cinp, p.OH, p.OW, rw, grid, w, d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb, d[4] / nb, d[5] / nb, d[6] / nb);
        }
    }
#endif
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
