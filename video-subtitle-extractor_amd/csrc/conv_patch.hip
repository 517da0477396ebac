// k x k stride-1 convolution with an LDS-RESIDENT INPUT PATCH (gfx950 / CDNA4).
//
// The implicit-GEMM kernel (conv_mfma.hip) re-fetches its activation tile from L2 for every filter tap; PMC showed
// it bound by the L2 -> L1 -> LDS fill path (TA busy 50-70 %, L1 hit rate < 20 %, MFMA busy < 20 %).  Here a block
// owns an 8 x 32 tile of output pixels of ONE image and, per 32-channel chunk, pulls the (8+kh-1) x (32+kw-1) halo
// patch into LDS ONCE; all kh*kw taps then read their MFMA B-fragments from that patch at shifted pixel addresses.
// Activation bytes crossing L2->LDS drop by ~kh*kw / halo overhead (9x9: 81 taps / 2.5 = 32x; 3x3: 9 / 1.33 = 6.8x);
// what is left to stream per (chunk, tap) step is the [BN][32] weight tile (4-8 KiB), through a 4-deep LDS-DMA ring.
//
//   tile   = TH x 32 output pixels x BN couts, TH = 16 when the halo patch fits 640 pixels (3x3, 1xk), else 8
//   block  = 512 threads = 8 waves; TH=8: 4 (pixel rows 2w,2w+1) x 2 (cout halves); TH=16: 8 x 1 (all couts):
//            wave tile = 64 px x {BN/2 | BN} couts, up to 32 MFMAs per wave between barriers
//   LDS    = 2 x 40 KiB patch (double-buffered across channel chunks) + 4 x 8 KiB weight ring = 112 KiB, one object
//   step   = FOUR filter taps (32 MFMAs per wave between barriers; 9x9 / 7x7 / 5x5 with the 960-pixel patch: TWO, the
//            weight ring then has 8 KiB stages); tap counts are padded with zero-weight taps to a whole step
//   sync   = ONE raw s_barrier per step; LDS-DMA completion by counted s_waitcnt vmcnt(N): per step every thread
//            issues exactly 2 weight DMAs (+5 patch DMAs at step 0 of a chunk, zero-page dummies included), so N is
//            a literal: 9 at steps 1,2 of a chunk (the patch DMAs of the NEXT chunk are younger than the stage
//            waited for), 4 otherwise (look-ahead 3 steps)
//   banks  = 64-byte rows; 16-byte slot s of row/pixel q holds logical k-vector s ^ ((q >> 2) & 3), applied on the
//            DMA source side; 16 consecutive pixels at ANY alignment then hit 16 distinct bank groups (shifted taps
//            stay conflict-free)
//   weights are packed [chunk][tap][Np][32] by the compiler for this kernel (F_PATCH) so the stream is sequential;
//   taps in COLUMN-major order (dx outer, dy inner): consecutive taps of a column share one of their two activation
//   fragments, which is carried in registers (LDS fragment reads per MFMA 1.0 -> 0.78 for 9x9).
#include <stdlib.h>
#include "conv_common.h"
#ifdef VSE_TRACE
#include <stdio.h>
#include <vector>
#endif

#ifndef VSE_ABLATE
#define VSE_ABLATE 0      // 1: no s_barrier  2: no fragment ds_reads  3: no weight/patch DMA  4: no MFMA   (timing experiments only)
#endif

#ifndef VSE_PIPE
#define VSE_PIPE 1        // software-pipelined fast step (A/B on one box: tools/ab.sh conv_patch VSE_PIPE ...)
#endif
#ifndef VSE_PIPE128
#define VSE_PIPE128 0     // ... also for the 128-cout LIGHT tile (spills: 68 bytes of scratch per lane)
#endif
#ifndef VSE_EDGE_SKIP
#define VSE_EDGE_SKIP 1   // waves whose output rows lie below the map skip their fragment reads and MFMAs (A/B: tools/ab.sh)
#endif
#define PTW 32
#define PRING 4

// -DVSE_TRACE: wave 0 of every block stamps s_memtime at phase boundaries into p.trace[block][8] (timing experiments
// only; tools/trace_patch.sh): 0 start, 1 setup done, 2 first barrier passed, 3 K loop done, 4 epilogue done,
// 5 = cycles spent in (wait + barrier) summed over the loop
#ifdef VSE_TRACE
#define TR_STAMP(i) do { if (tid == 0) tr[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TR_STAMP(i) do { } while (0)
#endif

// MODE 1 = BIGP: 960-pixel patch (16-row tiles under 9x9 / 7x7 / 5x5 filters) and a TWO-stage weight ring of 4-tap stages:
//   2 x 60 KiB patch + 2 x 16 KiB ring + 4 KiB dummy = 156 KiB;  MODE 0: 2 x 40 KiB + 4 x 16 KiB = 144 KiB.
// MODE 2 = LIGHT (3x3 / 1xk filters on 8-row tiles): 352-pixel patch, TWO taps per step, 64 or 128 couts per tile:
//   2 x 22 KiB patch + 2 x {8 | 16} KiB ring + dummies = 63 / 79 KiB -> TWO blocks per CU (4 waves per SIMD), the
//   occupancy the implicit-GEMM kernel has, with the patch kernel's activation reuse.
template <int PTH, int BN, int MODE>
__global__ __launch_bounds__(512, MODE == 2 ? 4 : 2) void conv_patch_kernel(const ConvParams p) {
    constexpr bool BIGP = MODE == 1, LIGHT = MODE == 2;
    static_assert(!BIGP || ((BN == 64 || BN == 32) && PTH == 16), "big patch variant");
    static_assert(!LIGHT || PTH == 8, "light variant");
    constexpr int WCO = PTH == 16 ? 1 : 2;          // waves along cout
    constexpr int TN = BN / (32 * WCO);             // 32-cout MFMA tiles per wave
    constexpr int PPIX = BIGP ? 960 : LIGHT ? 352 : 640;   // patch capacity in pixels
    constexpr int PNPL = BIGP ? 8 : LIGHT ? 3 : 5;  // patch DMAs per thread per chunk (512 threads x 16 B each)
    constexpr int RROWS = LIGHT ? BN : 64;          // weight rows per tap in a ring stage
    constexpr int TPS = LIGHT ? 2 : 4;              // filter taps per step: 16 / 32 MFMAs per wave between barriers
    constexpr int RING = (BIGP || LIGHT) ? 2 : PRING;   // weight ring stages (2 where LDS is tight: 960-pixel patch, two blocks per CU)
    constexpr bool PIPE = VSE_ABLATE == 0 && VSE_PIPE && (VSE_PIPE128 || !(LIGHT && BN == 128));   // fast step (below); the 128-cout LIGHT tile has no registers to spare
    constexpr int LOOK = RING - 1;                  // stages in flight ahead of the one being consumed
    constexpr int PATCH_HALFS = PPIX * 32, WSTAGE_HALFS = TPS * RROWS * 32;
    // landing zone of surplus DMAs (whole wave instructions past the patch / past a 64-row weight stage)
    constexpr int DUMMY_HALFS = BIGP ? 4 * 512 : LIGHT ? 2 * 512 : 0;
    static_assert(BN == 64 || (LIGHT && BN == 128) || (BIGP && BN == 32), "cout tile");
    __shared__ __attribute__((aligned(16))) half_t lds[2 * PATCH_HALFS + RING * WSTAGE_HALFS + DUMMY_HALFS + 4 * BN];   // the ONLY LDS object
    // ... + BN floats of bias + BN floats of F_DOT1 projection weights
    float* const sbias = reinterpret_cast<float*>(lds + 2 * PATCH_HALFS + RING * WSTAGE_HALFS + DUMMY_HALFS);
    float* const sdotw = sbias + BN;
    half_t* const patch0 = lds;
    half_t* const ring0 = lds + 2 * PATCH_HALFS;
    half_t* const dummy0 = ring0 + RING * WSTAGE_HALFS;   // BIGP: landing zone of the 4 surplus patch DMAs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = wave / WCO, wco = wave % WCO;
#ifdef VSE_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_sync = 0;
#endif
    TR_STAMP(0);

    // XCD-aware bijective block order (see conv_mfma.hip): contiguous logical range per XCD, cout tiles innermost
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int nt = t % p.ntn;  t /= p.ntn;
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * PTH, ox0 = tx * PTW, n0 = nt * BN;

    const int PW = PTW + p.kw - 1, PH = PTH + p.kh - 1, P = PW * PH;
    const int taps = p.kh * p.kw;
    const int pairs = (taps + TPS - 1) / TPS;              // steps per chunk (TPS taps each; the stream is padded to that)
    const int nchunks = (p.cinp + 31) >> 5;
    const int total = nchunks * pairs;

    // ---- DMA source state ---------------------------------------------------------------------------------
    const int kv = (lane & 3) ^ ((lane >> 4) & 3);        // logical k-vector this lane fetches (source-side swizzle)
    const bool src2 = !LIGHT && (p.flags & F_SRC2);        // virtual concat: vectors >= nv0 come from p.in2 (never planned onto LIGHT)
    long poff[PNPL], poff2[PNPL];
    bool pok[PNPL];
#pragma unroll
    for (int j = 0; j < PNPL; ++j) {
        const int q = 16 * (wave + 8 * j) + (lane >> 2);    // patch pixel index; wave instruction covers 16 pixels
        const int py = q / PW, px = q - py * PW;
        const int iy = oy0 - p.ph + py, ix = ox0 - p.pw + px;
        pok[j] = (q < P) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        poff[j] = ((img * p.Hs + (iy >> p.inshift)) * p.Ws + (ix >> p.inshift)) * (long)p.in_ld + kv * 8;
        poff2[j] = src2 ? ((img * p.in2_hs + (iy >> p.in2_shift)) * p.in2_ws + (ix >> p.in2_shift)) * (long)p.in2_ld
                              + (kv - p.nv0) * 8
                        : 0;
    }
    // weight DMA: thread -> cout row (tid>>2)&63; waves 0-3 fetch the even taps of a step, waves 4-7 the odd ones
    // (BIGP: 2 taps per step = 1 DMA per thread; otherwise 4 taps per step = 2 DMAs per thread)
    const int wr = (LIGHT && BN == 128) ? (tid >> 2) : ((tid >> 2) & 63);     // 128-cout tiles: thread -> row 0..127
    const bool wok = (wr < BN) && (n0 + wr < p.Np);
    // running DMA source of this thread: advanced by one step (TPS taps) per issue_w; the compiler appends
    // PATCH_WPAD_STEPS zero steps to the packed stream, so the look-ahead past the last real step reads real zeros
    // (no select in the loop); rows beyond the cout range park on the zero page and never move
    const half_t* wptr = wok ? p.w + (long)(n0 + wr) * 32 + kv * 8 + ((LIGHT && BN == 128) ? 0 : (long)(wave >> 2) * p.Np * 32) : p.zero;
    const long winc = wok ? (long)p.Np * 32 * TPS : 0;

    auto issue_patch = [&](int cc, int buf) {
        half_t* base = patch0 + buf * PATCH_HALFS;
        const bool live = (cc < nchunks) && (cc * 32 + kv * 8 < p.cinp);   // channel tail of the last chunk -> zeros
        const bool second = src2 && (cc * 4 + kv >= p.nv0);
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const half_t* src = p.zero;
            if (live && pok[j]) src = second ? p.in2 + poff2[j] + cc * 32 : p.in + poff[j] + cc * 32;
            half_t* dst = base + (wave + 8 * j) * 16 * 32;
            if (BIGP && j == PNPL - 1 && wave >= 4) { src = p.zero; dst = dummy0 + (wave - 4) * 512; }   // pixels >= 960
            if (LIGHT && j == PNPL - 1 && wave >= 6) { src = p.zero; dst = dummy0 + (wave - 6) * 512; }  // pixels >= 352
            glds16_asm(src, dst);
        }
    };
    auto issue_w = [&](int s) {                            // ring stage = taps 2s, 2s+1 of the packed stream
        half_t* st = ring0 + (s & (RING - 1)) * WSTAGE_HALFS;
        if constexpr (LIGHT && BN == 128) {          // two taps x 128 rows: both taps from every thread
            glds16_asm(wptr, st + wave * 16 * 32);
            glds16_asm(wptr + (wok ? (long)p.Np * 32 : 0), st + RROWS * 32 + wave * 16 * 32);
        } else if constexpr (LIGHT) {                // two taps x 64 rows: waves 0-3 tap 0, waves 4-7 tap 1
            glds16_asm(wptr, st + (wave >> 2) * RROWS * 32 + (wave & 3) * 16 * 32);
        } else {
            // waves 0-3: taps 0 and 2 of the step, waves 4-7: taps 1 and 3; 32-cout tiles (BIGP only: its waits do not count
            // weight DMAs) read rows 0..31 of the 64-row stage and the waves of rows 32..63 issue nothing
            if (BN == 32 && (wave & 2)) return;
            glds16_asm(wptr, st + (wave >> 2) * RROWS * 32 + (wave & 3) * 16 * 32);
            glds16_asm(wptr + (wok ? (long)p.Np * 64 : 0), st + (2 + (wave >> 2)) * RROWS * 32 + (wave & 3) * 16 * 32);
        }
        wptr += winc;
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------
    const int fx = lane & 31, fj = lane >> 5;
    // byte offsets; the second k half of a chunk (k-vectors 2,3) sits at (offset ^ 32): slot' = (2*ks + fj) ^ swz
    unsigned woffb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wco * (BN / WCO) + j * 32 + conv_wrow(fx);
        woffb[j] = (unsigned)(r * 64 + ((fj ^ ((r >> 2) & 3)) << 4));
    }
    const int qb0 = (2 * wpx) * PW + fx, qb1 = qb0 + PW;
    // ragged bottom edge: a wave whose two output rows both lie below the map only takes part in the DMA issue and the
    // barriers; its partner on the SIMD gets the matrix pipe to itself, so a tile with half of its rows valid costs about
    // half a tile (136 rows under 16-row tiles: 8.5 tiles' worth of time instead of 9)
    const bool wave_live = VSE_EDGE_SKIP == 0 || (oy0 + 2 * wpx) < p.OH;
    const char* const ring_b = reinterpret_cast<const char*>(ring0);

    float16v acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TR_STAMP(1);
    conv_stage_consts<true>(sbias, p.bias, p.zero, n0, BN, p.Np, wave, lane);                      // waves 0 .. BN/64-1
    if (p.flags & F_DOT1) conv_stage_consts<true>(sdotw, p.dotw, p.zero, n0, BN, p.Np, wave - 4, lane);   // waves 4 ..
    issue_patch(0, 0);
#pragma unroll
    for (int k = 0; k < LOOK; ++k) issue_w(k);

    int s = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const half_t* pbuf = patch0 + (cc & 1) * PATCH_HALFS;
        const bool klim1 = (p.cinp - cc * 32) <= 16;
        // taps are walked COLUMN by column (dx outer, dy inner; the compiler packs the weight stream in that order): the
        // wave's second output row under tap (dy, dx) reads the patch row its first output row reads under (dy+1, dx),
        // so inside a column every tap needs ONE new activation fragment per k half instead of two
        int tapoff = 0, dx = 0, dy = 0, tap = 0;           // tapoff = dy*PW + dx of tap
        half8 xc[2];                                       // row-1 fragments of the previous tap (k halves)
        xc[0] = xc[1] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        auto sync_issue = [&](int pr) __attribute__((always_inline)) {
            // stages s+1, s+2 may still fly (+ the next chunk's patch DMAs when they were issued 1-2 steps ago)
#ifdef VSE_TRACE
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (LIGHT) {                         // 2-stage ring as below, 3 patch DMAs per thread
                if (pr == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if constexpr (BIGP) {
                // 2-stage ring: the stage consumed now is the youngest weight DMA; only the next chunk's patch DMAs, issued
                // AFTER it in the previous step, may still fly
                if (pr == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (pr == 1 || pr == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
#if VSE_ABLATE != 1
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
#ifdef VSE_TRACE
            t_sync += __builtin_amdgcn_s_memtime() - tw0;
            if (s == 0) TR_STAMP(2);
#endif
#if VSE_ABLATE != 3
            if constexpr (BIGP || LIGHT) {                 // weights first: the patch may then outlive the next wait
                issue_w(s + LOOK);
                if (pr == 0) issue_patch(cc + 1, (cc + 1) & 1);
            } else {
                if (pr == 0) issue_patch(cc + 1, (cc + 1) & 1);
                issue_w(s + LOOK);
            }
#endif
        };
        int pr = 0;
        // FAST STEPS: all TPS taps are real and at most one of them starts a filter column (kh >= TPS).  Straight-line code,
        // software-pipelined over the 2*TPS (tap, k half) groups: the fragment reads of group g+1 are issued before the MFMAs
        // of group g into the registers of the other k half, so a wave's LDS latency hides under its own MFMAs (hipcc only
        // counts lgkmcnt when NO LDS-DMA builtin is in the kernel: glds16_asm).  The one row-0 fragment a column start needs
        // is read unconditionally at the top of the step (two reads per step) and selected per group by a block-uniform
        // compare, so there is no branch between the reads.  (Reading the next step's first activation fragments across the
        // wait + barrier was measured: +57 VGPRs from the peeled loop, 7 % slower.)
        if (PIPE && p.kh >= TPS) {
            for (; tap + TPS <= taps; ++pr, ++s) {
                sync_issue(pr);
                const unsigned wsb = (unsigned)(s & (RING - 1)) * (WSTAGE_HALFS * 2);
                const char* const pb = reinterpret_cast<const char*>(pbuf);
                int toff[TPS];
                int hstar = -1, offstar = tapoff;
#pragma unroll
                for (int h = 0; h < TPS; ++h) {
                    toff[h] = tapoff;
                    hstar = dy == 0 ? h : hstar;
                    offstar = dy == 0 ? tapoff : offstar;
                    ++tap;
                    const bool wrap = ++dy == p.kh;
                    dx += wrap ? 1 : 0;
                    tapoff = wrap ? dx : tapoff + PW;
                    dy = wrap ? 0 : dy;
                }
                if (!wave_live) continue;
                half8 x0s[2], wfb[2][TN], x1b[2];
                {
                    const unsigned q0 = (unsigned)(qb0 + offstar);
                    const unsigned a0 = (q0 << 6) + ((fj ^ ((q0 >> 2) & 3)) << 4);
                    x0s[0] = *reinterpret_cast<const half8*>(pb + a0);
                    x0s[1] = *reinterpret_cast<const half8*>(pb + (a0 ^ 32u));
                }
                unsigned a1t[TPS];
#pragma unroll
                for (int h = 0; h < TPS; ++h) {
                    const unsigned q1 = (unsigned)(qb1 + toff[h]);
                    a1t[h] = (q1 << 6) + ((fj ^ ((q1 >> 2) & 3)) << 4);
                }
                auto load = [&](int g) __attribute__((always_inline)) {
                    const int h = g >> 1, ks = g & 1;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        wfb[ks][j] = *reinterpret_cast<const half8*>(ring_b + wsb + h * (RROWS * 64) + (woffb[j] ^ (ks << 5)));
                    x1b[ks] = *reinterpret_cast<const half8*>(pb + (a1t[h] ^ (ks << 5)));
                };
                auto compute = [&](int g) __attribute__((always_inline)) {
                    const int h = g >> 1, ks = g & 1;
                    const half8 xf0 = hstar == h ? x0s[ks] : xc[ks];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfb[ks][j], xf0, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfb[ks][j], x1b[ks], acc[1][j], 0, 0, 0);
                    }
                    xc[ks] = x1b[ks];
                };
                load(0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + TN + 1, 0);      // x0s, group 0
#pragma unroll
                for (int g = 0; g < 2 * TPS; ++g) {
                    if (g + 1 < 2 * TPS) {
                        load(g + 1);
                        __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);  // reads of group g+1 ...
                    }
                    compute(g);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);      // ... then the MFMAs of group g
                }
            }
        }
        for (; pr < pairs; ++pr, ++s) {
            sync_issue(pr);
            const unsigned wsb = (unsigned)(s & (RING - 1)) * (WSTAGE_HALFS * 2);
            const char* const pb = reinterpret_cast<const char*>(pbuf);
            // GENERIC STEP (partial last step of a chunk, filters with fewer rows than taps per step, ablation builds)
#pragma unroll
            for (int h = 0; h < TPS; ++h) {
                if (h >= 1 && tap >= taps) break;          // tap count not a multiple of TPS: the appended zero-weight taps do no work
                const unsigned q0 = (unsigned)(qb0 + tapoff), q1 = (unsigned)(qb1 + tapoff);
                const unsigned a0 = (q0 << 6) + ((fj ^ ((q0 >> 2) & 3)) << 4), a1 = (q1 << 6) + ((fj ^ ((q1 >> 2) & 3)) << 4);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (!wave_live) break;
                    if (ks == 1 && klim1) break;           // channel tail <= 16: upper half of the chunk is all zeros
                    half8 wf[TN], xf[2];
#if VSE_ABLATE == 2
                    for (int j = 0; j < TN; ++j) for (int e = 0; e < 8; ++e) wf[j][e] = (half_t)(float)(q0 + e + j);
                    for (int e = 0; e < 8; ++e) { xf[0][e] = (half_t)(float)(q1 + e); xf[1][e] = (half_t)(float)(q0 - e); }
#else
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        wf[j] = *reinterpret_cast<const half8*>(ring_b + wsb + h * (RROWS * 64) + (woffb[j] ^ (ks << 5)));
                    if (dy == 0) xf[0] = *reinterpret_cast<const half8*>(pb + (a0 ^ (ks << 5)));
                    else xf[0] = xc[ks];
                    xf[1] = *reinterpret_cast<const half8*>(pb + (a1 ^ (ks << 5)));
                    xc[ks] = xf[1];
#endif
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#if VSE_ABLATE == 4
                            { acc[i][j][0] += (float)wf[j][0] * (float)xf[i][0]; asm volatile("" : "+v"(acc[i][j][0])); }
#else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
#endif
                }
                if (++tap < taps) {
                    if (++dy == p.kh) { dy = 0; tapoff = ++dx; } else { tapoff += PW; }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the zero-page dummies before LDS is released
    TR_STAMP(3);
#ifdef VSE_TRACE
    auto tr_flush = [&]() {
        if (tid == 0 && p.trace) {
            tr[4] = __builtin_amdgcn_s_memtime();
            tr[5] = t_sync;
            for (int i = 0; i < 8; ++i) p.trace[(unsigned long long)blockIdx.x * 8 + i] = tr[i];
        }
    };
#endif

    // ---- epilogue ---------------------------------------------------------------------------------------------
    if constexpr (WCO == 1) {
        if (p.flags & F_DOT1) {
            // fused 1x1 projection to one channel: this wave owns ALL couts of its pixels; lanes l and l+32 hold the
            // two halves of a pixel's couts -> one cross-half add, then lanes 0..31 store one value per pixel
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float dbias[16], dw[16];
                    conv_epilogue_consts(sbias, j * 32, lane, dbias);
                    conv_epilogue_consts(sdotw, j * 32, lane, dw);
                    part += conv_epilogue_dot(p, acc[i][j], dbias, dw);
                }
                part += __shfl_xor(part, 32);
                const int oy = oy0 + 2 * wpx + i, ox = ox0 + fx;
                if (fj == 0 && oy < p.OH && ox < p.OW) {
                    const long m = (img * p.OH + oy) * p.OW + ox;
                    const float z = vse_act(part + p.dotb, p.dotact, 0.f, 0.f);
                    if (p.dot_f32) reinterpret_cast<float*>(p.dot_out)[m * p.dot_ld] = z;
                    else reinterpret_cast<half_t*>(p.dot_out)[m * p.dot_ld] = (half_t)z;
                }
            }
#ifdef VSE_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tr_flush();
#endif
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wpx + i, ox = ox0 + fx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const long m = (img * p.OH + oy) * p.OW + ox;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bias[16];
            conv_epilogue_consts(sbias, wco * (BN / WCO) + j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, img, oy, ox, n0 + wco * (BN / WCO) + j * 32, lane);
        }
    }
#ifdef VSE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr_flush();
#endif
}

// Which variant serves a layer (mirrored by compiler.py, which packs the weight stream for it):
//   mode 2 (LIGHT, 8-row tiles, 64 or 128 couts, two blocks per CU) when the halo patch of an 8 x 32 tile fits 352 pixels
//   (3x3, 1xk) and no 1-channel projection is fused (that needs all couts of a pixel in one wave);
//   else 64 couts per tile, 16-row tiles when they fit the 960-pixel patch (mode 1 above 640 pixels), else 8-row tiles.
static int patch_light_policy() {       // VSE_PATCH_LIGHT: 0 never, 1 only layers with more than 64 couts, 2 every eligible layer
    static const int v = [] { const char* e = vse_dev_getenv("VSE_PATCH_LIGHT"); return e && e[0] ? atoi(e) : 2; }();
    return v;
}
void conv_patch_plan(int kh, int kw, int OH, int Np, int flags, int* th, int* bn, int* mode) {
    const bool fits = (8 + kh - 1) * (PTW + kw - 1) <= 352 && !(flags & (F_DOT1 | F_SRC2));
    const int pol = patch_light_policy();
    if (fits && (pol >= 2 || (pol == 1 && Np > 64))) {
        *th = 8; *bn = Np > 64 ? 128 : 64; *mode = 2;
        return;
    }
    *bn = 64;
    *th = conv_patch_th(kh, kw, OH, 64);
    *mode = (*th == 16 && (16 + kh - 1) * (PTW + kw - 1) > 640) ? 1 : 0;
    if (*mode == 1 && Np <= 32 && !(flags & F_DOT1)) *bn = 32;      // half the MFMAs and weight DMAs of a 64-cout tile
}
int conv_patch_bn(int Np) {
    (void)Np;
    return 64;
}

// Tile height: 16 rows when the halo patch fits the LDS patch buffer (960 pixels for BN = 64, else 640) and the map
// tiles at least as well as with 8 rows.
int conv_patch_th(int kh, int kw, int OH, int bn) {
    if (bn != 64) return 8;      // measured twice: a 16-row x 128-cout tile (64 px x 128 couts per wave, one block per CU, 3 or 4
                                 // taps per step, pipelined fast step, 254 VGPRs) is 7-20 % slower than LIGHT's two 8-row blocks:
                                 // a 3x3 K loop is too short to amortise an un-overlapped prologue + 128-cout epilogue
    const int cap = 960;
    if ((16 + kh - 1) * (PTW + kw - 1) > cap) return 8;
    const int pad16 = (OH + 15) / 16 * 16, pad8 = (OH + 7) / 8 * 8;
    return pad16 * 100 <= pad8 * 112 ? 16 : 8;      // accept <= 12 % extra row padding for the denser wave tile
}

int launch_conv_patch(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (p.sh != 1 || p.sw != 1 || p.kh * p.kw < 5 || (p.cinp & 7) || (p.flags & F_PIXSHUF)) return VSE_E_INVAL;
    if ((8 + p.kh - 1) * (PTW + p.kw - 1) > 640) return VSE_E_UNSUPPORTED;
    int th, bn, mode;
    conv_patch_plan(p.kh, p.kw, p.OH, p.Np, p.flags, &th, &bn, &mode);
    const bool big = mode == 1;
    p.ntn = (unsigned)((p.Np + bn - 1) / bn);
    if ((p.flags & F_DOT1) && (th != 16 || p.ntn != 1 || (p.flags & F_RES) || !p.dotw || !p.dot_out)) return VSE_E_UNSUPPORTED;
    p.tiles_h = (p.OH + th - 1) / th;
    p.tiles_w = (p.OW + PTW - 1) / PTW;
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
    p.trace = trace_dev;
#endif
    if (mode == 2 && bn == 128) hipLaunchKernelGGL((conv_patch_kernel<8, 128, 2>), grid, block, 0, st, p);
    else if (mode == 2) hipLaunchKernelGGL((conv_patch_kernel<8, 64, 2>), grid, block, 0, st, p);
    else if (big && bn == 32) hipLaunchKernelGGL((conv_patch_kernel<16, 32, 1>), grid, block, 0, st, p);
    else if (big) hipLaunchKernelGGL((conv_patch_kernel<16, 64, 1>), grid, block, 0, st, p);
    else if (th == 16) hipLaunchKernelGGL((conv_patch_kernel<16, 64, 0>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_patch_kernel<8, 64, 0>), grid, block, 0, st, p);
#ifdef VSE_TRACE
    {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
        double d[5] = {0, 0, 0, 0, 0};
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t b = 0; b < blocks; ++b) {
            const unsigned long long* t = &h[b * 8];
            d[0] += (double)(t[1] - t[0]); d[1] += (double)(t[2] - t[1]); d[2] += (double)(t[3] - t[2]);
            d[3] += (double)(t[4] - t[3]); d[4] += (double)t[5];
            if (t[0] < tmin) tmin = t[0];
            if (t[4] > tmax) tmax = t[4];
        }
        fprintf(stderr, "[patch trace] k%dx%d cin%d N%d %dx%d th%d big%d blocks %llu: per block (s_memtime ticks) setup %.0f, first wait %.0f, "
                "loop %.0f (of which wait+barrier %.0f), epilogue %.0f; kernel span %llu ticks = %.1f block-lifetimes/slot\n",
                p.kh, p.kw, p.cinp, p.Np, p.OH, p.OW, th, (int)big, blocks, d[0] / blocks, d[1] / blocks, d[2] / blocks, d[4] / blocks,
                d[3] / blocks, tmax - tmin, (double)(tmax - tmin) / ((d[0] + d[1] + d[2] + d[3]) / blocks));
    }
#endif
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
