// k x k stride-1 convolution, ONE FILTER COLUMN PER STEP over an LDS-resident input patch (gfx950 / CDNA4).
//
// Successor of conv_patch_kernel's 960-pixel variant for the tall filters (9x9 / 7x7 / 5x5 of the detector's large-kernel
// neck).  That kernel's K loop measured instruction-issue bound (8-9 non-MFMA instructions per MFMA: swizzled fragment
// addresses recomputed per tap, selects for the fragment a column start needs) with a ~1000-cycle matrix-pipe bubble at
// every step boundary (wait + barrier + LDS-DMA issue + first fragment reads) under 2048 cycles of MFMA work.  Here:
//
//   chunk  = 16 input channels = exactly one K slice of v_mfma_f32_32x32x16_f16; a patch pixel is a 32-byte LDS row.
//            The double-buffered patch shrinks to 2 x 36 KiB, which buys a FOUR-stage weight ring.
//   step   = one filter column dx of one chunk: KH taps x (2 output rows x BN/32 cout tiles) MFMAs per wave (36 for 9x9 x
//            64 couts), fully unrolled and static: the KH+1 activation fragments of the column slide through registers
//            (tap dy multiplies rows dy and dy+1), nothing is selected at run time.
//   addr   = patch rows are 48 pixels apart (>= 32 + kw - 1, and = 0 mod 16): the bank swizzle
//            slot = k-half ^ ((pixel >> 3) & 1) then depends on the column only, so every fragment read of a step is
//            ONE per-step base VGPR + an immediate (dy * 1536 bytes); weight rows likewise (dy * BN * 32 bytes).  A step
//            costs ~10 VALU instructions in total instead of several per MFMA.
//   sync   = one raw s_barrier per step.  The ring runs one stage AHEAD of the consumer: the wait in front of the barrier
//            that opens step s covers the DMAs of stage s+1, so stage s+1 is visible during step s and the first fragments
//            of step s+1 are read at the tail of step s, across the barrier — the matrix pipe has work the moment the
//            barrier releases.  Stage s+3 is issued after that barrier into the slot stage s-1 vacated.
//            vmcnt literals: every thread issues WNPL weight DMAs per step and PNPL patch DMAs (next chunk) before the
//            weights of a chunk's first step, dummies to the zero page included, so the wait is vmcnt(WNPL + PNPL) at a
//            chunk's second step and vmcnt(WNPL) everywhere else.
//   tile   = 16 x 32 output pixels x BN couts (64 | 32); 8 waves, wave w owns rows 2w, 2w+1 and all couts; waves whose
//            rows lie below the map only issue DMAs and take part in the barriers.
//   weights are packed [chunk16][dx][dy][Np][16] by the compiler (F_COL) + 3 zero stages for the look-ahead.
//   K order of the fp32 accumulation: chunk-major, then column-major taps — one K slice of 16 per MFMA, sequential.
#include <stdlib.h>
#include "conv_common.h"

#ifndef VSE_COL_XPRE
#define VSE_COL_XPRE 1    // read the next step's first fragments across the barrier (A/B: tools/ab.sh conv_col VSE_COL_XPRE)
#endif

#define CTH 16
#define CTW 32
#define CPW 48            // patch row stride in pixels
#define CRING 4

template <int KH, int BN>
__global__ __launch_bounds__(512, 2) void conv_col_kernel(const ConvParams p) {
    constexpr int TN = BN / 32;
    constexpr int CPH = CTH + KH - 1;
    constexpr int PPIX = CPH * CPW;                      // patch pixels (32 bytes each)
    constexpr int PINSTR = PPIX / 32;                    // wave DMA instructions per patch (1 KiB each)
    constexpr int PNPL = (PINSTR + 7) / 8;               // ... per thread
    constexpr int WROWS = KH * BN;                       // 32-byte weight rows per stage
    constexpr int WINSTR = WROWS / 32;
    constexpr int WNPL = (WINSTR + 7) / 8;
    constexpr int PATCH_HALFS = PPIX * 16, WSTAGE_HALFS = WROWS * 16;
    constexpr int PATCH_BYTES = PATCH_HALFS * 2, WSTAGE_BYTES = WSTAGE_HALFS * 2;
    constexpr int ROWB = CPW * 32;                       // bytes between patch rows
    static_assert(PPIX % 32 == 0 && WROWS % 32 == 0, "whole wave instructions");
    static_assert(((BN == 64 || BN == 32) && (KH == 9 || KH == 7 || KH == 5)) || (KH == 3 && BN == 128), "variants");
    __shared__ __attribute__((aligned(16))) half_t lds[2 * PATCH_HALFS + CRING * WSTAGE_HALFS + 512 + 4 * BN];   // the ONLY LDS object
    half_t* const patch0 = lds;
    half_t* const ring0 = lds + 2 * PATCH_HALFS;
    half_t* const dummy0 = ring0 + CRING * WSTAGE_HALFS;             // 1 KiB landing zone of the surplus DMAs (zeros)
    float* const sbias = reinterpret_cast<float*>(dummy0 + 512);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware bijective block order (see conv_mfma.hip)
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int nt = t % p.ntn;  t /= p.ntn;
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * CTH, ox0 = tx * CTW, n0 = nt * BN;
    if (conv_tile_right_of_sample<CTH, CTW>(p, img, oy0, ox0, n0, BN)) return;    // ragged batch: nothing to compute here

    const int kw = p.kw;
    const int nch1 = p.cinp >> 4;
    const int nchunks = (p.flags & F_HILO) ? 2 * nch1 : nch1;      // F_HILO: second pass over the patch chunks with the lo weights

    // ---- DMA source state ---------------------------------------------------------------------------------------
    // patch: instruction i covers pixels 32i .. 32i+31; lane -> pixel 32i + (lane >> 1), LDS slot lane & 1
    long poff[PNPL];
    bool pok[PNPL];
#pragma unroll
    for (int j = 0; j < PNPL; ++j) {
        const int q = 32 * (wave + 8 * j) + (lane >> 1);
        const int kh_ = (lane & 1) ^ ((q >> 3) & 1);               // logical k half stored in this lane's slot
        const int py = q / CPW, px = q - py * CPW;
        const int iy = oy0 - p.ph + py, ix = ox0 - p.pw + px;
        pok[j] = (q < PPIX) && (px < CTW + kw - 1) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        poff[j] = ((img * p.Hs + (iy >> p.inshift)) * p.Ws + (ix >> p.inshift)) * (long)p.in_ld + kh_ * 8;
    }
    // weights: instruction i covers stage rows 32i .. 32i+31 (row = dy * BN + r)
    const half_t* wptr[WNPL];
    long winc[WNPL];
#pragma unroll
    for (int j = 0; j < WNPL; ++j) {
        const int row = 32 * (wave + 8 * j) + (lane >> 1);
        const int kh_ = (lane & 1) ^ ((row >> 3) & 1);
        const int dy = row / BN, r = row - dy * BN;
        const bool ok = (row < WROWS) && (n0 + r < p.Np);
        wptr[j] = ok ? p.w + ((long)dy * p.Np + n0 + r) * 16 + kh_ * 8 : p.zero;
        winc[j] = ok ? (long)KH * p.Np * 16 : 0;
    }
    auto issue_patch = [&](int cc) {
        half_t* base = patch0 + (cc & 1) * PATCH_HALFS;
        const bool live = cc < nchunks;
#pragma unroll
        for (int j = 0; j < PNPL; ++j) {
            const int i = wave + 8 * j;
            const half_t* src = (live && pok[j]) ? p.in + poff[j] + (cc >= nch1 ? cc - nch1 : cc) * 16 : p.zero;
            half_t* dst = base + i * 512;
            if (i >= PINSTR) { src = p.zero; dst = dummy0; }
            glds16_asm(src, dst);
        }
    };
    auto issue_w = [&](int s) {
        half_t* st = ring0 + (s & (CRING - 1)) * WSTAGE_HALFS;
#pragma unroll
        for (int j = 0; j < WNPL; ++j) {
            const int i = wave + 8 * j;
            glds16_asm(wptr[j], i < WINSTR ? st + i * 512 : dummy0);
            wptr[j] += winc[j];
        }
    };

    // ---- fragment addressing (bytes) ------------------------------------------------------------------------------
    const int fx = lane & 31, fj = lane >> 5;
    unsigned woffb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = j * 32 + conv_wrow(fx);
        woffb[j] = (unsigned)(2 * PATCH_BYTES + r * 32 + ((fj ^ ((r >> 3) & 1)) << 4));
    }
    const unsigned xrow0 = (unsigned)(2 * wave * ROWB);
    auto xcol = [&](int dx, int buf) -> unsigned {       // byte address of the wave's row-0 fragment under column dx
        const unsigned c = (unsigned)(fx + dx);
        return (unsigned)buf * PATCH_BYTES + xrow0 + c * 32 + ((fj ^ ((c >> 3) & 1)) << 4);
    };
    const char* const ldsb = reinterpret_cast<const char*>(lds);
    // ragged bottom edge: see conv_patch_kernel
    const bool wave_live = (oy0 + 2 * wave) < p.OH;

    float16v acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    conv_stage_consts<true>(sbias, p.bias, p.zero, n0, BN, p.Np, wave, lane);                             // wave 0
    issue_patch(0);
    issue_w(0);
    issue_w(1);
    issue_w(2);

    wait_vm<WNPL>();                                    // constants, patch 0, stages 0 and 1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    half8 X0, X1, Wc[TN];
    // per-step base VGPRs, made opaque so that every fragment read of the step is base + immediate (hipcc otherwise adds
    // the wave-uniform stage offset with a VALU instruction per read)
    auto wbase = [&](int s_, unsigned (&wv)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wv[j] = (unsigned)(s_ & (CRING - 1)) * WSTAGE_BYTES + woffb[j];
            asm volatile("" : "+v"(wv[j]));
        }
    };
    auto preload = [&](unsigned xb, const unsigned (&wv)[TN]) __attribute__((always_inline)) {
        X0 = *reinterpret_cast<const half8*>(ldsb + xb);
        X1 = *reinterpret_cast<const half8*>(ldsb + xb + ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j) Wc[j] = *reinterpret_cast<const half8*>(ldsb + wv[j]);
    };
    unsigned wv[TN], wvn[TN];
    wbase(0, wv);
    if (wave_live) preload(xcol(0, 0), wv);

    int s = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        for (int dx = 0; dx < kw; ++dx, ++s) {
            // stage s+3 -> the slot stage s-1 vacated; a chunk's first step also starts the next chunk's patch (older than
            // the step's weights in the vmcnt queue)
            if (dx == 0) issue_patch(cc + 1);
            issue_w(s + 3);
            const bool last_col = dx + 1 == kw;
            unsigned xb = xcol(dx, cc & 1);
            unsigned xbn = last_col ? xcol(0, (cc + 1) & 1) : xcol(dx + 1, cc & 1);
            asm volatile("" : "+v"(xb), "+v"(xbn));
            wbase(s, wv);
            wbase(s + 1, wvn);
            if (wave_live) {
#if !VSE_COL_XPRE
                preload(xb, wv);
#endif
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    half8 Wn[TN], Xn, Xn0;
                    if (dy + 1 < KH) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            Wn[j] = *reinterpret_cast<const half8*>(ldsb + wv[j] + (dy + 1) * (BN * 32));
                        Xn = *reinterpret_cast<const half8*>(ldsb + xb + (dy + 2) * ROWB);
                        __builtin_amdgcn_sched_group_barrier(0x100, TN + 1, 0);
                    } else {
#if VSE_COL_XPRE
                        // the first fragments of step s+1: stage s+1 (and a next chunk's patch) became visible at the barrier
                        // that opened this step
#pragma unroll
                        for (int j = 0; j < TN; ++j) Wn[j] = *reinterpret_cast<const half8*>(ldsb + wvn[j]);
                        Xn0 = *reinterpret_cast<const half8*>(ldsb + xbn);
                        Xn = *reinterpret_cast<const half8*>(ldsb + xbn + ROWB);
                        __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#endif
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X0, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[j], X1, acc[1][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);
                    if (dy + 1 < KH) {
                        X0 = X1;
                        X1 = Xn;
#pragma unroll
                        for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
                    } else {
#if VSE_COL_XPRE
                        X0 = Xn0;
                        X1 = Xn;
#pragma unroll
                        for (int j = 0; j < TN; ++j) Wc[j] = Wn[j];
#endif
                    }
                }
            }
            // open step s+1: own DMAs of stage s+2 landed (+ the next chunk's patch unless it was issued in this step)
            if (dx == 0) wait_vm<WNPL + PNPL>();
            else wait_vm<WNPL>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the look-ahead / dummy DMAs before LDS is released

    // ---- epilogue (one wave = all couts of its 64 pixels) -----------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + fx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const long m = (img * p.OH + oy) * p.OW + ox;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, img, oy, ox, n0 + j * 32, lane);
        }
    }
}

// Eligibility (mirrored by compiler.py, which packs the weight stream for it, F_COL)
int conv_col_bn(int Np) { return Np > 32 ? 64 : 32; }
bool conv_col_ok(int kh, int kw, int sh, int sw, int cinp, int Np, int flags) {
    return sh == 1 && sw == 1 && (kh == 9 || kh == 7 || kh == 5) && kw >= 3 && CTW + kw - 1 <= CPW && (cinp & 15) == 0 && Np <= 64
           && !(flags & (F_SRC2 | F_PIXSHUF | F_DOT1));
}

#ifdef VSE_DEV_BUILD
// experiment (VSE_C3_WIDE=2, development builds only): 3x3 layers with 128 couts on this kernel's one-block-per-CU structure, all couts per block
int launch_conv_col3w(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    p.ntn = 1;
    p.tiles_h = (p.OH + CTH - 1) / CTH;
    p.tiles_w = (p.OW + CTW - 1) / CTW;
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    hipLaunchKernelGGL((conv_col_kernel<3, 128>), dim3((unsigned)blocks), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
#endif

int launch_conv_col(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (!conv_col_ok(p.kh, p.kw, p.sh, p.sw, p.cinp, p.Np, p.flags)) return VSE_E_UNSUPPORTED;
    const int bn = conv_col_bn(p.Np);
    p.ntn = (unsigned)((p.Np + bn - 1) / bn);
    p.tiles_h = (p.OH + CTH - 1) / CTH;
    p.tiles_w = (p.OW + CTW - 1) / CTW;
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w * p.ntn;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    const dim3 grid((unsigned)blocks), block(512);
    if (p.kh == 9 && bn == 64) hipLaunchKernelGGL((conv_col_kernel<9, 64>), grid, block, 0, st, p);
    else if (p.kh == 9) hipLaunchKernelGGL((conv_col_kernel<9, 32>), grid, block, 0, st, p);
    else if (p.kh == 7 && bn == 64) hipLaunchKernelGGL((conv_col_kernel<7, 64>), grid, block, 0, st, p);
    else if (p.kh == 7) hipLaunchKernelGGL((conv_col_kernel<7, 32>), grid, block, 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_col_kernel<5, 64>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_col_kernel<5, 32>), grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
