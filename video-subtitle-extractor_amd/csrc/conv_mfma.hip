// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950), NHWC fp16 in, fp32 accumulate.
//
// GEMM view:  D[cout][pixel] = sum_k  Wt[cout][k] * X[pixel][k],   k = (tap_y, tap_x, cin)
//   * "A" MFMA operand = weights (rows = cout), "B" operand = gathered activations (cols = pixel), so each
//     lane ends up holding 4 *consecutive output channels* of one pixel per accumulator quad -> 8-byte
//     NHWC stores, per-channel bias as a float4, residual loads of the same shape.
//   * v_mfma_f32_32x32x16_f16: lane l supplies row/col (l&31) and k-slice 8*(l>>5)..+7 of both operands.
//   * block = 256 threads = 4 waves arranged WM x WN over a BM(pixels) x BN(couts) tile, BK = 64 per step
//     (16 MFMAs per wave between barriers), one LDS stage with 144-byte rows (conflict-free b128 reads and
//     writes); the next K tile is prefetched into registers under the MFMAs and written after the barrier.
//   * activations are gathered as one 16-byte vector per (pixel, tap, 8-channel group): cin is padded to a
//     multiple of 8 by the compiler, so a vector never straddles taps; image borders, K padding and the
//     M tail are predicated to zero.  A nearest-upsampled input (FPN) is gathered with (y>>s, x>>s).
//   * weights arrive pre-tiled [K/64][Np][64] so a BN x 64 tile is one contiguous, fully coalesced chunk.
//   * epilogue: + bias (BN folded) -> activation -> scalar affine -> (+ residual, optionally upsampled)
//     -> activation2 -> fp16 (or fp32) store; 2x2/stride-2 transposed conv = same GEMM with a
//     pixel-shuffle store.

#include <stdlib.h>
#include "conv_common.h"

#define BK 32          // K elements per pipeline stage
#define STAGES 3       // LDS ring: tile k is consumed while tiles k+1 and k+2 are in flight
#define ROWB 64        // bytes per LDS row (BK halfs, no padding: LDS-DMA writes are lane-linear)

template <int BM, int BN, int WM, int WN, bool UPS>
__global__ __launch_bounds__(256, (BM + (BN < 64 ? 64 : BN)) <= 256 ? 3 : 2) void conv_mfma_kernel(const ConvParams p) {   // 3 / 2 blocks per CU by LDS
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BNR = BN < 64 ? 64 : BN;      // weight rows staged per tile (>= one wave instruction per wave)
    constexpr int NA = BM / 64;                 // LDS-DMA instructions per wave per stage, activations
    constexpr int NB = BNR / 64;                // ... weights
    constexpr int LPT = NA + NB;                // VMEM ops per thread per stage (vmcnt bookkeeping)
    constexpr int STAGE_HALFS = (BM + BNR) * BK;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile shape");

    __shared__ __attribute__((aligned(16))) half_t lds[STAGES * STAGE_HALFS + BN * 2];   // the ONLY LDS object: ring + bias
    float* const sbias = reinterpret_cast<float*>(lds + STAGES * STAGE_HALFS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (each XCD has a private 4 MiB L2).  Give every
    // XCD one CONTIGUOUS range of logical tiles (bijective for any tile count) and walk the N tiles of one pixel
    // tile first, so the kh x kw halo rows shared by neighbouring pixel tiles and the re-read of the activation
    // tile by the other cout tiles hit the same L2 instead of being fetched once per XCD.  Speed only: any
    // placement computes the same result.
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned mtile = logical / p.ntn, ntile = logical - mtile * p.ntn;
    const long m0 = (long)mtile * BM;
    const int n0 = ntile * BN;

    // ---- per-thread gather state ---------------------------------------------------------------------------
    // wave instruction j of this wave covers tile rows (j*4 + wave)*16 .. +15; lane l -> row +(l>>2), physical slot
    // l&3, i.e. logical k-vector kv = (l&3) ^ ((l>>4)&3)  (row>>2 & 3 == l>>4 & 3 because the 16-row base is 16-aligned)
    const int kv = (lane & 3) ^ ((lane >> 4) & 3);
    const int rsub = lane >> 2;
    // Gather addressing is incremental: per row one base pointer (tap (0,0), channel 0) and two validity bitmasks
    // (bit dy / bit dx set when that tap row / column is inside the image); per K step the thread computes ONE tap
    // offset shared by its rows.  The nearest-upsampled input (UPS) keeps the explicit (y>>s, x>>s) form.
    int ih0[NA], iw0[NA];
    long pbase[NA];
    const half_t* rowptr[NA];
    unsigned rmask[NA], cmask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const long m = m0 + (j * 4 + wave) * 16 + rsub;
        ih0[j] = 0; iw0[j] = 0; pbase[j] = 0; rowptr[j] = p.zero; rmask[j] = 0; cmask[j] = 0;
        if (m < p.M) {
            const int ow = (int)(m % p.OW);
            const long t = m / p.OW;
            const int oh = (int)(t % p.OH);
            const long n = t / p.OH;
            ih0[j] = oh * p.sh - p.ph;
            iw0[j] = ow * p.sw - p.pw;
            pbase[j] = n * p.Hs * p.Ws;
            if constexpr (!UPS) {
                rowptr[j] = p.in + (pbase[j] + (long)ih0[j] * p.Ws + iw0[j]) * p.in_ld;
                for (int d = 0; d < p.kh; ++d) rmask[j] |= (unsigned)(ih0[j] + d >= 0 && ih0[j] + d < p.H) << d;
                for (int d = 0; d < p.kw; ++d) cmask[j] |= (unsigned)(iw0[j] + d >= 0 && iw0[j] + d < p.W) << d;
            }
        } else if constexpr (UPS) {
            ih0[j] = -(1 << 28);
        }
    }
    int kc = kv * 8, dy = 0, dx = 0;
    while (kc >= p.cinp) {
        kc -= p.cinp;
        if (++dx == p.kw) { dx = 0; ++dy; }
    }
    const half_t* wrow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + (j * 4 + wave) * 16 + rsub;
        const bool ok = ((j * 4 + wave) * 16 + rsub < BN) && (n < p.Np);
        wrow[j] = ok ? p.w + (long)n * 64 + kv * 8 : nullptr;        // weights are tiled [Kp/64][Np][64]
    }
    const long wstep = (long)p.Np * 64;

    // issue the LDS-DMAs of K tile `kt` into ring slot `st` (always exactly LPT VMEM ops per thread)
    auto issue = [&](int kt, int st) {
        half_t* base = lds + st * STAGE_HALFS;
        if constexpr (UPS) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int ih = ih0[j] + dy, iw = iw0[j] + dx;
                const bool ok = (dy < p.kh) && (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.W);
                const half_t* src = p.zero;
                if (ok) src = p.in + (pbase[j] + (long)(ih >> p.inshift) * p.Ws + (iw >> p.inshift)) * p.in_ld + kc;
                glds16(src, base + (j * 4 + wave) * 16 * BK);
            }
        } else {
            const long toff = (long)(dy * p.Ws + dx) * p.in_ld + kc;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const bool ok = (rmask[j] >> dy) & (cmask[j] >> dx) & 1u;     // bits >= kh / kw are never set
                glds16(ok ? rowptr[j] + toff : p.zero, base + (j * 4 + wave) * 16 * BK);
            }
        }
        const long woff = (long)(kt >> 1) * wstep + (kt & 1) * 32;
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(wrow[j] ? wrow[j] + woff : p.zero, base + BM * BK + (j * 4 + wave) * 16 * BK);
        kc += BK;
        while (kc >= p.cinp) {
            kc -= p.cinp;
            if (++dx == p.kw) { dx = 0; ++dy; }
        }
        if (kt + 1 == p.nkh) {                   // F_HILO: the walk over the taps starts again for the lo weight tiles
            kc = kv * 8; dy = 0; dx = 0;
            while (kc >= p.cinp) {
                kc -= p.cinp;
                if (++dx == p.kw) { dx = 0; ++dy; }
            }
        }
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    conv_stage_consts(sbias, p.bias, p.zero, n0, BN, p.Np, wave, lane);
    issue(0, 0);
    if (p.nk > 1) issue(1, 1);

    const int frow = lane & 31;
    const int fj = lane >> 5;                      // logical k-vector within a 16-wide k sub-step
    // fragment byte offsets inside a stage (row*64 + swizzled slot*16), fixed per thread
    int xoff[TM][2], woff[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wm * WTM + i * 32 + frow;
            xoff[i][ks] = r * BK + (((ks * 2 + fj) ^ ((r >> 2) & 3)) << 3);
        }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wn * WTN + j * 32 + conv_wrow(frow);
            woff[j][ks] = BM * BK + r * BK + (((ks * 2 + fj) ^ ((r >> 2) & 3)) << 3);
        }

    int st = 0;
    for (int kt = 0; kt < p.nk; ++kt) {
        // tile kt has landed once at most the LPT ops of tile kt+1 are still outstanding (vmcnt counts in order)
        if (kt + 1 < p.nk) {
            if constexpr (LPT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (LPT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if constexpr (LPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (LPT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();              // every thread's part of tile kt is in LDS; compute(kt-1) is over
        asm volatile("" ::: "memory");
        if (kt + 2 < p.nk) {
            int s2 = st + 2;
            if (s2 >= STAGES) s2 -= STAGES;
            issue(kt + 2, s2);                     // refill the slot compute(kt-1) just released
        }
        const half_t* base = lds + st * STAGE_HALFS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 wf[TN], xf[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8*>(base + woff[j][ks]);
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8*>(base + xoff[i][ks]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (++st == STAGES) st = 0;
    }

    // ---- epilogue ----------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * WTM + i * 32 + (lane & 31);
        if (m >= p.M) continue;
        const int ow = (int)(m % p.OW);
        const long t = m / p.OW;
        const int oh = (int)(t % p.OH);
        const long n = t / p.OH;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bias[16];
            conv_epilogue_consts(sbias, wn * WTN + j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, n, oh, ow, n0 + wn * WTN + j * 32, lane);
        }
    }
}

int conv_tile_bn(int Np) {
    // tile selection: minimise padded-N waste, prefer the widest tile on ties
    auto padded = [&](int bn) { return ((Np + bn - 1) / bn) * bn; };
    if (Np <= 32) return 32;
    if (padded(64) < padded(128)) return 64;
    return 128;
}

int launch_conv(const ConvArgs& a, hipStream_t st) {
    ConvParams p;
    p.trace = nullptr;
    p.in = reinterpret_cast<const half_t*>(a.in.ptr);
    p.w = a.w;
    p.bias = a.bias;
    p.res = reinterpret_cast<const half_t*>(a.res.ptr);
    p.out = a.out.ptr;
    p.Hs = a.in.h;
    p.Ws = a.in.w;
    p.H = a.in.h << a.inshift;
    p.W = a.in.w << a.inshift;
    p.in_ld = a.in.ld;
    p.cinp = a.cinp;
    p.inshift = a.inshift;
    p.kh = a.kh; p.kw = a.kw; p.sh = a.sh; p.sw = a.sw; p.ph = a.ph; p.pw = a.pw;
    p.OH = (p.H + 2 * a.ph - a.kh) / a.sh + 1;
    p.OW = (p.W + 2 * a.pw - a.kw) / a.sw + 1;
    p.M = (long)a.in.n * p.OH * p.OW;
    p.Np = a.Np;
    p.nk = a.Kp / BK;
    p.nkh = p.nk;
    if (a.flags & F_HILO) p.nk *= 2;          // second pass over the same activations with the lo weight tiles
    p.zero = a.zero;
    p.out_ld = a.out.ld;
    p.out_f32 = (a.flags & F_OUT_F32) ? 1 : 0;
    p.res_ld = a.res.ld;
    p.resshift = a.resshift;
    p.res_hs = a.res.h;
    p.res_ws = a.res.w;
    p.act = a.act; p.act2 = a.act2;
    p.act_a = a.act_a; p.act_b = a.act_b; p.post_a = a.post_a; p.post_b = a.post_b;
    p.flags = a.flags;
    p.coutp = (a.flags & F_PIXSHUF) ? a.Np / 4 : a.Np;
    if (a.flags & F_SRC2) {
        if (!a.in2.ptr || a.in2.esize != 2 || (a.in2.ld & 7) || a.in.c + a.in2.c != a.cinp) return VSE_E_INVAL;
        if ((a.in2.h << a.in2shift) != p.H || (a.in2.w << a.in2shift) != p.W) return VSE_E_INVAL;
    } else if (a.in.c != a.cinp) {
        return VSE_E_INVAL;
    }
    // (F_UP2HEAD reads ONE channel of in0 at pixel stride ld: the dense map of an F_TAIL2 producer has ld = 1)
    if (a.in.esize != 2 || ((a.in.ld & 7) && !((a.flags & F_UP2HEAD) && a.in.ld == 1)) || (a.cinp & 7)) return VSE_E_INVAL;
    if ((a.flags & F_RES) && (a.res.esize != 2 || (a.res.ld & 3))) return VSE_E_INVAL;
    if ((!(a.flags & (F_DOT1 | F_ONECH)) && (a.out.ld & 3)) || (a.Np & 7)) return VSE_E_INVAL;
    if ((a.flags & F_ONECH) && (!(a.flags & F_PIXSHUF) || !(a.flags & F_OUT_F32) || a.Np != 32 || a.out.ld != 1 || a.out.esize != 4 || (a.flags & F_RES)))
        return VSE_E_INVAL;
    // sanity on the output view: [n, OH(*2), OW(*2)]
    const int mul = (a.flags & F_PIXSHUF) ? 2 : 1;
    if (a.out.h != p.OH * mul || a.out.w != p.OW * mul || a.out.n != a.in.n) return VSE_E_INVAL;

    p.vec16 = ((reinterpret_cast<uintptr_t>(a.out.ptr) & 15) == 0 && ((long)a.out.ld * a.out.esize) % 16 == 0 &&
               (!(a.flags & F_RES) || ((reinterpret_cast<uintptr_t>(a.res.ptr) & 15) == 0 && (a.res.ld & 7) == 0))) ? 1 : 0;
    p.dotw = a.dotw; p.dotb = a.dotb; p.dotact = a.dotact;
    p.dot_out = a.dot_out.ptr; p.dot_f32 = a.dot_out.esize == 4; p.dot_ld = a.dot_out.ld;
    p.in2 = reinterpret_cast<const half_t*>(a.in2.ptr); p.in2_ld = a.in2.ld; p.in2_shift = a.in2shift;
    p.in2_hs = a.in2.h; p.in2_ws = a.in2.w; p.nv0 = a.in.c >> 3;
    p.wimg_stride = 0; p.hw_img = 0; p.tiles_img = 0;
    p.wl_out = a.wl_out;
    p.lo_off = a.lo_off;
    p.res_lo_off = (a.flags & F_RES) ? a.res_lo_off : 0;
    if (p.res_lo_off && (!p.vec16 || a.resshift || (p.res_lo_off & 7))) return VSE_E_INVAL;
    p.in_lo_off = (a.flags & F_DWPRE) ? a.in_lo_off : 0;
    if (a.lo_off && (!p.vec16 || (a.flags & (F_OUT_F32 | F_ONECH | F_DOT1 | F_UP2HEAD)) || (a.lo_off & 7) || a.out.ld < a.lo_off + a.Np / ((a.flags & F_PIXSHUF) ? 4 : 1)))
        return VSE_E_INVAL;
    p.ogate = nullptr; p.ogate_ld = 0;
    if (a.flags & F_OGATE) {
        // (in2 carries the gate: not combined with the other users of that slot)
        if ((a.flags & (F_SRC2 | F_IMGW | F_PIXSHUF | F_DOT1 | F_UP2HEAD)) || !a.in2.ptr || a.in2.esize != 2 || a.in2.n != a.in.n || a.in2.h != 1 || a.in2.w != 1
            || a.in2.c < a.Np || (a.in2.ld & 7) || (reinterpret_cast<uintptr_t>(a.in2.ptr) & 15)) return VSE_E_INVAL;
        p.ogate = reinterpret_cast<const half_t*>(a.in2.ptr);
        p.ogate_ld = a.in2.ld;
    }
    p.u8src = a.u8src; p.u8_h = a.u8_h; p.u8_w = a.u8_w; p.u8_pitch = a.u8_pitch; p.u8_fstride = a.u8_fstride;
    if ((a.flags & F_U8SRC) && (!(a.flags & F_STEM) || !a.u8src || a.u8_h <= 0 || a.u8_w <= 0)) return VSE_E_INVAL;
    if (a.wl_out && (a.flags & (F_DOT1 | F_SRC2 | F_UP2HEAD | F_PIXSHUF))) return VSE_E_UNSUPPORTED;   // no per-sample width in these forms
    if (a.flags & F_IMGW) {
        // per-image weights (an SE gate folded into a 1x1 consumer): conv_gemm_kernel only
        if (a.kh != 1 || a.kw != 1 || (a.flags & (F_SRC2 | F_DOT1 | F_PATCH | F_COL | F_PW | F_HILO | F_PIXSHUF))) return VSE_E_UNSUPPORTED;
        p.wimg_stride = (long)a.Kp * a.Np;
        p.hw_img = p.OH * p.OW;
        return launch_conv_gemm(p, a.Kp, st);
    }
    if (a.flags & F_DWPRE) {
        if (!(a.flags & F_PW) || a.inshift || !a.dotw || (p.in_lo_off && ((p.in_lo_off & 7) || a.in.ld < p.in_lo_off + a.cinp))) return VSE_E_INVAL;
        return launch_conv_dwpw(p, st);
    }
    if ((a.flags & (F_DOT1 | F_SRC2)) && !(a.flags & (F_PATCH | F_COL))) return VSE_E_UNSUPPORTED;
    if (!a.zero) return VSE_E_INVAL;
    if (a.flags & F_UP2HEAD) return launch_conv_head_up2(p, a.in.n, st);
    if (a.flags & F_STEM) return launch_conv_stem(p, a.in.n, st);
    if (a.flags & F_PW) return launch_conv_pw(p, st);
    if (a.flags & F_COL) return (p.kh == 3 && p.kw == 3) ? launch_conv_c3(p, a.in.n, st) : launch_conv_col(p, a.in.n, st);
    if (a.flags & F_PATCH) return launch_conv_patch(p, a.in.n, st);
    if (a.Kp % 64) return VSE_E_INVAL;
    if (conv_smallk_ok(p)) {                             // a small 1x1 problem: one wave per 32 x 32 tile, all loads in flight (conv_smallm.hip)
        p.nkh = a.Kp / ((a.flags & F_WK32) ? 32 : 64);
        return launch_conv_smallm(p, st);
    }
    static const bool use_gemm = [] { const char* e = vse_dev_getenv("VSE_CONV_GEMM"); return !(e && e[0] == '0'); }();
    if (use_gemm) {
        const int rc = launch_conv_gemm(p, a.Kp, st);
        if (rc != VSE_E_UNSUPPORTED) return rc;
    }
    if (a.flags & F_WK32) return VSE_E_UNSUPPORTED;     // 32-deep weight tiles are read by conv_gemm_kernel only
    const int bn = conv_tile_bn(a.Np);
    dim3 block(256);
    const int bm = bn == 128 ? 128 : 256;
    p.ntn = (unsigned)((a.Np + bn - 1) / bn);
    const unsigned long long tiles = (unsigned long long)((p.M + bm - 1) / bm) * p.ntn;
    if (tiles == 0 || tiles > 0x7fffffffull) return VSE_E_INVAL;
    dim3 grid((unsigned)tiles);
    if (a.inshift) {
        if (bn == 128) hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 2, 2, true>), grid, block, 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((conv_mfma_kernel<256, 64, 4, 1, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<256, 32, 4, 1, true>), grid, block, 0, st, p);
    } else {
        if (bn == 128) hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 2, 2, false>), grid, block, 0, st, p);
        else if (bn == 64) hipLaunchKernelGGL((conv_mfma_kernel<256, 64, 4, 1, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<256, 32, 4, 1, false>), grid, block, 0, st, p);
    }
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
