// Depthwise k x k conv FUSED INTO the 1x1 conv that follows it (F_DWPRE): the PP-LCNetV3 unit  depthwise -> pointwise  of the mobile
// models, and the  depthwise -> project  half of a MobileNetV3 unit, as ONE streaming kernel.
//
// conv_pw_kernel's observation carries over: with <= 96 input channels the activations of a 1x1 conv need no staging — the MFMA B fragment
// of lane (pixel, k-half) is 8 consecutive channels of its pixel.  Here those 8 channels are not LOADED but COMPUTED: the lane gathers the
// k x k neighbourhood of its pixel for its 8 channels (16-byte loads straight from global memory, clamped addresses, served by L1 / L2:
// HBM sees the input once), applies the depthwise filter + bias + activation in fp32 and splits the result into an fp16 hi + lo pair —
// exactly the B operands of  W_hi x_hi + W_lo x_hi + W_hi x_lo.  The depthwise output (the widest tensor of the unit at 272 x 480 ...
// 68 x 120) is never written, never read back and never rounded to 11 bits; there is no LDS tile, no halo recompute and no barrier
// beyond the one behind the weight staging (contrast csrc/chain.hip, whose LDS-resident tiles lose to the layers they fuse).
//   block = 256 threads = 4 waves x 64 output pixels; 1x1 weights hi + lo [Np][KS * 16 (+ 8 pad)] and the depthwise table
//           [k * k + 1][KS * 16] fp32 (last row = bias) staged once per block (dynamic LDS)
//   input  = NHWC fp16, optionally an fp16 hi + lo pair (p.in_lo_off: both halves are filtered)
//   output = the shared conv epilogue (bias, activation, residual, gate, pair store)
// Bound: measured VALU (round 4 counters: ~500 vector instructions per 32-pixel tile and wave, the vector pipe 65-85 % busy) — by bytes it
// would be HBM (input once + output; L1 serves k * k x the input bytes).  The taps are v_fma_mix_f32 (fp16 x fp32 + fp32, no conversions).
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

// aux blob (fp32 words, ints by bit pattern): [0] k  [1] stride  [2] pad  [3] act  [4] act_a  [5] act_b  [6] post_a  [7] post_b, then the
// depthwise table [k * k + 1][KS * 16]
#ifdef VSE_DEV_BUILD
// ablation mask of tools/ablate_dwpw.sh (development builds only): 1 no output stores, 2 only the centre tap is loaded and multiplied,
// 4 all taps loaded but only the centre one multiplied, 8 no MFMAs
__device__ int dwpw_abl_dev = 0;
#define DWPW_ABL(bit) (abl & (bit))
#else
#define DWPW_ABL(bit) false
#endif

template <int KS, int K, bool LO>
__global__ __launch_bounds__(256, (KS <= 2 ? 4 : KS <= 4 ? 3 : 2)) void conv_dwpw_kernel(const ConvParams p) {
#ifdef VSE_DEV_BUILD
    const int abl = dwpw_abl_dev;
#endif
    extern __shared__ __attribute__((aligned(16))) char dlds[];
    constexpr int ROWH = KS * 16 + 8, CP = KS * 16, K2 = K * K;
    const int ntile = (p.Np + 31) >> 5;
    half_t* swt = reinterpret_cast<half_t*>(dlds);                              // [2][ntile * 32][ROWH]
    float* sdw = reinterpret_cast<float*>(dlds + (size_t)2 * ntile * 32 * ROWH * 2);     // [K2 + 1][CP]
    float* sbias = sdw + (K2 + 1) * CP;                                          // [ntile * 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fx = lane & 31, fj = lane >> 5;
    const float* aux = p.dotw;
    {
        // every table with all of a thread's loads in flight before its first LDS write (stage_batched, common.h: the 1x1 tables of a
        // 96 -> 96 unit used to cost five dependent memory round trips per 128-pixel block, the depthwise table four more)
        const int nvec = ntile * 32 * KS * 2, rmax = p.Np - 1;
#pragma unroll
        for (int tab = 0; tab < 2; ++tab) {                  // hi, lo
            const half_t* wt = p.w + (long)tab * p.Np * CP;
            half_t* dt = swt + tab * ntile * 32 * ROWH;
            stage_batched<4>(nvec, tid,
                [&](int v) { const int r = v / (KS * 2), c = v - r * (KS * 2); return *reinterpret_cast<const half8*>(wt + (long)min(r, rmax) * CP + c * 8); },
                [&](int v, half8 x) { const int r = v / (KS * 2), c = v - r * (KS * 2);
                                      *reinterpret_cast<half8*>(dt + r * ROWH + c * 8) = r <= rmax ? x : half8{0, 0, 0, 0, 0, 0, 0, 0}; });
        }
        // (the depthwise table [K2 + 1][CP] fp32 starts 32 bytes into the aux blob: 16-byte vectors, CP % 16 == 0)
        stage_batched<4>((K2 + 1) * CP / 4, tid, [&](int v) { return *reinterpret_cast<const float4v*>(aux + 8 + 4 * v); },
                         [&](int v, float4v x) { *reinterpret_cast<float4v*>(sdw + 4 * v) = x; });
        stage_batched<1>(ntile * 32, tid, [&](int c) { return p.bias[min(c, rmax)]; }, [&](int c, float b) { sbias[c] = c <= rmax ? b : 0.f; });
    }
    const int S = p.sh, PAD = p.ph, dact = __float_as_int(aux[3]);
    const float dact_a = aux[4], dact_b = aux[5], dpost_a = aux[6], dpost_b = aux[7];
    const int lo_in = p.in_lo_off;

    __syncthreads();
    const int wr = conv_wrow(fx);
#ifndef VSE_DWPW_TPW
#define VSE_DWPW_TPW 1          // 32-pixel tiles per wave; a block = 4 waves x TPW tiles.  1 beats 2 by 2.8 % on the box-exact mobile detector (6.40 -> 6.22 ms; layer by layer +-0: tools/ab_dwpw_tpw.sh, round 5) and 4 loses 3 %: a wave that stores and exits frees its slot for one that loads — stores share the in-order vmcnt queue with the next tile's loads
#endif
    constexpr int TPW = VSE_DWPW_TPW;
    const long m0 = (long)xcd_block(blockIdx.x, gridDim.x) * (128 * TPW) + wave * (32 * TPW);       // (XCD-contiguous block order: common.h)
    // one 32-pixel MFMA tile at a time (a rolled loop: the two tiles of a wave share no registers — unrolled, hipcc kept both tiles' loads,
    // fragments and accumulators live and spilled hundreds of bytes per lane)
#pragma unroll 1
    for (int i = 0; i < TPW; ++i) {
        const long mraw = m0 + i * 32 + fx;
        const long m = mraw < p.M ? mraw : 0;
        int ow, oh;
        long n;
        conv_pix_coords(p, m, n, oh, ow);
        half8 xh[KS], xl[KS];
        const half_t* img = p.in + n * (long)p.H * p.W * p.in_ld;
        const int iy0 = oh * S - PAD, ix0 = ow * S - PAD;
        const bool inside = __builtin_amdgcn_ballot_w64(!(iy0 >= 0 && iy0 + K - 1 < p.H && ix0 >= 0 && ix0 + K - 1 < p.W)) == 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = ks * 16 + fj * 8;
            const int cl = c0 < p.cinp ? c0 : 0;                      // (a half slice behind the channels: computed on channel 0.., zeroed below)
            float a8[8];
            {
                const float4v b0 = *reinterpret_cast<const float4v*>(sdw + K2 * CP + c0), b1 = *reinterpret_cast<const float4v*>(sdw + K2 * CP + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { a8[e] = b0[e]; a8[4 + e] = b1[e]; }
            }
            // one filter ROW per iteration of a ROLLED loop, its loads issued ONE ITERATION AHEAD (xn: the next row's K vectors are in
            // flight while this row's multiply-adds run).  Unrolled, hipcc hoisted the weight reads and loads of every row to the top of
            // the slice (a 5 x 5 filter: 200+ VGPRs, over a kilobyte of scratch per lane) — scheduling barriers alone did not stop it.
            half8 xr[K], lr[K];
            auto load_row = [&](int dy, half8 (&xv)[K], half8 (&lv)[K]) {
                const int cy = min(max(iy0 + dy, 0), p.H - 1);
                const half_t* rowp = img + (long)cy * p.W * p.in_ld + cl;
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const int cx = min(max(ix0 + dx, 0), p.W - 1);                   // clamped: the load is unconditional, the tap is selected
                    xv[dx] = *reinterpret_cast<const half8*>(rowp + (long)cx * p.in_ld);
                    if constexpr (LO) lv[dx] = *reinterpret_cast<const half8*>(rowp + (long)cx * p.in_ld + lo_in);
                }
            };
            const int dy_begin = DWPW_ABL(2) ? K / 2 : 0, dy_end = DWPW_ABL(2) ? K / 2 + 1 : K;
            load_row(dy_begin, xr, lr);
#pragma unroll 1
            for (int dy = dy_begin; dy < dy_end; ++dy) {
                half8 xn[K], ln[K];
                load_row(dy + 1 < dy_end ? dy + 1 : dy, xn, ln);          // (the last iteration re-reads its own row: an L1 hit, no branch)
                const int iy = iy0 + dy;
                // A tap outside the image is zeroed on the PACKED halves (4 selects per vector) — only in tiles that touch the border:
                // `inside` is wave-uniform (every lane's 3 x 3 window lies in the image: ~85 % of the tiles at 272 x 480), and the
                // multiply-adds take the fp16 values directly (vse_fma_h8: v_fma_mix_f32, no conversions): 8 VALU instructions per tap
                // and 8 channels instead of 20 — the kernel is VALU-bound (counters: the vector pipe 65-85 % busy)
                auto taps = [&](auto sel) {
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) {
                        if (DWPW_ABL(4) && dy != K / 2) {
                            asm volatile("" :: "v"(xr[dx]));
                            if constexpr (LO) asm volatile("" :: "v"(lr[dx]));
                            continue;
                        }
                        const int ix = ix0 + dx;
                        // (unsigned compares, bitwise and: a short-circuit && compiles to branches, which split the block)
                        const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                        const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                        const float4v w0 = *reinterpret_cast<const float4v*>(sdw + (dy * K + dx) * CP + c0);
                        const float4v w1 = *reinterpret_cast<const float4v*>(sdw + (dy * K + dx) * CP + c0 + 4);
                        vse_fma_h8(a8, (!decltype(sel)::value || ok) ? xr[dx] : z8, w0, w1);
                        if constexpr (LO) vse_fma_h8(a8, (!decltype(sel)::value || ok) ? lr[dx] : z8, w0, w1);
                    }
                };
                if (inside) taps(std::false_type{});
                else taps(std::true_type{});
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    xr[dx] = xn[dx];
                    if constexpr (LO) lr[dx] = ln[dx];
                }
            }
            vse_act_n<8>(a8, dact, dact_a, dact_b);
            // (uniform branches: the affine behind the activation is the identity for all but a handful of layers, and only a channel
            // count that is not a multiple of 16 has a half slice behind its channels — the kernel is VALU-bound)
            if (dpost_a != 1.f || dpost_b != 0.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = a8[e] * dpost_a + dpost_b;
            }
            if ((p.cinp & 15) && c0 >= p.cinp) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[ks][e] = (half_t)a8[e];
                xl[ks][e] = (half_t)(a8[e] - (float)xh[ks][e]);
            }
        }
        for (int j = 0; j < ntile; ++j) {
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 wh = *reinterpret_cast<const half8*>(swt + (j * 32 + wr) * ROWH + ks * 16 + fj * 8);
                const half8 wl = *reinterpret_cast<const half8*>(swt + ((ntile + j) * 32 + wr) * ROWH + ks * 16 + fj * 8);
                if (DWPW_ABL(8)) { acc[0] += (float)xh[ks][0] + (float)xl[ks][1] + (float)wh[0] + (float)wl[1]; continue; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], acc, 0, 0, 0);
            }
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            if (mraw < p.M && (!DWPW_ABL(1) || acc[3] == 1234.5678f)) conv_epilogue_tile(p, acc, bias, mraw, n, oh, ow, j * 32, lane);
        }
    }
}

// 5 x 5 filters stay on two launches (25 taps per lane: 0.21 against 0.09 + 0.04 ms, and the unrolled form spills): the 5 x 5 instantiations
// exist in development builds only (VSE_DEV_BUILD, compiler.py VSE_DWPW_K=3,5)
#ifdef VSE_DEV_BUILD
#define DWPW_K5 1
#else
#define DWPW_K5 0
#endif
bool conv_dwpw_ok(int k, int s, int cinp, int Np, int flags) {
    return (k == 3 || (DWPW_K5 && k == 5)) && (s == 1 || s == 2) && (cinp & 7) == 0 && cinp <= 96 && Np <= 192 && (flags & F_HILO)
           && !(flags & (F_SRC2 | F_DOT1 | F_PATCH | F_COL | F_PIXSHUF | F_IMGW | F_STEM));
}

template <int KS, int K, bool LO>
static int launch_dwpw_t(const ConvParams& p, hipStream_t st) {
    const int ntile = (p.Np + 31) >> 5;
    const size_t lds = (size_t)2 * ntile * 32 * (KS * 16 + 8) * 2 + (size_t)(K * K + 1) * KS * 16 * 4 + (size_t)ntile * 32 * 4;
    static VseDevOnce attr_once;          // (per device, thread-safe: common.h)
    if (!vse_dev_once(attr_once, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dwpw_kernel<KS, K, LO>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
        }))
        return VSE_E_HIP;
    const unsigned long long blocks = (unsigned long long)((p.M + 128 * VSE_DWPW_TPW - 1) / (128 * VSE_DWPW_TPW));
    if (blocks == 0 || p.M >= 0x7fffffffl || lds > 128 * 1024) return VSE_E_INVAL;          // (32-bit pixel arithmetic: conv_pix_coords)
#ifdef VSE_DEV_BUILD
    static int abl_set = -1;
    if (abl_set < 0) {
        abl_set = getenv("VSE_DWPW_ABL") ? atoi(getenv("VSE_DWPW_ABL")) : 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(dwpw_abl_dev), &abl_set, sizeof(int)) != hipSuccess) return VSE_E_HIP;
    }
#endif
    hipLaunchKernelGGL((conv_dwpw_kernel<KS, K, LO>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}

// p.kh / p.sh / p.ph describe the DEPTHWISE conv (the 1x1 conv has no geometry); p.dotw = the aux blob; p.in_lo_off = the input's pair offset
int launch_conv_dwpw(const ConvParams& p, hipStream_t st) {
    if (!conv_dwpw_ok(p.kh, p.sh, p.cinp, p.Np, p.flags) || p.kh != p.kw || p.sh != p.sw || p.ph != p.pw || !p.dotw) return VSE_E_UNSUPPORTED;
    const int ks = (p.cinp + 15) / 16;
#if DWPW_K5
#define DWPW(KS_) (p.in_lo_off ? (p.kh == 3 ? launch_dwpw_t<KS_, 3, true>(p, st) : launch_dwpw_t<KS_, 5, true>(p, st)) \
                               : (p.kh == 3 ? launch_dwpw_t<KS_, 3, false>(p, st) : launch_dwpw_t<KS_, 5, false>(p, st)))
#else
#define DWPW(KS_) (p.in_lo_off ? launch_dwpw_t<KS_, 3, true>(p, st) : launch_dwpw_t<KS_, 3, false>(p, st))
#endif
    switch (ks) {
        case 1: return DWPW(1);
        case 2: return DWPW(2);
        case 3: return DWPW(3);
        case 4: return DWPW(4);
        case 5: return DWPW(5);
        case 6: return DWPW(6);
        default: return VSE_E_UNSUPPORTED;
    }
#undef DWPW
}
