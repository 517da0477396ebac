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

// ---- ROW-STREAMING form (round 6): a wave walks DOWN a strip of 32 output columns --------------------------------------------------
// conv_dwpw_kernel gives every 32-pixel tile its own wave and every 128 pixels their own block: per output pixel and 8 channels it issues
// 9 (x 2 for a pair) 16-byte gathers, and every block pays the table staging + barrier in front of ~2 us of work (ablations, round 4:
// of 0.62 ms on the first pair unit, two of three filter rows' LOADS cost 0.12, the stores 0.24, and 0.25 is fixed cost).  Here a wave owns
// 32 columns x RS consecutive output rows of one image: a new output row needs ONE new input row (stride 1; two at stride 2) — 3 gathers
// instead of 9 — and the block's prologue is paid once per 4 x RS x 32 pixels.  The filter rows of an input row are applied to the
// (up to three) output rows it belongs to, each output row keeping its own fp32 accumulator: an accumulator still receives
// bias, then filter row 0 (dx 0 hi, lo, dx 1 ...), row 1, row 2 IN THE ORDER OF conv_dwpw_kernel, the activation / hi + lo split / MFMA
// order / epilogue are the same code — every output bit is the tile form's (development build: `tools/ab_env_digest.py VSE_DWPW_ROWS 0 1`,
// identical digests of both mobile detectors; the product tests hold it to the emulator and to the real detector's boxes).
//   stride 1: state = A (output row r: filter rows 0, 1 applied), B (row r + 1: filter row 0); input row r + 1 arrives:
//             A += row 2 -> finished;  A(reused) = bias + row 0 (output row r + 2);  B += row 1;  roles swap.
//   stride 2: state = A (output row r: filter row 0 applied); input row 2r: A += row 1; input row 2r + 1: A += row 2 -> finished;
//             A = bias + row 0 (output row r + 1).
// The next (input row, 16-channel slice) item's gathers are issued before the current item's multiply-adds.
template <int KS, bool LO, int S>
__global__ __launch_bounds__(256, 2) void conv_dwpw_rows_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char dlds[];
    constexpr int K = 3, ROWH = KS * 16 + 8, CP = KS * 16, K2 = 9;
    const int ntile = (p.Np + 31) >> 5;
    half_t* swt = reinterpret_cast<half_t*>(dlds);                              // [2][ntile * 32][ROWH]
    float* sdw = reinterpret_cast<float*>(dlds + (size_t)2 * ntile * 32 * ROWH * 2);     // [K2 + 1][CP]
    float* sbias = sdw + (K2 + 1) * CP;                                          // [ntile * 32]
    const half_t* const swt_ = swt;
    const float* const sdw_ = sdw;
    // The tables do not change from row to row, so hipcc may hoist the weight reads out of the row loop.  Where that fits the register
    // budget it is what makes this form fast — the 18 broadcast ds_read_b128 per row and slice weigh as much on the CU's one LDS pipe as
    // the 144 multiply-adds on a SIMD (pair units of <= 3 slices: 0.65 -> 0.50, 0.52 -> 0.45, 0.40 -> 0.30 ms); wider units would spill,
    // there the table pointers are laundered per use (`LAUNDER`) and the reads stay in the loop.
    constexpr bool LAUNDER = KS > 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fx = lane & 31, fj = lane >> 5;
    const float* aux = p.dotw;
    {
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
        stage_batched<4>((K2 + 1) * CP / 4, tid, [&](int v) { return *reinterpret_cast<const float4v*>(aux + 8 + 4 * v); },
                         [&](int v, float4v x) { *reinterpret_cast<float4v*>(sdw + 4 * v) = x; });
        stage_batched<1>(ntile * 32, tid, [&](int c) { return p.bias[min(c, rmax)]; }, [&](int c, float b) { sbias[c] = c <= rmax ? b : 0.f; });
    }
    const int dact = __float_as_int(aux[3]);
    const float dact_a = aux[4], dact_b = aux[5], dpost_a = aux[6], dpost_b = aux[7];
    const int lo_in = p.in_lo_off;
    __syncthreads();                                          // (the only barrier: a wave without a strip may leave behind it)
    const int wr = conv_wrow(fx);

    // this wave's strip: strips along a row fastest, then row segments, then images (XCD-contiguous block order: common.h)
    const int nstrips = p.tiles_w, nseg = p.tiles_h, RS = (int)p.ntiles;
    const long widx = (long)xcd_block(blockIdx.x, gridDim.x) * 4 + wave;
    const long nimg = p.M / ((long)p.OH * p.OW);
    if (widx >= nimg * nseg * nstrips) return;
    const int sx = (int)(widx % nstrips);
    const long t_ = widx / nstrips;
    const int sy = (int)(t_ % nseg);
    const long n = t_ / nseg;
    const int r0 = sy * RS, r1 = min(r0 + RS, p.OH);
    const int ow = sx * 32 + fx;
    const int ix0 = ow * S - 1;
    int cxo[K];
    bool okx[K];
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
        cxo[dx] = min(max(ix0 + dx, 0), p.W - 1) * p.in_ld;
        okx[dx] = (unsigned)(ix0 + dx) < (unsigned)p.W;
    }
    const bool colsafe = __builtin_amdgcn_ballot_w64(!(okx[0] & okx[2])) == 0;      // every lane's three columns lie in the image
    const half_t* img = p.in + n * (long)p.H * p.W * p.in_ld;

    // one item = the three column taps (hi, lo) of input row iy for slice ks, gathered from clamped addresses
    auto load_item = [&](int iy, int ks, half8 (&xv)[K], half8 (&lv)[K]) __attribute__((always_inline)) {
        const int c0 = ks * 16 + fj * 8;
        const int cl = c0 < p.cinp ? c0 : 0;
        const int cy = min(max(iy, 0), p.H - 1);
        const half_t* rowp = img + (long)cy * p.W * p.in_ld + cl;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            xv[dx] = *reinterpret_cast<const half8*>(rowp + cxo[dx]);
            if constexpr (LO) lv[dx] = *reinterpret_cast<const half8*>(rowp + cxo[dx] + lo_in);
        }
    };
    // filter row dy of slice ks applied to one accumulator, in conv_dwpw_kernel's order (dx ascending, hi then lo)
    auto taps = [&](float (&a8)[8], int dy, int ks, const half8 (&xv)[K], const half8 (&lv)[K], bool rowok) __attribute__((always_inline)) {
        const int c0 = ks * 16 + fj * 8;
        // (the tables do not change from row to row: un-laundered, hipcc hoists every tap's weight read out of the row loop — 72 VGPRs per
        // slice — and spills)
        const float* sdw = sdw_;
        if constexpr (LAUNDER) asm volatile("" : "+v"(sdw));
        auto body = [&](auto sel) __attribute__((always_inline)) {
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const bool ok = rowok & okx[dx];
                const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                const float4v w0 = *reinterpret_cast<const float4v*>(sdw + (dy * K + dx) * CP + c0);
                const float4v w1 = *reinterpret_cast<const float4v*>(sdw + (dy * K + dx) * CP + c0 + 4);
                vse_fma_h8(a8, (!decltype(sel)::value || ok) ? xv[dx] : z8, w0, w1);
                if constexpr (LO) vse_fma_h8(a8, (!decltype(sel)::value || ok) ? lv[dx] : z8, w0, w1);
            }
        };
        if (colsafe && rowok) body(std::false_type{});
        else body(std::true_type{});
    };
    auto set_bias = [&](float (&a8)[8], int ks) __attribute__((always_inline)) {
        const int c0 = ks * 16 + fj * 8;
        const float* sdw = sdw_;
        if constexpr (LAUNDER) asm volatile("" : "+v"(sdw));
        const float4v b0 = *reinterpret_cast<const float4v*>(sdw + K2 * CP + c0), b1 = *reinterpret_cast<const float4v*>(sdw + K2 * CP + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a8[e] = b0[e]; a8[4 + e] = b1[e]; }
    };
    half8 xh[KS], xl[KS];
    auto finish = [&](float (&a8)[8], int ks) __attribute__((always_inline)) {              // activation, affine, channel tail, fp16 hi + lo split: conv_dwpw_kernel's
        const int c0 = ks * 16 + fj * 8;
        vse_act_n<8>(a8, dact, dact_a, dact_b);
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
    };
    auto emit_row = [&](int oh) __attribute__((always_inline)) {                            // the 1x1 conv of one finished output row + the shared epilogue
        const long m = (n * p.OH + oh) * (long)p.OW + ow;
        const half_t* swt = swt_;
        if constexpr (LAUNDER) asm volatile("" : "+v"(swt));
        for (int j = 0; j < ntile; ++j) {
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 wh = *reinterpret_cast<const half8*>(swt + (j * 32 + wr) * ROWH + ks * 16 + fj * 8);
                const half8 wl = *reinterpret_cast<const half8*>(swt + ((ntile + j) * 32 + wr) * ROWH + ks * 16 + fj * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], acc, 0, 0, 0);
            }
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            if (ow < p.OW) conv_epilogue_tile(p, acc, bias, m, n, oh, ow, j * 32, lane);
        }
    };

    float A[KS][8], B[KS][8];
    half8 cx[K], cl_[K], nx[K], nl[K];
    // PF: the next item's gathers fly during the current item's multiply-adds (24 VGPRs); the widest stride-1 units (two live accumulator
    // sets of 8 KS registers each) gather in place instead and stay spill-free
    constexpr bool PF = !(S == 1 && KS >= 5);
    if constexpr (S == 1 && !PF) {
        auto item = [&](int iy, int ks) __attribute__((always_inline)) { load_item(iy, ks, cx, cl_); };
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            item(r0 - 1, ks);
            set_bias(A[ks], ks);
            taps(A[ks], 0, ks, cx, cl_, r0 - 1 >= 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            item(r0, ks);
            taps(A[ks], 1, ks, cx, cl_, true);
            set_bias(B[ks], ks);
            taps(B[ks], 0, ks, cx, cl_, true);
        }
        auto step = [&](int r, float (&P)[KS][8], float (&Q)[KS][8]) __attribute__((always_inline)) {
            const bool rowok = r + 1 < p.H;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                item(r + 1, ks);
                taps(P[ks], 2, ks, cx, cl_, rowok);
                finish(P[ks], ks);
                set_bias(P[ks], ks);
                taps(P[ks], 0, ks, cx, cl_, rowok);
                taps(Q[ks], 1, ks, cx, cl_, rowok);
            }
            emit_row(r);
        };
#pragma unroll 1
        for (int r = r0; r < r1; r += 2) {
            step(r, A, B);
            if (r + 1 < r1) step(r + 1, B, A);
        }
        (void)nx; (void)nl;
    } else if constexpr (S == 1) {
        // items in order: (r0 - 1, ks..), (r0, ks..), then per output row r: (r + 1, ks..)
        load_item(r0 - 1, 0, nx, nl);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {                    // input row r0 - 1: filter row 0 of output row r0
#pragma unroll
            for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
            if (ks + 1 < KS) load_item(r0 - 1, ks + 1, nx, nl); else load_item(r0, 0, nx, nl);
            set_bias(A[ks], ks);
            taps(A[ks], 0, ks, cx, cl_, r0 - 1 >= 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {                    // input row r0: filter row 1 of output row r0, filter row 0 of r0 + 1
#pragma unroll
            for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
            if (ks + 1 < KS) load_item(r0, ks + 1, nx, nl); else load_item(r0 + 1, 0, nx, nl);
            taps(A[ks], 1, ks, cx, cl_, true);
            set_bias(B[ks], ks);
            taps(B[ks], 0, ks, cx, cl_, true);
        }
        auto step = [&](int r, float (&P)[KS][8], float (&Q)[KS][8]) __attribute__((always_inline)) {      // P: output row r (rows 0, 1 applied), Q: row r + 1 (row 0)
            const bool rowok = r + 1 < p.H;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
                if (ks + 1 < KS) load_item(r + 1, ks + 1, nx, nl); else load_item(r + 2, 0, nx, nl);
                taps(P[ks], 2, ks, cx, cl_, rowok);
                finish(P[ks], ks);
                set_bias(P[ks], ks);                          // P now carries output row r + 2
                taps(P[ks], 0, ks, cx, cl_, rowok);
                taps(Q[ks], 1, ks, cx, cl_, rowok);
            }
            emit_row(r);
        };
#pragma unroll 1
        for (int r = r0; r < r1; r += 2) {
            step(r, A, B);
            if (r + 1 < r1) step(r + 1, B, A);
        }
    } else {
        // stride 2: output row r reads input rows 2r - 1, 2r, 2r + 1; items: (2 r0 - 1, ks..), then per row r and slice: (2r, ks), (2r + 1, ks)
        load_item(2 * r0 - 1, 0, nx, nl);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
            if (ks + 1 < KS) load_item(2 * r0 - 1, ks + 1, nx, nl); else load_item(2 * r0, 0, nx, nl);
            set_bias(A[ks], ks);
            taps(A[ks], 0, ks, cx, cl_, 2 * r0 - 1 >= 0);
        }
#pragma unroll 1
        for (int r = r0; r < r1; ++r) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
                load_item(2 * r + 1, ks, nx, nl);
                taps(A[ks], 1, ks, cx, cl_, 2 * r < p.H);
#pragma unroll
                for (int d = 0; d < K; ++d) { cx[d] = nx[d]; if constexpr (LO) cl_[d] = nl[d]; }
                if (ks + 1 < KS) load_item(2 * r, ks + 1, nx, nl); else load_item(2 * r + 2, 0, nx, nl);
                const bool rowok = 2 * r + 1 < p.H;
                taps(A[ks], 2, ks, cx, cl_, rowok);
                finish(A[ks], ks);
                set_bias(A[ks], ks);
                taps(A[ks], 0, ks, cx, cl_, rowok);
            }
            emit_row(r);
        }
        (void)B;
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

// the row-streaming form takes 3 x 3 'same' filters (pad 1) over <= 3 slices of 16 channels: there the row-invariant depthwise weights
// stay in registers (see LAUNDER in the kernel).  Wider units measured SLOWER than the tile form with the weight reads left in the row
// loop (96 -> 192 @34 x 60, stride 2: 0.208 vs 0.168 ms) and spill with them hoisted: they keep the tile form.
#define DWPW_ROWS_MAX_KS 3
#define DWPW_ROWS_MAX_KS_S1 3

// rows per strip segment of the row-streaming form: long strips amortise the two extra input rows and the block prologue, short ones keep
// enough waves in flight on small maps (>= ~8 k waves where the map allows it)
static int dwpw_rows_per_segment(long nimg, int OH, int strips) {
    const long tile_rows = nimg * OH * strips;
    long rs = tile_rows / 8192;
    if (rs < 4) rs = 4;
    if (rs > 16) rs = 16;
    const int nseg = (int)((OH + rs - 1) / rs);
    return (OH + nseg - 1) / nseg;                       // equal segments
}

template <int KS, bool LO>
static int launch_dwpw_rows_t(const ConvParams& pin, hipStream_t st) {
    ConvParams p = pin;
    const int ntile = (p.Np + 31) >> 5;
    const size_t lds = (size_t)2 * ntile * 32 * (KS * 16 + 8) * 2 + (size_t)(9 + 1) * KS * 16 * 4 + (size_t)ntile * 32 * 4;
    constexpr bool S1 = KS <= DWPW_ROWS_MAX_KS_S1;       // (the stride-1 form of wider units would spill: not instantiated)
    static VseDevOnce attr_once;
    if (!vse_dev_once(attr_once, [] {
            bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dwpw_rows_kernel<KS, LO, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
            if constexpr (S1)
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dwpw_rows_kernel<KS, LO, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
            return ok;
        }))
        return VSE_E_HIP;
    const long nimg = p.M / ((long)p.OH * p.OW);
    if (nimg <= 0 || p.M != nimg * p.OH * p.OW || p.M >= 0x7fffffffl || lds > 128 * 1024) return VSE_E_INVAL;
    p.tiles_w = (p.OW + 31) / 32;
    const int rs = dwpw_rows_per_segment(nimg, p.OH, p.tiles_w);
    p.ntiles = (unsigned)rs;
    p.tiles_h = (p.OH + rs - 1) / rs;
    const unsigned long long waves = (unsigned long long)nimg * p.tiles_h * p.tiles_w;
    const unsigned long long blocks = (waves + 3) / 4;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    if (p.sh == 1) {
        if constexpr (S1) hipLaunchKernelGGL((conv_dwpw_rows_kernel<KS, LO, 1>), dim3((unsigned)blocks), dim3(256), lds, st, p);
        else return VSE_E_UNSUPPORTED;
    } else hipLaunchKernelGGL((conv_dwpw_rows_kernel<KS, LO, 2>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}

// VSE_DWPW_ROWS=0 (development builds) keeps the tile form for A/B runs and the bit-identity check (tools/ab_env_digest.py)
// Only PAIR inputs take it: on plain fp16 inputs (the layer-by-layer programs) the tile form measures the same or better (16 -> 32
// @272 x 480: 0.373 vs 0.365 ms, 48 -> 48 @136 x 240: 0.215 vs 0.286) — half the gathers and half the multiply-adds per pixel leave
// little for the strip walk to save.
int conv_dwpw_rows_stride(int k, int pad, int s, int cinp, int lo_in) {
    static const bool on = [] { const char* e = vse_dev_getenv("VSE_DWPW_ROWS"); return !(e && e[0] == '0'); }();
    const int ks = (cinp + 15) / 16;
    return (on && lo_in != 0 && k == 3 && pad == 1 && ((s == 1 && ks <= DWPW_ROWS_MAX_KS_S1) || (s == 2 && ks <= DWPW_ROWS_MAX_KS))) ? s : 0;
}
static bool dwpw_rows_wanted(const ConvParams& p, int ks) { (void)ks; return conv_dwpw_rows_stride(p.kh, p.ph, p.sh, p.cinp, p.in_lo_off) != 0; }

// p.kh / p.sh / p.ph describe the DEPTHWISE conv (the 1x1 conv has no geometry); p.dotw = the aux blob; p.in_lo_off = the input's pair offset
int launch_conv_dwpw(const ConvParams& p, hipStream_t st) {
    if (!conv_dwpw_ok(p.kh, p.sh, p.cinp, p.Np, p.flags) || p.kh != p.kw || p.sh != p.sw || p.ph != p.pw || !p.dotw) return VSE_E_UNSUPPORTED;
    const int ks = (p.cinp + 15) / 16;
    if (dwpw_rows_wanted(p, ks)) {
#define DWPW_ROWS(KS_) launch_dwpw_rows_t<KS_, true>(p, st)
        switch (ks) {
            case 1: return DWPW_ROWS(1);
            case 2: return DWPW_ROWS(2);
            default: return DWPW_ROWS(3);
        }
#undef DWPW_ROWS
    }
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
