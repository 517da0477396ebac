// 3x3 stem convolution over an image-like input (<= 4 real channels, stored 8-channel padded): F_STEM.
//
// The generic kernels treat the stem as an implicit GEMM with K = 9 taps x 8 padded channels = 72 -> 128: 79 % of the
// MFMA work multiplies padding and every tap re-gathers its pixels through L2 (0.65 ms for the detector's 64 x 544 x 960
// stem against 0.27 ms of compulsory HBM traffic).  Here a block stages the input patch of an 8 x 32 output tile in LDS
// ONCE (first 4 channels = 8 bytes per pixel) and the whole weight matrix (<= 64 couts x 80) lives in 40 VGPRs per lane.
// K keeps the generic kernels' order — k slice ks = taps 2ks and 2ks+1 with 8 channels each (4..7 zero, tap 9 zero) — so
// the fp32 accumulation order, and with it every output bit, is the one of conv_mfma_kernel (a 12-tap x 4-channel K
// would save 8 of 20 MFMAs per wave, but changes the rounding of a few outputs per million, and box corners of the real
// detector moved by a pixel in the parity test; the kernel is bound by its loads and stores, not by the MFMAs).
//   block = 256 threads = 4 waves, wave w -> output rows 2w, 2w+1 of the tile, all couts (2 x 32-cout MFMA tiles)
//   LDS   = ((8-1)*sh+3) x ((32-1)*sw+3) pixels x 8 B  (17 x 65 x 8 = 8.8 KiB at stride 2) + bias
//   B fragment of k slice ks for lane (px, fj): tap 2 ks + fj -> one ds_read_b64 at the pixel that tap addresses (upper four
//   channels zero); tap 9 carries zero weights and reads tap 0's pixel.
// Same epilogue as every other conv kernel (conv_epilogue_tile).  Weights are packed [Np][10 taps][8] by the compiler.
#include "conv_common.h"
#include "resize_u8.h"

#ifndef VSE_STEM_WIDE
#define VSE_STEM_WIDE 1
#endif
#define ST_ROWS 8
#define ST_COLS 32

// HILO is a template parameter: the lo table costs 40 VGPRs per lane (a resident block per CU less) and 40 KiB of loads per block,
// which plain fp16 nets must not pay.  The weight tables are staged ONCE per block in LDS (<= 10 KiB each, 16 bytes per thread and
// pass) and the fragments read from there: fetching the 20 fragments per lane straight from global memory moved 40-80 KiB through
// the L1 / texture path per 256-pixel tile — more than the tile's input patch and output together.
// U8 (F_U8SRC, round 3): the detector's pre-processing rides in the patch staging — the block resizes the pixels of its input patch
// from the uint8 BGR frame (cv2 fixed-point INTER_LINEAR, resize_u8.h: the bytes det_preprocess_kernel writes) and stages
// [u0, u1, u2, 1] as fp16 (exact); the stem's weights carry the normalisation (compiler input_norm).  The 8-channel fp16 detector
// input (8.4 MB per 544 x 960 frame) is never written or read.
template <int SH, int SW, bool HILO, bool U8 = false>
__global__ __launch_bounds__(256) void conv_stem_kernel(const ConvParams p) {
    constexpr int PH_ = (ST_ROWS - 1) * SH + 3, PW_ = (ST_COLS - 1) * SW + 3, NPIX = PH_ * PW_;
    constexpr int NPIX4 = (NPIX * 4 + 7) & ~7;                                   // 16-byte aligned start of the weight tables
    constexpr int WTAB = 64 * 80;
    __shared__ __attribute__((aligned(16))) half_t patch[NPIX4 + (HILO ? 2 : 1) * WTAB + 128];      // + 64 floats of bias
    half_t* const swt = patch + NPIX4;
    float* const sbias = reinterpret_cast<float*>(swt + (HILO ? 2 : 1) * WTAB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fx = lane & 31, fj = lane >> 5;

    // (one tile per block: a block that walks 8 consecutive tiles was measured 15 % slower — its tiles run back to back
    // and nothing overlaps inside it, while 4 resident blocks per CU overlap each other's load / compute / store phases)
    unsigned t = xcd_block(blockIdx.x, gridDim.x);          // (XCD-contiguous tile order: vertical neighbours share their halo rows in one L2)
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
    const int iy0 = oy0 * SH - p.ph, ix0 = ox0 * SW - p.pw;

    // Round 5: EVERY global load of the prologue — weight tables and input patch — is issued unconditionally (clamped address, value
    // selected afterwards) and before the first use, so the block pays ONE memory round trip.  The loads used to sit under their
    // range checks inside the staging loops: hipcc branches around such a load and waits vmcnt(0) behind it, i.e. 3 + 5 dependent
    // round trips per block (the kernel ran at 2.2 TB/s of its bytes with its waves waiting 68 % of the time).
    constexpr int NIT = (NPIX + 255) / 256;                 // patch pixels per thread
    constexpr int NWV = (HILO ? 2 : 1) * 640;               // 16-byte vectors of the weight tables [Np][80] (hi, then lo)
    constexpr int NWIT = (NWV + 255) / 256;
    // resize coefficients of the patch's rows and columns (clamped to the image): one lin_coef per row / column and block instead
    // of two per patch pixel (double-precision division)
    __shared__ LinCoef scoef[U8 ? PH_ + PW_ : 1];
    if constexpr (U8) {
        if (tid < PH_ + PW_) {
            const bool row = tid < PH_;
            const int d = row ? iy0 + tid : ix0 + (tid - PH_);
            const int dst = row ? p.H : p.W;
            scoef[tid] = lin_coef(min(max(d, 0), dst - 1), dst, row ? p.u8_h : p.u8_w);
        }
        __syncthreads();
    }
    half8 wv[NWIT];
    {
        const int nvec = p.Np * 10;                                             // 16-byte vectors per table
#pragma unroll
        for (int k = 0; k < NWIT; ++k) {
            const int v = min(tid + 256 * k, NWV - 1);
            const int tab = v / 640, u = v - tab * 640;
            wv[k] = *reinterpret_cast<const half8*>(p.w + (long)tab * p.Np * 80 + min(u, nvec - 1) * 8);
        }
    }
    float bv = 0.f;
    if (tid < 64) bv = p.bias[min(tid, p.Np - 1)];

    // ---- input patch: channels 0..3 of every pixel, zeros outside the image ---------------------------------------------------
    if constexpr (U8) {
        const uint8_t* fb = p.u8src + img * p.u8_fstride;
        const int rowb = p.u8_w * 3;                       // bytes of a source row (launch_conv_stem: >= 8)
        unsigned long long q0[NIT], q1[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = min(tid + 256 * k, NPIX - 1);
            const int py = q / PW_, px = q - py * PW_;
            const LinCoef cy = scoef[py], cx = scoef[PH_ + px];
            // both source pixels of a row = 6 consecutive bytes: ONE (unaligned) 8-byte load per row; in the last columns the load starts
            // earlier and the bytes are shifted down (the second pixel is then the first one again: x1 = s0)
            const int b0 = min(cx.s0 * 3, rowb - 8);
            __builtin_memcpy(&q0[k], fb + (long)cy.s0 * p.u8_pitch + b0, 8);
            __builtin_memcpy(&q1[k], fb + (long)min(cy.s0 + 1, p.u8_h - 1) * p.u8_pitch + b0, 8);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = tid + 256 * k;
            const int qc = min(q, NPIX - 1);
            const int py = qc / PW_, px = qc - py * PW_;
            const int iy = iy0 + py, ix = ix0 + px;
            const LinCoef cy = scoef[py], cx = scoef[PH_ + px];
            const int sh0 = (cx.s0 * 3 - min(cx.s0 * 3, rowb - 8)) * 8;
            const int sh1 = sh0 + (cx.s0 + 1 < p.u8_w ? 24 : 0);
            half4 v = half4{0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int u = cv_bilinear_u8((int)((q0[k] >> (sh0 + 8 * c)) & 255), (int)((q0[k] >> (sh1 + 8 * c)) & 255),
                                             (int)((q1[k] >> (sh0 + 8 * c)) & 255), (int)((q1[k] >> (sh1 + 8 * c)) & 255), cx, cy);
                v[c] = (half_t)(float)u;
            }
            v[3] = (half_t)1.f;
            if (!(iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)) v = half4{0, 0, 0, 0};
            if (q < NPIX) *reinterpret_cast<half4*>(patch + q * 4) = v;
        }
    } else {
        const half_t* base = p.in + img * (long)p.Hs * p.Ws * p.in_ld;
        half4 pv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = min(tid + 256 * k, NPIX - 1);
            const int py = q / PW_, px = q - py * PW_;
            const int iy = min(max(iy0 + py, 0), p.H - 1), ix = min(max(ix0 + px, 0), p.W - 1);
            pv[k] = *reinterpret_cast<const half4*>(base + ((long)iy * p.Ws + ix) * p.in_ld);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = tid + 256 * k;
            const int py = q / PW_, px = q - py * PW_;
            const int iy = iy0 + py, ix = ix0 + px;
            const half4 v = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? pv[k] : half4{0, 0, 0, 0};
            if (q < NPIX) *reinterpret_cast<half4*>(patch + q * 4) = v;
        }
    }
    // ---- weight tables -> LDS, rows >= Np zero -----------------------------------------------------------------------------------
    {
        const int nvec = p.Np * 10;
#pragma unroll
        for (int k = 0; k < NWIT; ++k) {
            const int v = tid + 256 * k;
            const int tab = v / 640, u = v - tab * 640;
            if (v < NWV) *reinterpret_cast<half8*>(swt + tab * WTAB + u * 8) = u < nvec ? wv[k] : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    if (tid < 64) sbias[tid] = tid < p.Np ? bv : 0.f;
    __syncthreads();

    // ---- weights: lane (f = lane & 31, fj) supplies cout conv_wrow(f) of each 32-cout tile, k = 16 ks + 8 fj .. +7 ----------
    half8 wf[2][5];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 32 + conv_wrow(fx);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) wf[j][ks] = *reinterpret_cast<const half8*>(swt + r * 80 + ks * 16 + fj * 8);
    }

    // ---- 20 MFMAs per wave -------------------------------------------------------------------------------------------------------
    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        int t0 = 2 * ks + fj;
        t0 = t0 < 9 ? t0 : 0;                      // zero-weight tap: any valid pixel
        const int o0 = (t0 / 3) * PW_ + t0 % 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = ((2 * wave + i) * SH) * PW_ + fx * SW;
            const half4 a = *reinterpret_cast<const half4*>(patch + (q + o0) * 4);
            const half8 xf = half8{a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][ks], xf, acc[i][j], 0, 0, 0);
        }
    }

    if constexpr (HILO) {
        // F_HILO: the fp16 lo parts of the weights (second table), multiplied in a second pass into the same accumulators
        half8 wl[2][5];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = j * 32 + conv_wrow(fx);
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) wl[j][ks] = *reinterpret_cast<const half8*>(swt + WTAB + r * 80 + ks * 16 + fj * 8);
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            int t0 = 2 * ks + fj;
            t0 = t0 < 9 ? t0 : 0;
            const int o0 = (t0 / 3) * PW_ + t0 % 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = ((2 * wave + i) * SH) * PW_ + fx * SW;
                const half4 a = *reinterpret_cast<const half4*>(patch + (q + o0) * 4);
                const half8 xf = half8{a[0], a[1], a[2], a[3], 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j][ks], xf, acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + fx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const long m = (img * p.OH + oy) * p.OW + ox;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j * 32 >= p.Np) continue;
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, img, oy, ox, j * 32, lane);
        }
    }
}

int launch_conv_stem(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (p.kh != 3 || p.kw != 3 || p.ph != 1 || p.pw != 1 || p.cinp != 8 || p.inshift != 0 || p.Np > 64 || p.Np < 1) return VSE_E_INVAL;
    if (p.flags & (F_PIXSHUF | F_DOT1 | F_SRC2 | F_PATCH | F_UP2HEAD)) return VSE_E_INVAL;
    if (!((p.sh == 1 && p.sw == 1) || (p.sh == 2 && p.sw == 2))) return VSE_E_INVAL;
    p.tiles_h = (p.OH + ST_ROWS - 1) / ST_ROWS;
    p.tiles_w = (p.OW + ST_COLS - 1) / ST_COLS;
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    const dim3 grid((unsigned)blocks), block(256);
    const bool hilo = p.flags & F_HILO;
    if (p.flags & F_U8SRC) {
        if (!p.u8src || p.u8_h <= 0 || p.u8_w < 3) return VSE_E_INVAL;          // (8-byte row loads: >= 9 bytes per source row)
        if (p.sh == 2 && hilo) hipLaunchKernelGGL((conv_stem_kernel<2, 2, true, true>), grid, block, 0, st, p);
        else if (p.sh == 2) hipLaunchKernelGGL((conv_stem_kernel<2, 2, false, true>), grid, block, 0, st, p);
        else if (hilo) hipLaunchKernelGGL((conv_stem_kernel<1, 1, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_stem_kernel<1, 1, false, true>), grid, block, 0, st, p);
        return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
    }
    if (p.sh == 2 && hilo) hipLaunchKernelGGL((conv_stem_kernel<2, 2, true>), grid, block, 0, st, p);
    else if (p.sh == 2) hipLaunchKernelGGL((conv_stem_kernel<2, 2, false>), grid, block, 0, st, p);
    else if (hilo) hipLaunchKernelGGL((conv_stem_kernel<1, 1, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_stem_kernel<1, 1, false>), grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
