// 1x1 convolution over a HANDFUL of pixels (M <= 256): the SE / ESE gate convs — one pixel per image behind a global average
// pool, 256 .. 1024 channels in and out (gfx950 / CDNA4).
//
// conv_gemm_kernel gives such a layer Np / 128 blocks of a 128 x 128 tile (2 .. 8 blocks on 256 CUs), each walking all of K through
// its LDS ring: 22 - 29 us per layer, five layers per recogniser launch sequence and per detector pass.  The layer is a GEMV-shaped
// weight stream (Np x K x 2 bytes, L2-resident) against a tiny activation matrix, so here
//   wave   = one block = 32 couts x 32 pixels (one accumulator tile); grid = Np / 32 x ceil(M / 32) waves all over the chip;
//   K loop = 16-channel slices, both MFMA operands straight from global memory (the weight fragment of lane (cout row, k half) is
//            one 16-byte load from the [K / KT][Np][KT] tiles conv_gemm_kernel reads; the activation fragment of lane (pixel, k half)
//            one 16-byte NHWC load), eight slices of loads in flight;
//   order  = k ascending, one fp32 accumulator — conv_gemm_kernel's order, so the bits are identical;
//   epilogue = conv_epilogue_tile (bias, activation, residual, ragged mask, fp16 / fp32 stores), as everywhere.
// No LDS but the 32 bias values, no barrier in the K loop.
#include "conv_common.h"

#define SM_UNROLL 8

template <int KT, bool HILO>       // HILO (round 5, the mobile detectors' SE convs): the lo weight tiles follow the hi tiles; K is walked twice
__device__ __forceinline__ void conv_smallm_body(const ConvParams& p, float* sbias) {
    const int lane = threadIdx.x;
    const int fx = lane & 31, fj = lane >> 5;
    const int n0 = blockIdx.x * 32;                          // first cout of this wave
    const long m0 = (long)blockIdx.y * 32;                   // first pixel
    if (lane < 32) sbias[lane] = (n0 + lane < p.Np) ? p.bias[n0 + lane] : 0.f;

    const int co = n0 + conv_wrow(fx);                       // the cout whose weights this lane supplies
    const bool wok = co < p.Np;
    const half_t* wl = p.w + (long)(wok ? co : 0) * KT + fj * 8;
    const long wstep = (long)p.Np * KT;                      // elements between K tiles
    const long m = m0 + fx;                                  // the pixel whose activations this lane supplies
    const bool xok = m < p.M;
    const half_t* xl = p.in + (xok ? m : 0) * (long)p.in_ld + fj * 8;

    float16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const half8 zero8 = half8{0, 0, 0, 0, 0, 0, 0, 0};
    const int nslice = (p.cinp + 15) >> 4;                   // (cinp % 16 == 8: the upper half of the last slice lies behind the pixel's channels: zeros)
    // (hi pass, then lo pass over the same activations into the same accumulator: conv_gemm_kernel's two-pass K walk, identical bits)
#pragma unroll 1
    for (int pass = 0; pass < (HILO ? 2 : 1); ++pass) {
        const half_t* wp = wl + (long)pass * p.nkh * wstep;
        for (int s0 = 0; s0 < nslice; s0 += SM_UNROLL) {
            half8 wf[SM_UNROLL], xf[SM_UNROLL];
#pragma unroll
            for (int u = 0; u < SM_UNROLL; ++u) {
                const int k = (s0 + u) << 4;
                const bool live = k + fj * 8 < p.cinp;
                wf[u] = (live && wok) ? *reinterpret_cast<const half8*>(wp + (long)(k / KT) * wstep + (k % KT)) : zero8;
                xf[u] = (live && xok) ? *reinterpret_cast<const half8*>(xl + k) : zero8;
            }
#pragma unroll
            for (int u = 0; u < SM_UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[u], xf[u], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (xok) {
        const int hw = p.OH * p.OW;
        const long n = m / hw;
        const int r = (int)(m - n * hw);
        float bias[16];
        conv_epilogue_consts(sbias, 0, lane, bias);
        conv_epilogue_tile(p, acc, bias, m, n, r / p.OW, r % p.OW, n0, lane);
    }
}

template <int KT>
__global__ __launch_bounds__(64) void conv_smallm_kernel(const ConvParams p) {
    __shared__ float sbias[32];
    conv_smallm_body<KT, false>(p, sbias);
}
template <int KT>
__global__ __launch_bounds__(64) void conv_smallm_hl_kernel(const ConvParams p) {
    __shared__ float sbias[32];
    conv_smallm_body<KT, true>(p, sbias);
}

// Layers this kernel serves: conv_gemm_kernel's unmasked 1x1 mode (mode 2) at stride 1 with at most 256 pixels, one weight stream.
bool conv_smallm_shape_ok(int mode, long M, int sh, int sw, int same_hw, int flags, int cinp) {
    static const bool on = [] { const char* e = vse_dev_getenv("VSE_SMALLM"); return !(e && e[0] == '0'); }();
    return on && mode == 2 && M <= 256 && sh == 1 && sw == 1 && same_hw && !(flags & (F_IMGW | F_PIXSHUF | F_DOT1 | F_SRC2)) && (cinp & 15) == 0;
}
bool conv_smallm_ok(const ConvParams& p, int mode) {
    return conv_smallm_shape_ok(mode, p.M, p.sh, p.sw, p.H == p.OH && p.W == p.OW && p.Hs == p.H && p.Ws == p.W, p.flags, p.cinp);
}
// ... and (round 5) SMALL 1x1 PROBLEMS whatever their route would be: <= 256 input channels (any multiple of 8: the SVTR necks' 120 / 240
// are not multiples of 32 and ran on the 128 x 128 tile of conv_mfma_kernel), <= 4096 (32-cout x 32-pixel) wave tiles — a recogniser
// sequence's [crops, 1, T, 120] layers.  Such a launch is a handful of K steps behind a prologue and in front of an epilogue on 13-50 of
// 256 CUs (13-23 us launch to launch); here every wave is its own block with ALL its loads in flight at once.  Same K order, same bits.
bool conv_smallk_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int inshift, int same_hw, int flags, int cinp, long M, int Np) {
    static const bool on = [] { const char* e = vse_dev_getenv("VSE_SMALLK"); return !(e && e[0] == '0'); }();
    return on && kh == 1 && kw == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && !inshift && same_hw && cinp > 0 && cinp <= 256 && (cinp & 7) == 0
           && !(flags & (F_IMGW | F_PIXSHUF | F_DOT1 | F_SRC2 | F_PATCH | F_COL | F_PW | F_STEM | F_UP2HEAD | F_DWPRE | F_ONECH | F_TAIL2 | F_HLSUM))
           && M > 0 && ((M + 31) / 32) * ((Np + 31) / 32) <= 4096;
}
bool conv_smallk_ok(const ConvParams& p) {
    return conv_smallk_shape_ok(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.inshift, p.H == p.OH && p.W == p.OW && p.Hs == p.H && p.Ws == p.W, p.flags, p.cinp,
                                p.M, p.Np);
}

int launch_conv_smallm(const ConvParams& p, hipStream_t st) {
    if ((p.M + 31) / 32 > 65535) return VSE_E_UNSUPPORTED;
    const dim3 grid((unsigned)((p.Np + 31) / 32), (unsigned)((p.M + 31) / 32)), block(64);
    if (p.flags & F_HILO) {
        if (p.flags & F_WK32) hipLaunchKernelGGL((conv_smallm_hl_kernel<32>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_smallm_hl_kernel<64>), grid, block, 0, st, p);
    } else if (p.flags & F_WK32) hipLaunchKernelGGL((conv_smallm_kernel<32>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_smallm_kernel<64>), grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
