// Pointwise (1x1, stride 1) convolution over SMALL channel counts (cin <= 64, cout <= 64): F_PW.
//
// The implicit-GEMM kernels stage a 256-pixel tile and its weights through an LDS ring; with K = 16..64 that is a prologue, one or
// two K steps and an epilogue per block — the detector's IntraCL 1x1 layers (32 <-> 64 channels @136x240) take 0.10-0.20 ms against
// 0.06 ms of compulsory HBM traffic.  With so few channels nothing needs staging: a pixel's channels are 32-128 contiguous bytes, so
// the MFMA B fragment of lane (pixel, k-half) is ONE 16-byte global load (consecutive lanes = consecutive pixels: fully coalesced when
// the tensor is dense), the whole weight matrix lives in <= 32 VGPRs per lane, and a wave streams 64 pixels at a time.
//   block = 256 threads = 4 waves x 64 pixels (2 MFMA pixel tiles) x all couts (TN = 1 | 2 tiles of 32)
//   K order: slices of 16 channels in order — the accumulation order of the implicit-GEMM kernels (bit-identical results)
// Weights: plain [Np][Kp] fp16 (Kp = cinp), bias fp32 [Np].  Same epilogue as every conv kernel (conv_epilogue_tile).
#include "conv_common.h"

template <int TN, int KS>       // 32-cout tiles, 16-channel K slices
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvParams p) {
    __shared__ float sbias[64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fx = lane & 31, fj = lane >> 5;
    if (tid < 64) sbias[tid] = tid < p.Np ? p.bias[tid] : 0.f;

    half8 wf[TN][KS];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = j * 32 + conv_wrow(fx);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[j][ks] = r < p.Np ? *reinterpret_cast<const half8*>(p.w + (long)r * (KS * 16) + ks * 16 + fj * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const long m0 = (long)blockIdx.x * 256 + wave * 64;
    half8 xf[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = m0 + i * 32 + fx;
        const half_t* src = p.in + (m < p.M ? m : 0) * (long)p.in_ld + fj * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[i][ks] = *reinterpret_cast<const half8*>(src + ks * 16);
    }
    float16v acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][ks], xf[i][ks], acc[i][j], 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = m0 + i * 32 + fx;
        if (m >= p.M) continue;
        const int ow = (int)(m % p.OW);
        const long t = m / p.OW;
        const int oh = (int)(t % p.OH);
        const long n = t / p.OH;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bias[16];
            conv_epilogue_consts(sbias, j * 32, lane, bias);
            conv_epilogue_tile(p, acc[i][j], bias, m, n, oh, ow, j * 32, lane);
        }
    }
}

bool conv_pw_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Np, int inshift, int flags) {
    return kh == 1 && kw == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && inshift == 0 && (cinp & 15) == 0 && cinp <= 64 && Np <= 64
           && !(flags & (F_SRC2 | F_PIXSHUF | F_DOT1 | F_HILO | F_PATCH | F_COL));
}

int launch_conv_pw(const ConvParams& p, hipStream_t st) {
    if (!conv_pw_ok(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.cinp, p.Np, p.inshift, p.flags)) return VSE_E_UNSUPPORTED;
    const unsigned long long blocks = (unsigned long long)((p.M + 255) / 256);
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    const dim3 grid((unsigned)blocks), block(256);
    const int ks = p.cinp / 16;
    const bool two = p.Np > 32;
#define PW_CASE(K) case K: if (two) hipLaunchKernelGGL((conv_pw_kernel<2, K>), grid, block, 0, st, p); \
                           else hipLaunchKernelGGL((conv_pw_kernel<1, K>), grid, block, 0, st, p); break;
    switch (ks) { PW_CASE(1) PW_CASE(2) PW_CASE(3) PW_CASE(4) default: return VSE_E_UNSUPPORTED; }
#undef PW_CASE
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
