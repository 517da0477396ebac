// Pointwise (1x1, stride 1) convolution over FEW INPUT CHANNELS (cin <= 64, <= 96 with hi + lo weights; any cout up to 256): F_PW.
//
// The implicit-GEMM kernels stage a 256-pixel tile and its weights through an LDS ring; with K = 16..64 that is a prologue, one or
// two K steps and an epilogue per block — the detector's IntraCL 1x1 layers (32 <-> 64 channels @136x240) take 0.10-0.20 ms against
// 0.06 ms of compulsory HBM traffic, the 2x2 transposed convs of its head (64 -> 4 x 64 / 4 x 8, pixel-shuffle store) 0.41 / 0.46 ms.
// With so few input channels the activations need no staging: a pixel's channels are 32-128 contiguous bytes, so the MFMA B fragment
// of lane (pixel, k-half) is ONE 16-byte global load (consecutive lanes = consecutive pixels: fully coalesced when the tensor is
// dense) and a wave keeps its 64 pixels in registers while it walks the cout tiles.
//   block = 256 threads = 4 waves x 64 pixels (2 MFMA pixel tiles); the weight matrix [Np][cinp] is staged ONCE per block in LDS
//           (rows padded by 16 bytes: conflict-free fragment reads) and read per 32-cout tile
//   K order: slices of 16 channels in order — the accumulation order of the implicit-GEMM kernels (bit-identical results)
// Weights: plain [Np][cinp] fp16, bias fp32 [Np].  Same epilogue as every conv kernel (conv_epilogue_tile, F_PIXSHUF included).
// F_HILO (round 3; the mobile detectors run with fp16 hi + lo weight pairs and used to fall back to the generic kernel for all their
// 1x1 layers): the lo table [Np][cinp] follows the hi table; both are staged (Np <= 128) and the K slices are walked twice over the
// SAME activation fragments — hi slices, then lo slices, the order of the implicit-GEMM kernels' two-pass K walk.
#include "conv_common.h"

#define PW_MAXN 256
#define PW_MAXN_HILO 128

template <int KS, bool HILO = false>       // 16-channel K slices
__global__ __launch_bounds__(256, (KS <= 2 ? 4 : KS <= 4 ? 3 : 2)) void conv_pw_kernel(const ConvParams p) {
    constexpr int ROWH = KS * 16 + 8;                    // halfs per staged weight row (16 bytes of padding)
    constexpr int ROWS = HILO ? PW_MAXN_HILO : PW_MAXN, NT = HILO ? 2 : 1;
    __shared__ __attribute__((aligned(16))) half_t swt[NT * ROWS * ROWH];
    __shared__ float sbias[PW_MAXN];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fx = lane & 31, fj = lane >> 5;
    const int ntile = (p.Np + 31) >> 5;
    for (int v = tid; v < ntile * 32 * KS * 2; v += 256) {      // 16-byte vectors of the (zero-padded) weight matrix
        const int r = v / (KS * 2), c = v - r * (KS * 2);
        half8 x = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r < p.Np) x = *reinterpret_cast<const half8*>(p.w + (long)r * (KS * 16) + c * 8);      // rows are KS * 16 wide (zero columns behind cinp)
        *reinterpret_cast<half8*>(swt + r * ROWH + c * 8) = x;
        if constexpr (HILO) {
            half8 y = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (r < p.Np) y = *reinterpret_cast<const half8*>(p.w + (long)(p.Np + r) * (KS * 16) + c * 8);
            *reinterpret_cast<half8*>(swt + (ROWS + r) * ROWH + c * 8) = y;
        }
    }
    for (int c = tid; c < ntile * 32; c += 256) sbias[c] = c < p.Np ? p.bias[c] : 0.f;

    const long m0 = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + wave * 64;       // (XCD-contiguous block order: common.h)
    half8 xf[2][KS];
    long mm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mm[i] = m0 + i * 32 + fx;
        const half_t* src = p.in + (mm[i] < p.M ? mm[i] : 0) * (long)p.in_ld + fj * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // cinp % 16 == 8 (24 / 40 / 56 channels): the last half slice lies behind the pixel's channels — zeros, not the next pixel
            xf[i][ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ks * 16 + fj * 8 < p.cinp) xf[i][ks] = *reinterpret_cast<const half8*>(src + ks * 16);
        }
    }
    int oh[2], ow[2];
    long nn[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = mm[i] < p.M ? mm[i] : 0;
        conv_pix_coords(p, m, nn[i], oh[i], ow[i]);
    }
    __syncthreads();
    const int wr = conv_wrow(fx);
    for (int j = 0; j < ntile; ++j) {
        half8 wf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const half8*>(swt + (j * 32 + wr) * ROWH + ks * 16 + fj * 8);
        float16v acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf[i][ks], acc[i], 0, 0, 0);
        if constexpr (HILO) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const half8*>(swt + (ROWS + j * 32 + wr) * ROWH + ks * 16 + fj * 8);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf[i][ks], acc[i], 0, 0, 0);
        }
        float bias[16];
        conv_epilogue_consts(sbias, j * 32, lane, bias);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (mm[i] < p.M) conv_epilogue_tile(p, acc[i], bias, mm[i], nn[i], oh[i], ow[i], j * 32, lane);
    }
}

bool conv_pw_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Np, int inshift, int flags) {
    return kh == 1 && kw == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && inshift == 0 && (cinp & 7) == 0
           && cinp <= ((flags & F_HILO) ? 96 : 64)      // (hi + lo nets: a 48-channel PAIR tensor is 96 input channels — round 5)
           && Np <= ((flags & F_HILO) ? PW_MAXN_HILO : PW_MAXN) && !(flags & (F_SRC2 | F_DOT1 | F_PATCH | F_COL));
}

int launch_conv_pw(const ConvParams& p, hipStream_t st) {
    if (!conv_pw_ok(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.cinp, p.Np, p.inshift, p.flags)) return VSE_E_UNSUPPORTED;
    const unsigned long long blocks = (unsigned long long)((p.M + 255) / 256);
    if (blocks == 0 || p.M >= 0x7fffffffl) return VSE_E_INVAL;          // (32-bit pixel arithmetic: conv_pix_coords)
    const dim3 grid((unsigned)blocks), block(256);
    if (p.flags & F_HILO) {
        switch ((p.cinp + 15) / 16) {
            case 1: hipLaunchKernelGGL((conv_pw_kernel<1, true>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((conv_pw_kernel<2, true>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((conv_pw_kernel<3, true>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((conv_pw_kernel<4, true>), grid, block, 0, st, p); break;
            case 5: hipLaunchKernelGGL((conv_pw_kernel<5, true>), grid, block, 0, st, p); break;
            case 6: hipLaunchKernelGGL((conv_pw_kernel<6, true>), grid, block, 0, st, p); break;
            default: return VSE_E_UNSUPPORTED;
        }
        return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
    }
    switch ((p.cinp + 15) / 16) {
        case 1: hipLaunchKernelGGL((conv_pw_kernel<1>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((conv_pw_kernel<2>), grid, block, 0, st, p); break;
        case 3: hipLaunchKernelGGL((conv_pw_kernel<3>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((conv_pw_kernel<4>), grid, block, 0, st, p); break;
        default: return VSE_E_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
