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

// F_TAIL2 (round 5; the server detector's head tail: transposed conv 2x2 s2 c0 -> c1 + BN + relu, whose output f feeds BOTH the local
// refinement conv and a second transposed conv 2x2 s2 c1 -> 1 + sigmoid = the base map u): the second transposed conv rides in the
// epilogue of the first.  The accumulator tile of stage A — lane = pixel, register 8 g + e = channel 32 j + 16 g + 8 h + e — rounded
// to fp16 (the values stored as f) IS the MFMA B operand of K slice 2 j + g of stage B, a block-diagonal 1x1 conv 4 c1 -> 16 whose
// output 4 r + c is pixel (4 y + r, 4 x + c) of u (chain_pw2_kernel's scheme, chain.hip).  The slices of the other three sub-pixels add
// exact zeros, the four slices of a sub-pixel come in the unfused conv's order with its fp16 weights: u is BIT-IDENTICAL to the
// separate conv_pw launch it replaces (tests/test_gpu_nets.py), f is no longer re-read (1.07 GB per 64 frames) and u is stored as a dense
// fp16 map (ld = 1: 67 MB instead of 535 MB of 8-channel groups, and 67 instead of 535 MB fetched by conv_head_up2r_kernel).
// aux (p.dotw): stage B's A fragments [Np / 16 slices][k half][16 rows][8] fp16 (rows 16 .. 31 of the MFMA tile are zero and not stored).
template <int KS, bool HILO, bool TAIL>       // KS 16-channel K slices
__device__ __forceinline__ void conv_pw_body(const ConvParams& p, half_t* swt, float* sbias, half_t* swb) {
    constexpr int ROWH = KS * 16 + 8;                    // halfs per staged weight row (16 bytes of padding)
    constexpr int ROWS = HILO ? PW_MAXN_HILO : PW_MAXN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fx = lane & 31, fj = lane >> 5;
    const int ntile = (p.Np + 31) >> 5;
    // the wave's 64 pixels first, then the weight tables: every load of the prologue is unconditional and in flight before the first
    // use (stage_batched, common.h: the tables used to cost one memory round trip per 256 vectors — eight for 256 couts x 64 channels)
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
            const bool in = ks * 16 + fj * 8 < p.cinp;
            const half8 t = *reinterpret_cast<const half8*>(in ? src + ks * 16 : src - fj * 8);
            xf[i][ks] = in ? t : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    {
        const int nvec = ntile * 32 * KS * 2;                // 16-byte vectors of one (zero-padded) weight table
        const int rmax = p.Np - 1;
#pragma unroll
        for (int tab = 0; tab < (HILO ? 2 : 1); ++tab) {
            const half_t* wt = p.w + (long)tab * p.Np * (KS * 16);                 // rows are KS * 16 wide (zero columns behind cinp)
            half_t* dt = swt + tab * ROWS * ROWH;
            stage_batched<(HILO ? 4 : 8)>(nvec, tid,
                [&](int v) { const int r = v / (KS * 2), c = v - r * (KS * 2); return *reinterpret_cast<const half8*>(wt + (long)min(r, rmax) * (KS * 16) + c * 8); },
                [&](int v, half8 x) { const int r = v / (KS * 2), c = v - r * (KS * 2);
                                      *reinterpret_cast<half8*>(dt + r * ROWH + c * 8) = r <= rmax ? x : half8{0, 0, 0, 0, 0, 0, 0, 0}; });
        }
        stage_batched<1>(ntile * 32, tid, [&](int c) { return p.bias[min(c, rmax)]; }, [&](int c, float b) { sbias[c] = c <= rmax ? b : 0.f; });
        if constexpr (TAIL) {
            const half8* src = reinterpret_cast<const half8*>(p.dotw);
            stage_batched<2>(ntile * 2 * 32, tid, [&](int v) { return src[v]; }, [&](int v, half8 x) { reinterpret_cast<half8*>(swb)[v] = x; });      // 2 slices per cout tile x 32 vectors
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
    float16v acc2[2];
    if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
    }
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
        if constexpr (TAIL) {
            // stage A's epilogue in place (bias -> activation -> fp16, the pixel-shuffle store of conv_epilogue_tile), the fp16 values on
            // into stage B
            half8 wb[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const half8 t = *reinterpret_cast<const half8*>(swb + (((2 * j + g) * 2 + fj) * 16 + (fx & 15)) * 8);
                wb[g] = fx < 16 ? t : half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[i][e] + bias[e];
                vse_act_n(v, p.act, p.act_a, p.act_b);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    half8 y;
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (half_t)v[g * 8 + e];
                    const int c0 = j * 32 + g * 16 + fj * 8;
                    if (mm[i] < p.M && c0 < p.Np) {
                        const int quad = c0 / p.coutp;
                        const long opix = (nn[i] * (2 * p.OH) + 2 * oh[i] + (quad >> 1)) * (2L * p.OW) + 2 * ow[i] + (quad & 1);
                        *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.out) + opix * p.out_ld + (c0 - quad * p.coutp)) = y;
                    }
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[g], y, acc2[i], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (mm[i] < p.M) conv_epilogue_tile(p, acc[i], bias, mm[i], nn[i], oh[i], ow[i], j * 32, lane);
        }
    }
    if constexpr (TAIL) {
        // stage B's 16 outputs = the 4 x 4 block of u under this input pixel; this lane half's registers 0 .. 7 = rows 2 h, 2 h + 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (mm[i] >= p.M) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc2[i][e] + p.dotb;
            vse_act_n<8>(v, p.dotact, 0.f, 0.f);
            half_t* um = reinterpret_cast<half_t*>(p.dot_out) + ((nn[i] * (4 * p.OH) + 4 * oh[i] + 2 * fj) * (4L * p.OW) + 4 * ow[i]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                *reinterpret_cast<half4*>(um + (long)rr * 4 * p.OW) = half4{(half_t)v[4 * rr], (half_t)v[4 * rr + 1], (half_t)v[4 * rr + 2], (half_t)v[4 * rr + 3]};
        }
    }
}

template <int KS, bool HILO = false>
__global__ __launch_bounds__(256, (KS <= 2 ? 4 : KS <= 4 ? 3 : 2)) void conv_pw_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) half_t swt[(HILO ? 2 * PW_MAXN_HILO : PW_MAXN) * (KS * 16 + 8)];
    __shared__ float sbias[PW_MAXN];
    conv_pw_body<KS, HILO, false>(p, swt, sbias, nullptr);
}
template <int KS>
__global__ __launch_bounds__(256, 3) void conv_pw_tail_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) half_t swt[PW_MAXN * (KS * 16 + 8)];
    __shared__ float sbias[PW_MAXN];
    __shared__ __attribute__((aligned(16))) half_t swb[PW_MAXN / 16 * 2 * 16 * 8];      // 8 KiB
    conv_pw_body<KS, false, true>(p, swt, sbias, swb);
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
    if (p.flags & F_TAIL2) {
        // stage A without residual / gate / affine / second activation, whole 32-cout tiles, a dense fp16 map out
        if ((p.flags & (F_HILO | F_RES | F_OGATE | F_DOT1 | F_ONECH)) || !(p.flags & F_PIXSHUF) || p.out_f32 || !p.vec16 || (p.Np & 31) || !p.dotw || !p.dot_out
            || p.dot_f32 || p.dot_ld != 1 || p.act2 || p.post_a != 1.f || p.post_b != 0.f || p.lo_off || p.wl_out
            || (reinterpret_cast<uintptr_t>(p.dot_out) & 7) || (reinterpret_cast<uintptr_t>(p.dotw) & 15)) return VSE_E_INVAL;
        switch ((p.cinp + 15) / 16) {
            case 2: hipLaunchKernelGGL((conv_pw_tail_kernel<2>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((conv_pw_tail_kernel<4>), grid, block, 0, st, p); break;
            default: return VSE_E_UNSUPPORTED;
        }
        return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
    }
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
