// Shared by the implicit-GEMM conv kernel (conv_mfma.hip) and the LDS-resident-patch conv kernel (conv_patch.hip).
#pragma once
#include "common.h"

struct ConvParams {
    const half_t* in;
    const half_t* w;
    const float* bias;
    const half_t* res;
    const half_t* zero;      // 4 KiB of zeros: gather target for padding / out-of-range lanes
    void* out;
    int H, W, Hs, Ws, in_ld, cinp, inshift;
    int OH, OW;
    long M;
    int kh, kw, sh, sw, ph, pw;
    int Np, nk;
    int out_ld, out_f32;
    int res_ld, resshift, res_hs, res_ws;
    int act, act2;
    float act_a, act_b, post_a, post_b;
    int flags, coutp;
    unsigned ntn;       // number of cout tiles
    unsigned ntiles;    // persistent kernels: total output tiles
    int tiles_h, tiles_w;   // patch kernel: output tile grid per image
    const float* dotw;      // F_DOT1: per-cout weights of the fused 1-channel projection
    float dotb;
    int dotact, dot_f32, dot_ld;
    void* dot_out;
    const half_t* in2;      // F_SRC2: channels [nv0*8, cinp) come from this tensor (own pixel grid / shift / stride)
    int in2_ld, in2_shift, in2_hs, in2_ws, nv0;
    int vec16;              // output (and residual) rows allow 16-byte accesses at every 8-channel group
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 16-byte LDS-DMA per lane: global -> LDS without touching VGPRs.  The LDS destination of a wave instruction
// is wave-uniform base + lane*16 (1 KiB), so any bank-conflict swizzle is applied on the SOURCE side.
__device__ __forceinline__ void glds16(const void* g, half_t* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// Accumulator tile layout.  v_mfma_f32_32x32x16_f16 leaves lane l with rows 8q + 4(l>>5) + e (q, e in 0..3; register
// 4q + e) of column l & 31.  Columns are pixels; rows are couts THROUGH THE PERMUTATION "swap bits 2 and 3": the lane
// that supplies weight row f of a 32-cout tile reads cout conv_wrow(f), so lane l ends up with the 16 couts
//     cbase + 16g + 8(l>>5) + {0..7},  g = 0, 1      (registers 8g .. 8g+7 in order)
// i.e. two runs of 8 consecutive channels = 16-byte NHWC stores (the natural order gives 4-channel / 8-byte runs, and
// the 8-byte partial-line writes cost ~30 % of a whole 1x1 layer).  The permutation keeps the weight-fragment
// ds_read_b128 bank-conflict free under both LDS swizzles (64-byte and 128-byte rows).
__device__ __forceinline__ int conv_wrow(int f) { return (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1); }

// Epilogue of one 32(cout) x 32(pixel) accumulator tile: lane l owns pixel (l & 31) — passed in as (m, n, oh, ow).
//   + bias (BN folded) -> activation -> scalar affine -> (+ residual, optionally nearest-upsampled) -> activation2
//   -> fp16 / fp32 store; F_PIXSHUF scatters a 2x2-stride-2 transposed conv.
__device__ __forceinline__ void conv_epilogue_tile(const ConvParams& p, const float16v& acc, long m, long n, int oh,
                                                   int ow, int cbase, int lane) {
    const bool pixshuf = p.flags & F_PIXSHUF;
    const bool has_res = p.flags & F_RES;
    long res_pix = m;
    if (has_res && p.resshift) res_pix = (n * p.res_hs + (oh >> p.resshift)) * p.res_ws + (ow >> p.resshift);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int c0 = cbase + g * 16 + (lane >> 5) * 8;
        if (c0 >= p.Np) continue;
        const float4v b0 = *reinterpret_cast<const float4v*>(p.bias + c0);
        const float4v b1 = *reinterpret_cast<const float4v*>(p.bias + c0 + 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = acc[g * 8 + e] + (e < 4 ? b0[e & 3] : b1[e & 3]);
            x = vse_act(x, p.act, p.act_a, p.act_b);
            v[e] = x * p.post_a + p.post_b;
        }
        long opix = m;
        int oc = c0;
        if (pixshuf) {                                   // coutp % 8 == 0: a run of 8 never straddles two quads
            const int quad = c0 / p.coutp;
            oc = c0 - quad * p.coutp;
            opix = (n * (2 * p.OH) + 2 * oh + (quad >> 1)) * (2L * p.OW) + 2 * ow + (quad & 1);
        }
        if (has_res) {
            const half_t* rp = p.res + res_pix * p.res_ld + oc;
            if (p.vec16) {
                const half8 r8 = *reinterpret_cast<const half8*>(rp);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
            } else {
                const half4 r0 = *reinterpret_cast<const half4*>(rp), r1 = *reinterpret_cast<const half4*>(rp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += (float)r0[e]; v[e + 4] += (float)r1[e]; }
            }
        }
        if (p.act2 != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vse_act(v[e], p.act2, 0.f, 0.f);
        }
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + opix * p.out_ld + oc;
            *reinterpret_cast<float4v*>(op) = float4v{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4v*>(op + 4) = float4v{v[4], v[5], v[6], v[7]};
        } else {
            half_t* op = reinterpret_cast<half_t*>(p.out) + opix * p.out_ld + oc;
            if (p.vec16) {
                *reinterpret_cast<half8*>(op) = half8{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3],
                                                      (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
            } else {
                *reinterpret_cast<half4*>(op) = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4*>(op + 4) = half4{(half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
            }
        }
    }
}

// F_DOT1 variant: returns this lane's partial  sum_c y[c] * dotw[c]  over the couts it owns in one accumulator tile
// (y = the full epilogue value); nothing is stored.
__device__ __forceinline__ float conv_epilogue_dot(const ConvParams& p, const float16v& acc, int cbase, int lane) {
    float part = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int c0 = cbase + g * 16 + (lane >> 5) * 8;
        if (c0 >= p.Np) continue;
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const float4v b4 = *reinterpret_cast<const float4v*>(p.bias + c0 + 4 * h4);
            const float4v w4 = *reinterpret_cast<const float4v*>(p.dotw + c0 + 4 * h4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[g * 8 + h4 * 4 + e] + b4[e];
                x = vse_act(x, p.act, p.act_a, p.act_b) * p.post_a + p.post_b;
                x = vse_act(x, p.act2, 0.f, 0.f);
                part += x * w4[e];
            }
        }
    }
    return part;
}

int launch_conv_patch(const ConvParams& p, int n_img, hipStream_t st);
// scalar-addressed implicit GEMM (conv_gemm.hip): VSE_E_UNSUPPORTED when the layer is not eligible
int launch_conv_gemm(ConvParams& p, int Kp, hipStream_t st);
int conv_gemm_config(int Np, int cinp, long M);   // index into the tile-configuration table of conv_gemm.hip
int conv_gemm_mode(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Kp, int inshift, int flags);
int conv_patch_th(int kh, int kw, int OH, int bn);
int conv_patch_bn(int Np);
