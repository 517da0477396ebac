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
    int tiles_h, tiles_w;   // patch kernel: output tile grid per image
    const float* dotw;      // F_DOT1: per-cout weights of the fused 1-channel projection
    float dotb;
    int dotact, dot_f32, dot_ld;
    void* dot_out;
    const half_t* in2;      // F_SRC2: channels [nv0*8, cinp) come from this tensor (own pixel grid / shift / stride)
    int in2_ld, in2_shift, in2_hs, in2_ws, nv0;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 16-byte LDS-DMA per lane: global -> LDS without touching VGPRs.  The LDS destination of a wave instruction
// is wave-uniform base + lane*16 (1 KiB), so any bank-conflict swizzle is applied on the SOURCE side.
__device__ __forceinline__ void glds16(const void* g, half_t* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// Epilogue of one 32(cout) x 32(pixel) accumulator tile held MFMA-style: lane l owns pixel (l & 31) — passed in as
// (m, n, oh, ow) — and couts cbase + 8*q + 4*(l>>5) + e for q,e in 0..3.
//   + bias (BN folded) -> activation -> scalar affine -> (+ residual, optionally nearest-upsampled) -> activation2
//   -> fp16 / fp32 store; F_PIXSHUF scatters a 2x2-stride-2 transposed conv.
__device__ __forceinline__ void conv_epilogue_tile(const ConvParams& p, const float16v& acc, long m, long n, int oh,
                                                   int ow, int cbase, int lane) {
    const bool pixshuf = p.flags & F_PIXSHUF;
    const bool has_res = p.flags & F_RES;
    long res_pix = m;
    if (has_res && p.resshift) res_pix = (n * p.res_hs + (oh >> p.resshift)) * p.res_ws + (ow >> p.resshift);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = cbase + q * 8 + (lane >> 5) * 4;
        if (c0 >= p.Np) continue;
        const float4v b4 = *reinterpret_cast<const float4v*>(p.bias + c0);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[q * 4 + e] + b4[e];
            x = vse_act(x, p.act, p.act_a, p.act_b);
            v[e] = x * p.post_a + p.post_b;
        }
        long opix = m;
        int oc = c0;
        if (pixshuf) {
            const int quad = c0 / p.coutp;
            oc = c0 - quad * p.coutp;
            opix = (n * (2 * p.OH) + 2 * oh + (quad >> 1)) * (2L * p.OW) + 2 * ow + (quad & 1);
        }
        if (has_res) {
            const half4 r4 = *reinterpret_cast<const half4*>(p.res + res_pix * p.res_ld + oc);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
        }
        if (p.act2 != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = vse_act(v[e], p.act2, 0.f, 0.f);
        }
        if (p.out_f32) {
            float4v o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4v*>(reinterpret_cast<float*>(p.out) + opix * p.out_ld + oc) = o;
        } else {
            half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.out) + opix * p.out_ld + oc) = o;
        }
    }
}

// F_DOT1 variant: returns this lane's partial  sum_c y[c] * dotw[c]  over the couts it owns in one accumulator tile
// (y = the full epilogue value); nothing is stored.
__device__ __forceinline__ float conv_epilogue_dot(const ConvParams& p, const float16v& acc, int cbase, int lane) {
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = cbase + q * 8 + (lane >> 5) * 4;
        if (c0 >= p.Np) continue;
        const float4v b4 = *reinterpret_cast<const float4v*>(p.bias + c0);
        const float4v w4 = *reinterpret_cast<const float4v*>(p.dotw + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[q * 4 + e] + b4[e];
            x = vse_act(x, p.act, p.act_a, p.act_b) * p.post_a + p.post_b;
            x = vse_act(x, p.act2, 0.f, 0.f);
            part += x * w4[e];
        }
    }
    return part;
}

int launch_conv_patch(const ConvParams& p, int n_img, hipStream_t st);
int conv_patch_th(int kh, int kw, int OH, int bn);
int conv_patch_bn(int Np);
