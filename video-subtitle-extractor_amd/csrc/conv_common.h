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
    int nkh;                // F_HILO: K tiles of ONE pass over the taps (nk = 2 * nkh: hi weights, then lo weights); else = nk
    int out_ld, out_f32;
    int res_ld, resshift, res_hs, res_ws;
    int act, act2;
    float act_a, act_b, post_a, post_b;
    int flags, coutp;
    unsigned ntn;       // number of cout tiles
    unsigned ntiles;    // conv_c3w_kernel: pixel tiles of the launch (the persistent blocks share them out)
    int tiles_h, tiles_w;   // patch kernel: output tile grid per image
    const float* dotw;      // F_DOT1: per-cout weights of the fused 1-channel projection
    float dotb;
    int dotact, dot_f32, dot_ld;
    void* dot_out;
    const half_t* in2;      // F_SRC2: channels [nv0*8, cinp) come from this tensor (own pixel grid / shift / stride)
    int in2_ld, in2_shift, in2_hs, in2_ws, nv0;
    unsigned long long* trace;   // -DVSE_TRACE builds only: per-block phase stamps
    int vec16;              // output (and residual) rows allow 16-byte accesses at every 8-channel group
    long wimg_stride;       // F_IMGW: weight elements per image (Kp * Np); M tiles are then aligned to images
    int hw_img, tiles_img;  // F_IMGW: output pixels per image, M tiles per image
    const int* wl_out;      // ragged plans: per-image output width; pixels at ow >= wl_out[n] are stored as zeros
    int in_lo_off;          // F_DWPRE: != 0: the INPUT is an fp16 hi + lo pair (both halves are filtered)
    int res_lo_off;         // != 0: the residual is an fp16 hi + lo pair: its lo half sits res_lo_off channels behind the hi half
    int lo_off;             // != 0: fp16 hi + lo pair output: fp16(v - fp16(v)) goes lo_off channels behind the hi value
    const half_t* ogate;    // F_OGATE: per-(image, cout) gate [n][ogate_ld] fp16; the value is multiplied by (1 + gate) ahead of the residual
    int ogate_ld;
    int wnp;                // conv_c3_kernel: weight rows per tap of a ring stage (= Np; F_HLSUM: 64 = hi 32 | lo 32 while Np stays 32)
    const uint8_t* u8src;   // F_U8SRC (stem): uint8 BGR frames [n][u8_h][u8_w][3], row pitch / frame stride in bytes
    int u8_h, u8_w;
    long u8_pitch, u8_fstride;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 16-byte LDS-DMA per lane: global -> LDS without touching VGPRs.  The LDS destination of a wave instruction
// is wave-uniform base + lane*16 (1 KiB), so any bank-conflict swizzle is applied on the SOURCE side.
__device__ __forceinline__ void glds16(const void* g, half_t* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// The same DMA as an asm statement, hidden from hipcc's s_waitcnt bookkeeping: with the builtin inside a loop, hipcc
// (ROCm 7.2) drains lgkmcnt to 0 in front of every DMA and degrades every LDS wait of the loop to lgkmcnt(0), which
// defeats fragment prefetching.  The caller counts completion itself (s_waitcnt vmcnt(N) + barrier before the ds_reads)
// and guarantees that no ds_read of the destination is outstanding.  M0 (destination base) is saved and restored.
__device__ __forceinline__ void glds16_asm(const void* g, half_t* l) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)l);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// Buffer-addressed LDS-DMA helpers (conv_gemm.hip): an offset with bit 31 set is out of range for the
// 2 GiB descriptors these kernels build, so the DMA writes zeros.
#define OOB 0x80000000u
typedef __attribute__((address_space(3))) void* ldsv_t;
typedef int rsrc4_t __attribute__((ext_vector_type(4)));
// raw buffer descriptor over [base, base + 2 GiB): stride 0, num_records 0x7fffffff, dword3 0x00020000 (what
// __builtin_amdgcn_make_buffer_rsrc builds), as four SGPRs an asm statement can take
__device__ __forceinline__ rsrc4_t make_rsrc4(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    return rsrc4_t{(int)__builtin_amdgcn_readfirstlane((unsigned)b), (int)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu),
                   0x7fffffff, 0x00020000};
}
// `buffer_load_dwordx4 voff, rsrc, soff offen lds` as an asm statement (see glds16_asm for why); `l` is the wave-uniform
// LDS destination.  The leading s_nop covers a descriptor / soffset SGPR freshly written by v_readfirstlane.
__device__ __forceinline__ void bufdma16_asm(rsrc4_t rsrc, unsigned voff, int soff, const void* l) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)l);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 16, "vmcnt literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// Accumulator tile layout.  v_mfma_f32_32x32x16_f16 leaves lane l with rows 8q + 4(l>>5) + e (q, e in 0..3; register
// 4q + e) of column l & 31.  Columns are pixels; rows are couts THROUGH THE PERMUTATION "swap bits 2 and 3": the lane
// that supplies weight row f of a 32-cout tile reads cout conv_wrow(f), so lane l ends up with the 16 couts
//     cbase + 16g + 8(l>>5) + {0..7},  g = 0, 1      (registers 8g .. 8g+7 in order)
// i.e. two runs of 8 consecutive channels = 16-byte NHWC stores (the natural order gives 4-channel / 8-byte runs, and
// the 8-byte partial-line writes cost ~30 % of a whole 1x1 layer).  The permutation keeps the weight-fragment
// ds_read_b128 bank-conflict free under both LDS swizzles (64-byte and 128-byte rows).
// Flattened output pixel -> (image, row, column) by 32-BIT unsigned divisions: a 64-bit division by a runtime value expands to ~100
// instructions, and the small-K streaming kernels (conv_pw, conv_dwpw) do two of them per 32-pixel tile.  Their launchers refuse M >= 2^31.
__device__ __forceinline__ void conv_pix_coords(const ConvParams& p, long m, long& n, int& oh, int& ow) {
    const unsigned mu = (unsigned)m, t = mu / (unsigned)p.OW, nn = t / (unsigned)p.OH;
    ow = (int)(mu - t * (unsigned)p.OW);
    oh = (int)(t - nn * (unsigned)p.OH);
    n = (long)nn;
}

__device__ __forceinline__ int conv_wrow(int f) { return (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1); }

// Per-cout epilogue constants (bias, F_DOT1 projection weights) are staged ONCE per block in LDS (conv_stage_consts;
// visible after the K loop's first wait + barrier) and read back per accumulator
// tile in the lane's register order.  Reading them from global memory inside the epilogue costs one dependent memory
// round trip per 8 couts (s_memtime trace: 7 us of a 20 us tile on the detector's last layer); holding them in
// registers across the K loop costs the occupancy the ring was sized for.
// The staging itself is a 4-byte-per-lane LDS-DMA issued BEFORE the prologue DMAs: it is then the oldest entry of the
// issuing wave's vmcnt queue, so every counted wait of the K loop covers it without changing a literal, no VGPR is
// involved and the compiler adds no wait of its own.  Wave w stages couts 64w .. 64w+63 of the block's cout tile.
// ASM = true issues it as an asm statement (kernels whose other DMAs are glds16_asm: ONE builtin LDS-DMA anywhere in a
// kernel is enough for hipcc to drop counted lgkmcnt waits everywhere in it).
template <bool ASM = false>
__device__ __forceinline__ void conv_stage_consts(float* dst, const float* src, const half_t* zero, int n0, int bn, int Np,
                                                  int wave, int lane) {
    if (wave >= 0 && wave * 64 < bn) {
        const int c = wave * 64 + lane;
        const void* g = (c < bn && n0 + c < Np) ? (const void*)(src + n0 + c) : (const void*)zero;
        if constexpr (ASM) {
            unsigned keep;
            const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)(dst + wave * 64));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(d) : "memory");
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + wave * 64), 4, 0, 0);
        }
    }
}
// `tab` = the staged table of this block's cout tile, `c` = first cout of the 32-cout accumulator tile inside it
__device__ __forceinline__ void conv_epilogue_consts(const float* tab, int c, int lane, float (&out)[16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* q = tab + c + g * 16 + (lane >> 5) * 8;
        const float4v a = *reinterpret_cast<const float4v*>(q), b = *reinterpret_cast<const float4v*>(q + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { out[g * 8 + e] = a[e]; out[g * 8 + 4 + e] = b[e]; }
    }
}

// Epilogue of one 32(cout) x 32(pixel) accumulator tile: lane l owns pixel (l & 31) — passed in as (m, n, oh, ow).
//   + bias (BN folded) -> activation -> scalar affine -> (+ residual, optionally nearest-upsampled) -> activation2
//   -> fp16 / fp32 store; F_PIXSHUF scatters a 2x2-stride-2 transposed conv.
__device__ __forceinline__ void conv_epilogue_tile(const ConvParams& p, const float16v& acc, const float (&bias)[16], long m,
                                                   long n, int oh, int ow, int cbase, int lane) {
    const bool pixshuf = p.flags & F_PIXSHUF;
    const bool has_res = p.flags & F_RES;
    long res_pix = m;
    if (has_res && p.resshift) res_pix = (n * p.res_hs + (oh >> p.resshift)) * p.res_ws + (ow >> p.resshift);
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = acc[e] + bias[e];
    vse_act_n(v, p.act, p.act_a, p.act_b);
#ifndef VSE_EPI_SKIP
#define VSE_EPI_SKIP 1
#endif
    // the scalar affine after the activation is the identity for all but a handful of layers: one uniform branch instead of
    // 16 multiply-adds per accumulator tile (the epilogue is VALU-bound)
    if (!VSE_EPI_SKIP || p.post_a != 1.f || p.post_b != 0.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * p.post_a + p.post_b;
    }
    long opix[2];
    int oc[2];
    bool live[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int c0 = cbase + g * 16 + (lane >> 5) * 8;
        live[g] = c0 < p.Np;
        opix[g] = m;
        oc[g] = c0;
        if (pixshuf) {                                   // coutp % 8 == 0: a run of 8 never straddles two quads
            const int quad = c0 / p.coutp;
            oc[g] = c0 - quad * p.coutp;
            opix[g] = (n * (2 * p.OH) + 2 * oh + (quad >> 1)) * (2L * p.OW) + 2 * ow + (quad & 1);
        }
    }
    if (p.ogate != nullptr) {
        // an SE block with shortcut behind a 1x1 conv, x + x * gate(mean(x)), folded into the conv: the gate comes from the mean
        // of the conv's INPUT (the mean commutes with a 1x1 conv), so x itself is never written (compiler.py _rewrite_se_laterals)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (!live[g]) continue;
            const half8 g8 = *reinterpret_cast<const half8*>(p.ogate + n * p.ogate_ld + oc[g]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g * 8 + e] *= 1.0f + (float)g8[e];
        }
    }
    if (has_res) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (!live[g]) continue;
            const half_t* rp = p.res + res_pix * p.res_ld + oc[g];
            if (p.vec16) {
                const half8 r8 = *reinterpret_cast<const half8*>(rp);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[g * 8 + e] += (float)r8[e];
                if (p.res_lo_off) {
                    const half8 l8 = *reinterpret_cast<const half8*>(rp + p.res_lo_off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[g * 8 + e] += (float)l8[e];
                }
            } else {
                const half4 r0 = *reinterpret_cast<const half4*>(rp), r1 = *reinterpret_cast<const half4*>(rp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[g * 8 + e] += (float)r0[e]; v[g * 8 + 4 + e] += (float)r1[e]; }
            }
        }
    }
    vse_act_n(v, p.act2, 0.f, 0.f);
    if (p.wl_out != nullptr && ow >= p.wl_out[n]) {
        // ragged batch: this pixel lies right of its sample's own width — the next layer must see what zero padding would
        // have given it there
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
    }
    if (p.flags & F_ONECH) {
        // pixel-shuffle conv to ONE channel, fp32 map out (ld = 1): this lane's 8-channel run g is one quad; its first value is the pixel
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (live[g]) reinterpret_cast<float*>(p.out)[opix[g]] = v[g * 8];
        return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!live[g]) continue;
        const float* w = v + g * 8;
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + opix[g] * p.out_ld + oc[g];
            *reinterpret_cast<float4v*>(op) = float4v{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<float4v*>(op + 4) = float4v{w[4], w[5], w[6], w[7]};
        } else {
            half_t* op = reinterpret_cast<half_t*>(p.out) + opix[g] * p.out_ld + oc[g];
            if (p.vec16) {
                const half8 hi8 = half8{(half_t)w[0], (half_t)w[1], (half_t)w[2], (half_t)w[3],
                                        (half_t)w[4], (half_t)w[5], (half_t)w[6], (half_t)w[7]};
                *reinterpret_cast<half8*>(op) = hi8;
                if (p.lo_off) {          // the tensor feeds an OP_CHAIN: what fp16 dropped travels beside it (launch_conv checks vec16)
                    half8 lo8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) lo8[e] = (half_t)(w[e] - (float)hi8[e]);
                    *reinterpret_cast<half8*>(op + p.lo_off) = lo8;
                }
            } else {
                *reinterpret_cast<half4*>(op) = half4{(half_t)w[0], (half_t)w[1], (half_t)w[2], (half_t)w[3]};
                *reinterpret_cast<half4*>(op + 4) = half4{(half_t)w[4], (half_t)w[5], (half_t)w[6], (half_t)w[7]};
            }
        }
    }
}

// Ragged batches: an output tile that lies entirely right of its sample's width holds zeros by definition — the block writes
// them (one 16-byte store per pixel and 8-channel group) and skips its prologue, K loop and epilogue.  Call before the first
// DMA / barrier; the decision is block-uniform.  TH x TW = the tile, bn = couts of the block's tile.
template <int TH, int TW>
__device__ __forceinline__ bool conv_tile_right_of_sample(const ConvParams& p, long img, int oy0, int ox0, int n0, int bn) {
    if (p.wl_out == nullptr || p.out_f32 || (p.flags & (F_DOT1 | F_PIXSHUF)) || !p.vec16) return false;
    if (ox0 < p.wl_out[img]) return false;
    const int c1 = min(n0 + bn, p.Np);
    for (int i = threadIdx.x; i < TH * TW; i += blockDim.x) {
        const int oy = oy0 + i / TW, ox = ox0 + i % TW;
        if (oy >= p.OH || ox >= p.OW) continue;
        half_t* op = reinterpret_cast<half_t*>(p.out) + ((img * p.OH + oy) * p.OW + ox) * (long)p.out_ld;
        for (int c = n0; c < c1; c += 8) *reinterpret_cast<half8*>(op + c) = half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    return true;
}

// F_DOT1 variant: returns this lane's partial  sum_c y[c] * dotw[c]  over the couts it owns in one accumulator tile
// (y = the full epilogue value); nothing is stored.  Padded couts carry zero weights.
__device__ __forceinline__ float conv_epilogue_dot(const ConvParams& p, const float16v& acc, const float (&bias)[16],
                                                   const float (&dotw)[16]) {
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = acc[e] + bias[e];
    vse_act_n(v, p.act, p.act_a, p.act_b);
    if (!VSE_EPI_SKIP || p.post_a != 1.f || p.post_b != 0.f) {        // as conv_epilogue_tile: the affine is the identity almost everywhere
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * p.post_a + p.post_b;
    }
    vse_act_n(v, p.act2, 0.f, 0.f);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) part += v[e] * dotw[e];
    return part;
}

int launch_conv_patch(const ConvParams& p, int n_img, hipStream_t st);
// one filter column per step over a 16-channel patch (conv_col.hip, F_COL): 9x9 / 7x7 / 5x5, <= 64 couts
int launch_conv_col(const ConvParams& p, int n_img, hipStream_t st);
int launch_conv_col3w(const ConvParams& p, int n_img, hipStream_t st);
// depthwise k x k conv fused in front of a 1x1 conv (conv_dwpw.hip, F_DWPRE): p.kh / sh / ph = the depthwise geometry, p.dotw = its table
int launch_conv_dwpw(const ConvParams& p, hipStream_t st);
bool conv_dwpw_ok(int k, int s, int cinp, int Np, int flags);
int conv_dwpw_rows_stride(int k, int pad, int s, int cinp, int lo_in);      // stride of the row-streaming form (conv_dwpw_rows_kernel), 0 = the tile form
// 3x3 sibling, two blocks per CU (conv_c3.hip, F_COL with kh = kw = 3)
int launch_conv_c3(const ConvParams& p, int n_img, hipStream_t st);
double conv_c3_plan(int OH, int OW, int* rw_out);
// all couts per block, one persistent block per CU (conv_c3w.hip): the 3x3 layers with 128 couts
int launch_conv_c3w(const ConvParams& p, int n_img, hipStream_t st);
bool conv_c3w_ok(const ConvParams& p);
bool conv_c3_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int flags);
// pointwise conv over <= 64 input channels and <= 64 couts, no LDS staging (conv_pw.hip, F_PW)
int launch_conv_pw(const ConvParams& p, hipStream_t st);
bool conv_pw_ok(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Np, int inshift, int flags);
int conv_col_bn(int Np);
bool conv_col_ok(int kh, int kw, int sh, int sw, int cinp, int Np, int flags);
// DB head evaluated on the low-resolution grid (conv_head.hip, F_UP2HEAD)
int launch_conv_head_up2(const ConvParams& p, int n_img, hipStream_t st);
// 3x3 stem over <= 4 real input channels (conv_stem.hip, F_STEM)
int launch_conv_stem(const ConvParams& p, int n_img, hipStream_t st);
// scalar-addressed implicit GEMM (conv_gemm.hip): VSE_E_UNSUPPORTED when the layer is not eligible
int launch_conv_gemm(ConvParams& p, int Kp, hipStream_t st);
int conv_gemm_config(int Np, int cinp, long M);
// 1x1 conv over <= 256 pixels, operands straight from global memory (conv_smallm.hip); `mode` = conv_gemm_mode()
bool conv_smallm_ok(const ConvParams& p, int mode);
bool conv_smallm_shape_ok(int mode, long M, int sh, int sw, int same_hw, int flags, int cinp);
bool conv_smallk_ok(const ConvParams& p);
bool conv_smallk_shape_ok(int kh, int kw, int sh, int sw, int ph, int pw, int inshift, int same_hw, int flags, int cinp, long M, int Np);
int launch_conv_smallm(const ConvParams& p, hipStream_t st);   // index into the tile-configuration table of conv_gemm.hip
int conv_gemm_mode(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Kp, int inshift, int flags);
int conv_patch_th(int kh, int kw, int OH, int bn);
int conv_patch_bn(int Np);
void conv_patch_plan(int kh, int kw, int OH, int Np, int flags, int* th, int* bn, int* mode);
