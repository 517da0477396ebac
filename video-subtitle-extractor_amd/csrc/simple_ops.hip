// HBM-bound NHWC fp16 kernels of the engine: depthwise conv, pooling, global average, SE gating, adds,
// nearest resize / concat copies, unary affine+activation, LayerNorm, fused SVTR attention, class softmax with
// arg-max, LSTM recurrence.  One thread handles one 16-byte (8-channel) vector of one pixel wherever the
// layout allows it, so a wave reads/writes 1 KiB per instruction with consecutive lanes on consecutive
// channel groups of the same pixel (coalesced NHWC).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#define GRID_CAP 16384

static inline int grid_for(long items, int block) {
    long g = (items + block - 1) / block;
    if (g > GRID_CAP) g = GRID_CAP;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ half8 ld8(const TView& v, long pix, int c) {
    return *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(v.ptr) + pix * v.ld + c);
}
__device__ __forceinline__ void st8(const TView& v, long pix, int c, half8 x) {
    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(v.ptr) + pix * v.ld + c) = x;
}

// ------------------------------------------------------------------------------------------------ depthwise
// gate.ptr != nullptr (F_GATE): the input is the un-gated tensor of an SE block; x * gate[n, c] is rounded to fp16 on load,
// exactly what the separate scale pass would have stored.
// mode 0: off, 1: x * g, 2: x * g + x (residual SE, OP_SCALE with F_RES)
__device__ __forceinline__ half8 dw_gate(half8 x, const half8& g, int mode) {
    if (mode) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)x[e] * (float)g[e];
            x[e] = (half_t)(mode == 2 ? v + (float)x[e] : v);
        }
    }
    return x;
}
// Compile-time gate mode for the row kernel: 1 = x * g as FOUR packed fp16 multiplies per vector (the product of two fp16 values is
// exact in fp32, so one rounding to fp16 either way; the fp32 form costs 24 conversions / multiplies per vector and made the gated
// stage-transition conv of the detector VALU-bound: 0.58 vs 0.34 ms); a run-time switch between the forms cost the UNGATED kernel
// 60 % (0.34 -> 0.55 ms: register allocation), hence the template parameter.
template <int GM> __device__ __forceinline__ half8 dw_gate_t(half8 x, const half8& g) {
    if constexpr (GM == 1) return x * g;
    else if constexpr (GM == 2) return dw_gate(x, g, 2);
    else return x;
}
__global__ __launch_bounds__(256) void dwconv_kernel(TView in, TView out, TView gate, int gmode, int hilo, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int kh, int kw, int sh, int sw,
                                                     int ph, int pw, int act, float act_a, float act_b, float post_a,
                                                     float post_b, const int* __restrict__ wl_out) {
    const int gated = gate.ptr != nullptr ? gmode : 0;
    const int lo_off = ((hilo >> 1) & 0xfff) << 3;   // != 0: the output is an fp16 hi + lo pair (P_LO_OUT): fp16(v - fp16(v)) lo_off channels behind
    const int lo_in = ((hilo >> 13) & 0xfff) << 3;   // != 0: the INPUT is a pair: its lo half sits lo_in channels behind the hi half
    hilo &= 1;
    const int cg = in.c >> 3;
    const long total = (long)out.n * out.h * out.w * cg;
    for (long i = xcd_block(blockIdx.x, gridDim.x) * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        long pix = vse_divmod(i, cg, g);
        int ow;
        long t = vse_divmod(pix, out.w, ow);
        int oh;
        const long n = vse_divmod(t, out.h, oh);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias[g * 8 + e];
        const half8 gv = gated ? ld8(gate, n, g * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < kh; ++dy) {
            const int ih = oh * sh - ph + dy;
            if (ih < 0 || ih >= in.h) continue;
            for (int dx = 0; dx < kw; ++dx) {
                const int iw = ow * sw - pw + dx;
                if (iw < 0 || iw >= in.w) continue;
                const half8 x = dw_gate(ld8(in, (n * in.h + ih) * in.w + iw, g * 8), gv, gated);
                const half8 xl = lo_in ? ld8(in, (n * in.h + ih) * in.w + iw, g * 8 + lo_in) : half8{0, 0, 0, 0, 0, 0, 0, 0};
                // (the filter table is fp32 [kh * kw][C]: the compiler stores fp16(w) — or fp16 hi + fp16 lo with F_HILO — as ONE fp32 value)
                const float4v k0 = *reinterpret_cast<const float4v*>(w + (long)(dy * kw + dx) * in.c + g * 8);
                const float4v k1 = *reinterpret_cast<const float4v*>(w + (long)(dy * kw + dx) * in.c + g * 8 + 4);
                vse_fma_h8(acc, x, k0, k1);
                if (lo_in) vse_fma_h8(acc, xl, k0, k1);
            }
        }
        half8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = vse_act(acc[e], act, act_a, act_b) * post_a + post_b;
            o[e] = (half_t)v;
            ol[e] = (half_t)(v - (float)o[e]);
        }
        if (wl_out != nullptr && ow >= wl_out[n]) o = half8{0, 0, 0, 0, 0, 0, 0, 0};      // ragged batch: right of the sample's width
        st8(out, pix, g * 8, o);
        if (lo_off) st8(out, pix, g * 8 + lo_off, ol);
    }
}

#ifndef VSE_DWROW_ABL
#define VSE_DWROW_ABL 0
#endif
// Row-blocked variant: one thread produces FOUR consecutive output pixels of one row for one 8-channel group, so every
// input vector of a filter row is loaded once for the outputs that share it ((4-1)*SW + KW loads instead of 4*KW) and every
// weight vector once per tap instead of once per output.  Same accumulation order per output as dwconv_kernel (bias, then
// taps row-major) -> bit-identical results.  The mobile (PP-LCNetV3 / MobileNetV3) models spend half of their time here.
template <int KW, int SW, int GM>
__global__ __launch_bounds__(256) void dwconv_row_kernel(TView in, TView out, TView gate, int hilo, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int kh, int sh, int ph, int pw,
                                                         int act, float act_a, float act_b, float post_a, float post_b,
                                                         const int* __restrict__ wl_out) {
    constexpr int OUTW = 4, WIN = (OUTW - 1) * SW + KW;
    const int lo_off = ((hilo >> 1) & 0xfff) << 3, lo_in = ((hilo >> 13) & 0xfff) << 3;          // (see dwconv_kernel)
    hilo &= 1;
    const bool affine = post_a != 1.f || post_b != 0.f;
    const int cg = in.c >> 3;
    const int owq = (out.w + OUTW - 1) / OUTW;
    const long total = (long)out.n * out.h * owq * cg;
    for (long i = xcd_block(blockIdx.x, gridDim.x) * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        long t = vse_divmod(i, cg, g);
        int q;
        t = vse_divmod(t, owq, q);
        int oh;
        const long n = vse_divmod(t, out.h, oh);
        const int ow0 = q * OUTW, iw0 = ow0 * SW - pw;
        float acc[OUTW][8];
#pragma unroll
        for (int o = 0; o < OUTW; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = bias[g * 8 + e];
        const half8 gv = GM ? ld8(gate, n, g * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        // wave-uniform: every lane's WIN input columns lie inside the image (all waves but those that hold a first / last quad of a row):
        // the column tests of the loads and of the taps — ~550 of the 1660 vector instructions of a 5 x 5 thread, and the kernel is
        // VALU-bound — are compiled out of that path.  Same operations on the same values: bit-identical.
        const bool inside = __builtin_amdgcn_ballot_w64(!(iw0 >= 0 && iw0 + WIN - 1 < in.w)) == 0;
        auto rows = [&](auto chk) {
            constexpr bool CHK = decltype(chk)::value;
            for (int dyp = 0; dyp < (lo_in ? 2 * kh : kh); ++dyp) {
                // (a pair input: every filter row is walked twice, over the hi and over the lo half of the same pixels)
                const int dy = dyp < kh ? dyp : dyp - kh;
                const int coff = dyp < kh ? 0 : lo_in;
                const int ih = oh * sh - ph + dy;
                if (ih < 0 || ih >= in.h) continue;
                const long rowpix = (n * in.h + ih) * in.w;
                half8 x[WIN];
#pragma unroll
                for (int c = 0; c < WIN; ++c) {
                    const int iw = iw0 + c;
#if defined(VSE_DEV_BUILD) && VSE_DWROW_ABL == 2      // timing-only ablation: ONE gather per filter row (results wrong)
                    if (c > 0) { x[c] = x[0]; continue; }
#endif
                    x[c] = (!CHK || (iw >= 0 && iw < in.w)) ? ld8(in, rowpix + iw, g * 8 + coff) : half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                // all loads of the row first, arithmetic afterwards: multiplying each vector as it arrives serialises the loads
                // (0.52 ms gated against 0.32 ms with the loads batched on the detector's maps)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (GM != 0) {
#pragma unroll
                    for (int c = 0; c < WIN; ++c) x[c] = dw_gate_t<GM>(x[c], gv);
                }
#pragma unroll
                for (int dx = 0; dx < KW; ++dx) {
#if defined(VSE_DEV_BUILD) && VSE_DWROW_ABL == 1      // timing-only ablation: 1 / KW of the multiply-adds, every gather kept (results wrong)
                    if (dx > 0) { asm volatile("" :: "v"(x[dx]), "v"(x[WIN - 1])); continue; }
#endif
                    const float4v k0 = *reinterpret_cast<const float4v*>(w + (long)(dy * KW + dx) * in.c + g * 8);
                    const float4v k1 = *reinterpret_cast<const float4v*>(w + (long)(dy * KW + dx) * in.c + g * 8 + 4);
#pragma unroll
                    for (int o = 0; o < OUTW; ++o) {
                        const int iw = iw0 + o * SW + dx;
                        if (CHK && (iw < 0 || iw >= in.w)) continue;      // the reference kernel skips padded taps (no +0 rounding issue, same sums)
                        vse_fma_h8(acc[o], x[o * SW + dx], k0, k1);
                    }
                }
            }
        };
        if (inside) rows(std::false_type{});
        else rows(std::true_type{});
        // ONE activation switch for the thread's 32 values (a per-element switch is a chain of scalar branches per element)
#pragma unroll
        for (int o = 0; o < OUTW; ++o) vse_act_n<8>(acc[o], act, act_a, act_b);
#pragma unroll
        for (int o = 0; o < OUTW; ++o) {
            if (ow0 + o >= out.w) continue;
            half8 r, rl = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc[o][e];
                if (affine) v = v * post_a + post_b;          // (uniform: the identity for all but a handful of layers)
                r[e] = (half_t)v;
                if (lo_off) rl[e] = (half_t)(v - (float)r[e]);
            }
            if (wl_out != nullptr && ow0 + o >= wl_out[n]) r = half8{0, 0, 0, 0, 0, 0, 0, 0};
            st8(out, (n * out.h + oh) * out.w + ow0 + o, g * 8, r);
            if (lo_off) st8(out, (n * out.h + oh) * out.w + ow0 + o, g * 8 + lo_off, rl);
        }
    }
}

// Column-walk variant (round 6): one thread produces OUTW consecutive output pixels x TWO channels of every row of a segment of output
// rows, walking down the input rows once, with its KH x KW x 2 filter weights and every accumulator in REGISTERS.  An input row is gathered
// once (OUTW + KW - 1 dwords) and converted to fp32 once; its filter rows are applied to the (up to KH / SH) output rows it belongs to, each
// keeping its own fp32 accumulator in a ring of R = ceil(KH / SH) slots; the multiply-adds are v_pk_fma_f32 (both channels per instruction):
// the inner loop is 80 % multiply-adds, no LDS, no table reads.  dwconv_row_kernel gathers KH x (3 SW + KW) vectors, re-reads the table per
// tap and issues KH x KW x 8 v_fma_mix_f32 per output pixel; its ablations (round 6, V4_ch_rec_fast 56 x 896: 1 / KW of the multiply-adds
// -0.21 of 0.65 ms, one gather per row -0.13) show multiply-adds, gathers and per-item index arithmetic each a third of it.  (A first
// column-walk form with 8 channels per thread and the table in LDS — 50 ds_read_b128 per input row — was 15 % SLOWER than the row kernel.)
// Per accumulator the order is the row kernel's — bias, then taps row-major; rows outside the image skipped, columns outside read as zeros
// (adding 0 * w is the skip, bit for bit, unless an accumulator is exactly -0) — and an fp32 fma of the exactly converted fp16 value is what
// v_fma_mix_f32 computes: identical bits (development build: tools/ab_env_digest.py VSE_DW_COL 0 1).
// Plain fp16 tensors only (a pair input walks its hi rows, then its lo rows, into one chain: not streamable in that order), no gate, KW = KH
// in {3, 5}, 'same' padding, horizontal stride 1, vertical stride 1 or 2: the mobile recognisers' depthwise layers.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int KH, int SH, int OUTW>
__global__ __launch_bounds__(256) void dwconv_col_kernel(TView in, TView out, const float* __restrict__ w, const float* __restrict__ bias,
                                                         int rs, int nseg, int act, float act_a, float act_b, float post_a, float post_b,
                                                         const int* __restrict__ wl_out) {
    constexpr int KW = KH, P = KH / 2, R = (KH + SH - 1) / SH, U = R * SH, WIN = OUTW + KW - 1;
    const int C = in.c, cp = C >> 1;
    const bool affine = post_a != 1.f || post_b != 0.f;
    const int owq = (out.w + OUTW - 1) / OUTW;
    const long total = (long)out.n * nseg * owq * cp;
    const long i = xcd_block(blockIdx.x, gridDim.x) * 256L + threadIdx.x;
    if (i >= total) return;
    int g, q, sg;
    long t = vse_divmod(i, cp, g);                             // g: channel pair
    t = vse_divmod(t, owq, q);
    const long n = vse_divmod(t, nseg, sg);
    const int o0 = sg * rs, o1 = min(o0 + rs, out.h);          // this thread's output rows
    const int ow0 = q * OUTW, iw0 = ow0 - P;
    int coff[WIN];
    bool okc[WIN];
#pragma unroll
    for (int c = 0; c < WIN; ++c) {
        okc[c] = (unsigned)(iw0 + c) < (unsigned)in.w;
        coff[c] = min(max(iw0 + c, 0), in.w - 1) * in.ld;
    }
    const bool inside = __builtin_amdgcn_ballot_w64(!(okc[0] & okc[WIN - 1])) == 0;      // every lane's WIN columns lie in the image
    const half_t* img = reinterpret_cast<const half_t*>(in.ptr) + n * (long)in.h * in.w * in.ld + g * 2;
    float2v wk[KH * KW];
#pragma unroll
    for (int k = 0; k < KH * KW; ++k) wk[k] = *reinterpret_cast<const float2v*>(w + (long)k * C + g * 2);
    const float2v bias2 = *reinterpret_cast<const float2v*>(bias + g * 2);
    float2v acc[R][OUTW];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < OUTW; ++o) acc[r][o] = bias2;
    const int wl = wl_out != nullptr ? wl_out[n] : 0x7fffffff;
    half_t* const obase = reinterpret_cast<half_t*>(out.ptr) + g * 2;
    auto finish = [&](int oh, float2v (&a)[OUTW]) __attribute__((always_inline)) {      // activation, affine, fp16 store of one output row; slot back to bias
#pragma unroll
        for (int o = 0; o < OUTW; ++o) {
            float v2[2] = {a[o][0], a[o][1]};
            a[o] = bias2;
            vse_act_n<2>(v2, act, act_a, act_b);
            if (ow0 + o >= out.w) continue;
            half2v r2;
#pragma unroll
            for (int e = 0; e < 2; ++e) r2[e] = (half_t)(affine ? v2[e] * post_a + post_b : v2[e]);
            if (ow0 + o >= wl) r2 = half2v{0, 0};
            *reinterpret_cast<half2v*>(obase + ((n * out.h + oh) * (long)out.w + ow0 + o) * out.ld) = r2;
        }
    };
    auto load_row = [&](int ih, half2v (&xv)[WIN]) __attribute__((always_inline)) {
        const half_t* rowp = img + (long)min(max(ih, 0), in.h - 1) * in.w * in.ld;
#pragma unroll
        for (int c = 0; c < WIN; ++c) xv[c] = *reinterpret_cast<const half2v*>(rowp + coff[c]);
    };
    // local input row l = ih - o0 * SH runs from -U (a whole unrolled round in front of the segment: rows that reach no output row of the
    // segment are skipped by a wave-uniform test) to the last row the segment reads
    const int l_end = (o1 - 1 - o0) * SH - P + KH - 1;
    half2v nx[WIN];
    load_row(o0 * SH - P, nx);
#pragma unroll 1
    for (int lb = -U; lb <= l_end; lb += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int l = lb + j, ih = o0 * SH + l;
            if (l < -P || l > l_end) continue;                                  // (uniform)
            if (!inside) {                                                      // (uniform) columns outside the image read as zeros
#pragma unroll
                for (int c = 0; c < WIN; ++c)
                    if (!okc[c]) nx[c] = half2v{0, 0};
            }
            float2v xf[WIN];                                                    // this row in fp32 (exact conversions), then the next row's
#pragma unroll
            for (int c = 0; c < WIN; ++c) xf[c] = float2v{(float)nx[c][0], (float)nx[c][1]};      // gathers fly during this row's arithmetic
            load_row(ih + 1, nx);
            if (ih >= 0 && ih < in.h) {                                         // (uniform: rows outside the image contribute nothing)
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const int num = j + P - dy;                                 // = SH * (oh - o0 - lb / SH)
                    if (((num % SH) + SH) % SH != 0) continue;                  // (compile time after unrolling)
                    const int ohl = (num >= 0 ? num / SH : -((-num + SH - 1) / SH));        // floor division, compile time
                    const int slot = ((ohl % R) + R) % R;
                    const int oh = o0 + lb / SH + ohl;                          // (lb is a multiple of U = R * SH)
                    if (oh < o0 || oh >= o1) continue;                          // (uniform)
#pragma unroll
                    for (int dx = 0; dx < KW; ++dx)
#pragma unroll
                        for (int o = 0; o < OUTW; ++o) acc[slot][o] = __builtin_elementwise_fma(xf[o + dx], wk[dy * KW + dx], acc[slot][o]);
                }
            }
            {   // the output row whose LAST filter row this input row carries is complete
                const int num = j + P - (KH - 1);
                if (((num % SH) + SH) % SH == 0) {
                    const int ohl = (num >= 0 ? num / SH : -((-num + SH - 1) / SH));
                    const int slot = ((ohl % R) + R) % R;
                    const int oh = o0 + lb / SH + ohl;
                    if (oh >= o0 && oh < o1) finish(oh, acc[slot]);
                }
            }
        }
    }
}

#ifdef VSE_DEV_BUILD
// (development builds only: measured, bit-identical to the row kernel and 10-25 % SLOWER on every model — DESIGN §3.2 log)
// LDS-tile variant (round 4): the row kernel walks its kh filter rows as kh dependent global round trips per thread (loads of a row,
// wait, ~200 VALU instructions, next row) at 3-4 waves per SIMD — on the mobile models' 5 x 5 layers it sits at 20 % of its bytes' time
// (rec_fast: 14 depthwise layers = 39 % of the net).  Here a block of 256 threads owns TR output rows x 32 output columns x CGB 8-channel
// groups (TR * CGB = 32): it pulls the input patch ((TR - 1) * sh + kh rows, 31 * SW + KW columns) into LDS with ONE batch of independent
// 16-byte loads per thread (the SE gate applied on the way, once per element instead of once per use), and every thread then computes the
// same four outputs as in the row kernel, its windows read from LDS.  Same arithmetic per output (bias, taps row-major, hi + lo weights
// summed in fp32, padded taps skipped) -> bit-identical results.
//   LDS layout: [plane hi | lo][row][column][CGB vectors of 16 bytes (+ pad)]; the pad (16 bytes for CGB = 4, 32 for CGB = 8) makes the 16
//   lanes of a ds_read_b128 phase — CGB channel groups x 16 / CGB quads, quads 4 * SW columns apart — hit 16 distinct bank groups (SW = 1)
template <int KW, int SW, int GM>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(TView in, TView out, TView gate, int hilo, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int kh, int sh, int ph, int pw,
                                                          int act, float act_a, float act_b, float post_a, float post_b,
                                                          const int* __restrict__ wl_out, int trl, int tiles_w, int tiles_h, int cgblocks) {
    extern __shared__ __attribute__((aligned(16))) char tlds[];
    constexpr int OUTW = 4, TC = 32, WIN = (OUTW - 1) * SW + KW, IC = (TC - 1) * SW + KW;
    const int lo_off = ((hilo >> 1) & 0xfff) << 3, lo_in = ((hilo >> 13) & 0xfff) << 3;          // (see dwconv_kernel)
    hilo &= 1;
    const int TR = 1 << trl, cgbl = 5 - trl, CGB = 1 << cgbl;
    const int IR = (TR - 1) * sh + kh;
    const int colstride = CGB * 16 + (CGB == 4 ? 16 : CGB == 8 ? 32 : 0), rowstride = IC * colstride, plane = IR * rowstride;
    const int cg = in.c >> 3;
    unsigned b = xcd_block(blockIdx.x, gridDim.x);
    const int cgb = (int)(b % (unsigned)cgblocks);  b /= (unsigned)cgblocks;
    const int tx = (int)(b % (unsigned)tiles_w);  b /= (unsigned)tiles_w;
    const int ty = (int)(b % (unsigned)tiles_h);
    const long n = b / (unsigned)tiles_h;
    const int cg0 = cgb * CGB, oy0 = ty * TR, ox0 = tx * TC;
    const int iy0 = oy0 * sh - ph, ix0 = ox0 * SW - pw;
    const int tid = threadIdx.x;
    const bool dead_tile = wl_out != nullptr && ox0 >= wl_out[n];          // (ragged batch: right of the sample — zeros, nothing to read)
    if (!dead_tile) {
        const int nvec = IR * IC * CGB;
        for (int v = tid; v < nvec; v += 256) {
            const int g = v & (CGB - 1), rc = v >> cgbl, r = rc / IC, c = rc - r * IC;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = iy >= 0 && iy < in.h && ix >= 0 && ix < in.w && cg0 + g < cg;
            half8 x = half8{0, 0, 0, 0, 0, 0, 0, 0}, xl = x;
            if (ok) {
                const long pix = (n * in.h + iy) * in.w + ix;
                x = ld8(in, pix, (cg0 + g) * 8);
                if constexpr (GM != 0) x = dw_gate_t<GM>(x, ld8(gate, n, (cg0 + g) * 8));
                if (lo_in) xl = ld8(in, pix, (cg0 + g) * 8 + lo_in);
            }
            *reinterpret_cast<half8*>(tlds + r * rowstride + c * colstride + g * 16) = x;
            if (lo_in) *reinterpret_cast<half8*>(tlds + plane + r * rowstride + c * colstride + g * 16) = xl;
        }
    }
    __syncthreads();
    const int g = tid & (CGB - 1), q = (tid >> cgbl) & 7, r = tid >> (cgbl + 3);
    const int oh = oy0 + r, ow0 = ox0 + q * OUTW, gc = cg0 + g;
    if (oh >= out.h || ow0 >= out.w || gc >= cg) return;
    const int iw0 = ow0 * SW - pw;
    float acc[OUTW][8];
#pragma unroll
    for (int o = 0; o < OUTW; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = bias[gc * 8 + e];
    if (!dead_tile) {
        const char* base = tlds + (r * sh) * rowstride + (q * OUTW * SW) * colstride + g * 16;
        for (int dyp = 0; dyp < (lo_in ? 2 * kh : kh); ++dyp) {
            // (a pair input: every filter row is walked twice, over the hi and over the lo half of the same pixels)
            const int dy = dyp < kh ? dyp : dyp - kh;
            const int ih = oh * sh - ph + dy;
            if (ih < 0 || ih >= in.h) continue;
            const char* rowp = base + (dyp < kh ? 0 : plane) + dy * rowstride;
            half8 x[WIN];
#pragma unroll
            for (int c = 0; c < WIN; ++c) x[c] = *reinterpret_cast<const half8*>(rowp + c * colstride);
#pragma unroll
            for (int dx = 0; dx < KW; ++dx) {
                const float4v k0 = *reinterpret_cast<const float4v*>(w + (long)(dy * KW + dx) * in.c + gc * 8);
                const float4v k1 = *reinterpret_cast<const float4v*>(w + (long)(dy * KW + dx) * in.c + gc * 8 + 4);
#pragma unroll
                for (int o = 0; o < OUTW; ++o) {
                    const int iw = iw0 + o * SW + dx;
                    if (iw < 0 || iw >= in.w) continue;           // the reference kernel skips padded taps (no +0 rounding issue, same sums)
                    vse_fma_h8(acc[o], x[o * SW + dx], k0, k1);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < OUTW; ++o) {
        if (ow0 + o >= out.w) continue;
        half8 rr, rl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = vse_act(acc[o][e], act, act_a, act_b) * post_a + post_b;
            rr[e] = (half_t)v;
            rl[e] = (half_t)(v - (float)rr[e]);
        }
        if (wl_out != nullptr && ow0 + o >= wl_out[n]) rr = half8{0, 0, 0, 0, 0, 0, 0, 0};
        st8(out, (n * out.h + oh) * out.w + ow0 + o, gc * 8, rr);
        if (lo_off) st8(out, (n * out.h + oh) * out.w + ow0 + o, gc * 8 + lo_off, rl);
    }
}

#endif

// ------------------------------------------------------------------------------------------------ pooling
// wl_in / wl_out (ragged batch): the sample's own input / output width — the window is clipped to the sample, not to the
// batch tensor, and outputs right of the sample are zeros.
__global__ __launch_bounds__(256) void pool_kernel(TView in, TView out, int kh, int kw, int sh, int sw, int ph, int pw,
                                                   int is_max, int exclusive, const int* __restrict__ wl_in,
                                                   const int* __restrict__ wl_out) {
    const int cg = in.c >> 3;
    const long total = (long)out.n * out.h * out.w * cg;
    for (long i = xcd_block(blockIdx.x, gridDim.x) * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        long pix = vse_divmod(i, cg, g);
        int ow;
        long t = vse_divmod(pix, out.w, ow);
        int oh;
        const long n = vse_divmod(t, out.h, oh);
        const int inw = wl_in != nullptr ? wl_in[n] : in.w;
        if (wl_out != nullptr && ow >= wl_out[n]) {
            st8(out, pix, g * 8, half8{0, 0, 0, 0, 0, 0, 0, 0});
            continue;
        }
        if (is_max == 1) {
            // max pooling on the packed fp16 values themselves (v_pk_max_f16: 4 instructions per tap instead of 8 conversions + 8 fp32
            // maxima; a maximum is exact in any precision: the same bits as the fp32 form below)
            half8 m = {(half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f, (half_t)-65504.f};
            for (int dy = 0; dy < kh; ++dy) {
                const int ih = oh * sh - ph + dy;
                if (ih < 0 || ih >= in.h) continue;
                for (int dx = 0; dx < kw; ++dx) {
                    const int iw = ow * sw - pw + dx;
                    if (iw < 0 || iw >= inw) continue;
                    m = __builtin_elementwise_max(m, ld8(in, (n * in.h + ih) * in.w + iw, g * 8));
                }
            }
            st8(out, pix, g * 8, m);
            continue;
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = is_max ? -65504.f : 0.f;
        int cnt = 0;
        for (int dy = 0; dy < kh; ++dy) {
            const int ih = oh * sh - ph + dy;
            if (ih < 0 || ih >= in.h) continue;
            for (int dx = 0; dx < kw; ++dx) {
                const int iw = ow * sw - pw + dx;
                if (iw < 0 || iw >= inw) continue;
                const half8 x = ld8(in, (n * in.h + ih) * in.w + iw, g * 8);
                ++cnt;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = is_max ? fmaxf(acc[e], (float)x[e]) : acc[e] + (float)x[e];
            }
        }
        half8 o;
        if (is_max) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        } else {
            // exclusive: divide by the number of in-bounds taps; inclusive: by the window clipped to the padded input
            float div = (float)cnt;
            if (!exclusive) {
                const int h1 = min(oh * sh - ph + kh, in.h + ph), w1 = min(ow * sw - pw + kw, inw + pw);
                div = (float)((h1 - (oh * sh - ph)) * (w1 - (ow * sw - pw)));
            }
            const float inv = 1.f / div;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[e] * inv);
        }
        st8(out, pix, g * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------ global average
// grid (n, ceil(cg/8), splits); block 256 = 8 channel groups x 32 pixel lanes.  Partials (fp32) go to `part`
// [n][splits][c]; gap_finish reduces them.  With splits == 1 the partial pass writes the result directly.
__global__ __launch_bounds__(256) void gap_partial_kernel(TView in, float* __restrict__ part, int splits) {
    __shared__ float red[32][8][8];
    const int cgl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int g = blockIdx.y * 8 + cgl;
    const int n = blockIdx.x, s = blockIdx.z;
    const int cg = in.c >> 3;
    const long hw = (long)in.h * in.w;
    const long per = (hw + splits - 1) / splits;
    const long p0 = s * per, p1 = min(hw, p0 + per);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < cg) {
#pragma unroll 4
        for (long pp = p0 + pl; pp < p1; pp += 32) {
            const half8 x = ld8(in, n * hw + pp, g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl][cgl][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;   // 8 groups x 8 elems
        float sum = 0.f;
        for (int q = 0; q < 32; ++q) sum += red[q][c >> 3][c & 7];
        const int ch = blockIdx.y * 64 + c;
        if (ch < in.c) part[((long)n * splits + s) * in.c + ch] = sum;
    }
}
__global__ void gap_finish_kernel(const float* __restrict__ part, TView out, int splits, float inv_hw) {
    const long total = (long)out.n * out.c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / out.c;
        const int c = (int)(i % out.c);
        float s = 0.f;
        for (int q = 0; q < splits; ++q) s += part[(n * splits + q) * out.c + c];
        reinterpret_cast<half_t*>(out.ptr)[n * out.ld + c] = (half_t)(s * inv_hw);
    }
}

// Ragged batches: the same pool with a summation order that depends on the pixel's (row, column) only — never on the width of
// the batch tensor: grid (n, ceil(cg/8), h); lane pl of a block sums columns pl, pl + 32, ... of ROW blockIdx.z up to the
// sample's own width, the 32 lane sums are added in lane order, gap_rows_finish adds the rows in order and divides by
// h * width[n].  A sample therefore gets the same bits whatever batch it rides in (columns right of it would add zeros).
__global__ __launch_bounds__(256) void gap_rows_kernel(TView in, float* __restrict__ part, const int* __restrict__ wl_in) {
    __shared__ float red[32][8][8];
    const int cgl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int g = blockIdx.y * 8 + cgl;
    const int n = blockIdx.x, y = blockIdx.z;
    const int cg = in.c >> 3;
    const int wn = min(wl_in[n], in.w);
    const long row = ((long)n * in.h + y) * in.w;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < cg) {
#pragma unroll 4
        for (int x = pl; x < wn; x += 32) {
            const half8 v = ld8(in, row + x, g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl][cgl][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        float sum = 0.f;
        for (int q = 0; q < 32; ++q) sum += red[q][c >> 3][c & 7];
        const int ch = blockIdx.y * 64 + c;
        if (ch < in.c) part[((long)n * in.h + y) * in.c + ch] = sum;
    }
}
__global__ void gap_rows_finish_kernel(const float* __restrict__ part, TView out, int rows, const int* __restrict__ wl_in) {
    const long total = (long)out.n * out.c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / out.c;
        const int c = (int)(i % out.c);
        float s = 0.f;
        for (int q = 0; q < rows; ++q) s += part[(n * rows + q) * out.c + c];
        const float inv_hw = 1.f / ((float)rows * (float)wl_in[n]);
        reinterpret_cast<half_t*>(out.ptr)[n * out.ld + c] = (half_t)(s * inv_hw);
    }
}

// ------------------------------------------------------------------------------------------------ SE gate
__global__ __launch_bounds__(256) void scale_kernel(TView x, TView s, TView out, int add_x) {
    const int cg = x.c >> 3;
    const long hw = (long)x.h * x.w;
    const long total = (long)x.n * hw * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        const long pix = vse_divmod(i, cg, g);
        const long n = pix / hw;
        const half8 a = ld8(x, pix, g * 8);
        const half8 b = ld8(s, n, g * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)a[e] * (float)b[e];
            o[e] = (half_t)(add_x ? v + (float)a[e] : v);
        }
        st8(out, pix, g * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------ binary add/mul
__global__ __launch_bounds__(256) void binary_kernel(TView x, TView y, TView out, int is_mul, int shift, int act) {
    const int cg = x.c >> 3;
    const long total = (long)x.n * x.h * x.w * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        const long pix = vse_divmod(i, cg, g);
        long ypix = pix;
        if (shift) {
            int w;
            const long t = vse_divmod(pix, x.w, w);
            int h;
            const long n = vse_divmod(t, x.h, h);
            ypix = (n * y.h + (h >> shift)) * y.w + (w >> shift);
        }
        const half8 a = ld8(x, pix, g * 8);
        const half8 b = ld8(y, ypix, g * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = is_mul ? (float)a[e] * (float)b[e] : (float)a[e] + (float)b[e];
            o[e] = (half_t)vse_act(v, act, 0.f, 0.f);
        }
        st8(out, pix, g * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------ resize / copy
__global__ __launch_bounds__(256) void resize_kernel(TView in, TView out, int shift) {
    const int cg = out.c >> 3;
    const long total = (long)out.n * out.h * out.w * cg;
    for (long i = xcd_block(blockIdx.x, gridDim.x) * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        const long pix = vse_divmod(i, cg, g);
        int w;
        const long t = vse_divmod(pix, out.w, w);
        int h;
        const long n = vse_divmod(t, out.h, h);
        const long ipix = (n * in.h + (h >> shift)) * in.w + (w >> shift);
        st8(out, pix, g * 8, ld8(in, ipix, g * 8));
    }
}

// Gated form (round 4): out = up(x) * (1 + gate[n, c]) — an SE block with shortcut whose result only feeds a concat (the p-levels of
// the mobile detectors' RSE-FPN): the multiply, the shortcut add and the (up-sampled) copy into the concat slot are ONE pass, computed
// in fp32 and rounded once.  Two sources with ADJACENT slots share a launch (F_SRC2): a thread block then writes 2 x C contiguous
// channels per pixel instead of C (partial-line writes are what these copies cost: 0.7 TB/s measured on 48-byte pieces).
__global__ __launch_bounds__(256) void resize_gate_kernel(TView inA, TView gateA, int shiftA, TView inB, TView gateB, int shiftB, TView out, int plus1) {
    const int cgA = inA.c >> 3, cg = out.c >> 3;
    const long total = (long)out.n * out.h * out.w * cg;
    for (long i = xcd_block(blockIdx.x, gridDim.x) * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        const long pix = vse_divmod(i, cg, g);
        int w;
        const long t = vse_divmod(pix, out.w, w);
        int h;
        const long n = vse_divmod(t, out.h, h);
        const bool second = g >= cgA;
        const TView& in = second ? inB : inA;
        const TView& gate = second ? gateB : gateA;
        const int shift = second ? shiftB : shiftA, gi = second ? g - cgA : g;
        const long ipix = (n * in.h + (h >> shift)) * in.w + (w >> shift);
        const half8 x = ld8(in, ipix, gi * 8);
        half8 o = x;
        if (gate.ptr) {
            const half8 gv = ld8(gate, n, gi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] * ((plus1 ? 1.0f : 0.0f) + (float)gv[e]));
        }
        st8(out, pix, g * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------ unary
// out = act(x*pre_a+pre_b)*post_a+post_b.  Vector path when both sides are fp16 with 8-aligned spans; scalar
// path otherwise (e.g. the final 1-channel fp32 probability map).
__global__ __launch_bounds__(256) void unary_vec_kernel(TView in, TView out, int act, float act_a, float act_b,
                                                        float pre_a, float pre_b, float post_a, float post_b,
                                                        const int* __restrict__ wl_out) {
    const int cg = out.c >> 3;
    const long total = (long)out.n * out.h * out.w * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g;
        const long pix = vse_divmod(i, cg, g);
        const half8 a = ld8(in, pix, g * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (half_t)(vse_act((float)a[e] * pre_a + pre_b, act, act_a, act_b) * post_a + post_b);
        if (wl_out != nullptr && (int)(pix % out.w) >= wl_out[pix / ((long)out.h * out.w)]) o = half8{0, 0, 0, 0, 0, 0, 0, 0};
        st8(out, pix, g * 8, o);
    }
}
__global__ __launch_bounds__(256) void unary_scalar_kernel(TView in, TView out, int act, float act_a, float act_b,
                                                           float pre_a, float pre_b, float post_a, float post_b,
                                                           const int* __restrict__ wl_out) {
    const long total = (long)out.n * out.h * out.w * out.c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c;
        const long pix = vse_divmod(i, out.c, c);
        float x;
        if (in.esize == 2) x = (float)reinterpret_cast<const half_t*>(in.ptr)[pix * in.ld + c];
        else x = reinterpret_cast<const float*>(in.ptr)[pix * in.ld + c];
        float y = vse_act(x * pre_a + pre_b, act, act_a, act_b) * post_a + post_b;
        if (wl_out != nullptr && (int)(pix % out.w) >= wl_out[pix / ((long)out.h * out.w)]) y = 0.f;
        if (out.esize == 2) reinterpret_cast<half_t*>(out.ptr)[pix * out.ld + c] = (half_t)y;
        else reinterpret_cast<float*>(out.ptr)[pix * out.ld + c] = y;
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// 16 lanes per row (8 channels each; C <= 128), 4 rows per wave, two-pass mean/variance in fp32.
__global__ __launch_bounds__(256) void layernorm_kernel(TView in, TView out, const float* __restrict__ gb, float eps,
                                                        const int* __restrict__ wl_out) {
    const int C = in.c;
    const long rows = (long)in.n * in.h * in.w;
    const int sub = threadIdx.x & 15;
    const long row0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 4;
    const long rstride = ((long)gridDim.x * blockDim.x) >> 4;
    // the 16 lanes of a row group share r, so the width-16 shuffles only touch lanes with the same trip count
    for (long r = row0; r < rows; r += rstride) {
        const bool has = sub * 8 < C;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (has) {
            const half8 x = ld8(in, r, sub * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
        const float mean = s / (float)C;
        float q = 0.f;
        if (has) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q += (v[e] - mean) * (v[e] - mean);
        }
        for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o, 16);
        const float rstd = rsqrtf(q / (float)C + eps);
        if (has) {
            half8 o8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o8[e] = (half_t)((v[e] - mean) * rstd * gb[sub * 8 + e] + gb[C + sub * 8 + e]);
            if (wl_out != nullptr && (int)(r % in.w) >= wl_out[r / ((long)in.h * in.w)]) o8 = half8{0, 0, 0, 0, 0, 0, 0, 0};
            st8(out, r, sub * 8, o8);
        }
    }
}

// ------------------------------------------------------------------------------------------------ attention
// One block per (batch, head).  K and V of the head live in LDS as fp32 [T][hd]; thread t owns query row t and
// runs an online-softmax pass over all keys.  T <= a few hundred, hd <= 16: tiny FLOPs, latency-bound.
// tl (ragged batch): the sample's own sequence length — keys / queries at t >= tl[b] do not exist (their output rows are zeros).
__global__ __launch_bounds__(256) void attn_kernel(TView qkv, TView out, int heads, int hd, float scale, const int* __restrict__ tl) {
    extern __shared__ float kvs[];   // K [T][16] then V [T][16], head dim zero-padded to 16
    const int Tfull = qkv.w;
    const int b = blockIdx.x / heads, hix = blockIdx.x % heads;
    const int T = tl != nullptr ? min(tl[b], Tfull) : Tfull;
    const int C = heads * hd;
    float* Ks = kvs;
    float* Vs = kvs + (long)T * 16;
    const half_t* base = reinterpret_cast<const half_t*>(qkv.ptr) + (long)b * Tfull * qkv.ld;
    for (int i = threadIdx.x; i < T * 16; i += blockDim.x) {
        const int t = i >> 4, d = i & 15;
        const bool ok = d < hd;
        Ks[i] = ok ? (float)base[(long)t * qkv.ld + C + hix * hd + d] : 0.f;
        Vs[i] = ok ? (float)base[(long)t * qkv.ld + 2 * C + hix * hd + d] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float q[16], o[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            q[d] = d < hd ? (float)base[(long)t * qkv.ld + hix * hd + d] * scale : 0.f;
            o[d] = 0.f;
        }
        float mx = -1e30f, l = 0.f;
        for (int j = 0; j < T; ++j) {
            const float4v* kr = reinterpret_cast<const float4v*>(Ks + j * 16);
            const float4v* vr = reinterpret_cast<const float4v*>(Vs + j * 16);
            float s = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const float4v k4 = kr[d4];
#pragma unroll
                for (int e = 0; e < 4; ++e) s += q[d4 * 4 + e] * k4[e];
            }
            const float nm = fmaxf(mx, s);
            const float corr = __expf(mx - nm);
            const float pj = __expf(s - nm);
            l = l * corr + pj;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const float4v v4 = vr[d4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[d4 * 4 + e] = o[d4 * 4 + e] * corr + pj * v4[e];
            }
            mx = nm;
        }
        const float inv = 1.f / l;
        half_t* op = reinterpret_cast<half_t*>(out.ptr) + ((long)b * Tfull + t) * out.ld + hix * hd;
#pragma unroll
        for (int d = 0; d < 16; ++d)
            if (d < hd) op[d] = (half_t)(o[d] * inv);
    }
    for (int i = T * hd + threadIdx.x; i < Tfull * hd; i += blockDim.x)
        reinterpret_cast<half_t*>(out.ptr)[((long)b * Tfull + i / hd) * out.ld + hix * hd + i % hd] = (half_t)0.f;
}

// ------------------------------------------------------------------------------------------------ class softmax
// One block per (b,t) row of logits: max, sum-exp, arg-max (first index on ties, like numpy argmax);
// writes {argmax:int32, maxprob:fp32} and optionally the full fp32 probability row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(TView in, TView idxp, TView probs, int ncls, int want_probs) {
    __shared__ float smax[4];
    __shared__ int sidx[4];
    __shared__ float ssum[4];
    const long row = blockIdx.x;
    const T* x = reinterpret_cast<const T*>(in.ptr) + row * in.ld;
    float mx = -1e30f;
    int mi = 0x7fffffff;
    for (int c = threadIdx.x; c < ncls; c += blockDim.x) {
        const float v = (float)x[c];
        if (v > mx) { mx = v; mi = c; }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const float om = __shfl_xor(mx, o);
        const int oi = __shfl_xor(mi, o);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smax[wave] = mx; sidx[wave] = mi; }
    __syncthreads();
    mx = smax[0]; mi = sidx[0];
    for (int q = 1; q < 4; ++q)
        if (smax[q] > mx || (smax[q] == mx && sidx[q] < mi)) { mx = smax[q]; mi = sidx[q]; }
    float s = 0.f;
    for (int c = threadIdx.x; c < ncls; c += blockDim.x) s += expf((float)x[c] - mx);
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ssum[wave] = s;
    __syncthreads();
    s = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    const float inv = 1.f / s;
    if (threadIdx.x == 0) {
        int* ip = reinterpret_cast<int*>(idxp.ptr) + row * idxp.ld;
        ip[0] = mi;
        reinterpret_cast<float*>(ip)[1] = inv;   // max prob = exp(0)/sum
    }
    if (want_probs) {
        float* pr = reinterpret_cast<float*>(probs.ptr) + row * probs.ld;
        for (int c = threadIdx.x; c < ncls; c += blockDim.x) pr[c] = expf((float)x[c] - mx) * inv;
    }
}

// The same row softmax with the row held in REGISTERS (round 5): NV 16-byte vectors per thread, all loaded before the first use —
// ONE memory round trip per row where softmax_kernel's three scalar passes over 6625 classes make 3 x 26 dependent ones (77 us per
// 56-crop recogniser sequence, launch-to-launch).  Same arg-max rule (largest value, smallest index among equals); the sum is taken in
// this kernel's own (fixed) order.  Rows must be 16-byte aligned (the launcher checks); classes behind ncls count as -inf.
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_reg_kernel(TView in, TView idxp, TView probs, int ncls, int want_probs) {
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    __shared__ float smax[4];
    __shared__ int sidx[4];
    __shared__ float ssum[4];
    const long row = blockIdx.x;
    const T* x = reinterpret_cast<const T*>(in.ptr) + row * in.ld;
    const int nvec = (ncls + VW - 1) / VW;
    vec_t xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) xv[k] = *reinterpret_cast<const vec_t*>(x + (long)min(k * 256 + (int)threadIdx.x, nvec - 1) * VW);
    float v[NV][VW];
    float mx = -1e30f;
    int mi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int c = (k * 256 + (int)threadIdx.x) * VW + e;
            v[k][e] = c < ncls ? (float)xv[k][e] : -1e30f;
            if (v[k][e] > mx) { mx = v[k][e]; mi = c; }          // (c ascends inside a thread: the first of equal values is kept)
        }
    for (int o = 32; o >= 1; o >>= 1) {
        const float om = __shfl_xor(mx, o);
        const int oi = __shfl_xor(mi, o);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smax[wave] = mx; sidx[wave] = mi; }
    __syncthreads();
    mx = smax[0]; mi = sidx[0];
    for (int q = 1; q < 4; ++q)
        if (smax[q] > mx || (smax[q] == mx && sidx[q] < mi)) { mx = smax[q]; mi = sidx[q]; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int c = (k * 256 + (int)threadIdx.x) * VW + e;
            v[k][e] = c < ncls ? expf(v[k][e] - mx) : 0.f;
            s += v[k][e];
        }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ssum[wave] = s;
    __syncthreads();
    s = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    const float inv = 1.f / s;
    if (threadIdx.x == 0) {
        int* ip = reinterpret_cast<int*>(idxp.ptr) + row * idxp.ld;
        ip[0] = mi;
        reinterpret_cast<float*>(ip)[1] = inv;   // max prob = exp(0)/sum
    }
    if (want_probs) {
        float* pr = reinterpret_cast<float*>(probs.ptr) + row * probs.ld;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int c = (k * 256 + (int)threadIdx.x) * VW + e;
                if (c < ncls) pr[c] = v[k][e] * inv;
            }
    }
}

// ------------------------------------------------------------------------------------------------ LSTM
// One block per batch row; gates fp32 [B,1,T,4H] hold x.W_ih^T + b_ih + b_hh for every step (one MFMA GEMM up
// front); this kernel adds h.W_hh^T and runs the cell.  W_hh^T is fp16 [H][4H] read through L2 every step.
// H = 256 -> 1024 gate columns; thread j (of 256) owns hidden unit j and computes its 4 gates.
__global__ __launch_bounds__(256) void lstm_kernel(TView gates, TView out, const half_t* __restrict__ whh, int H, int rev,
                                                   const int* __restrict__ tl) {
    extern __shared__ float hs[];   // h [H]
    const int b = blockIdx.x;
    const int Tfull = gates.w;
    const int T = tl != nullptr ? min(tl[b], Tfull) : Tfull;     // ragged batch: the sample's own length (the reverse pass starts at ITS end)
    const int j = threadIdx.x;
    float c = 0.f;
    if (j < H) hs[j] = 0.f;
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = rev ? T - 1 - step : step;
        const float* g = reinterpret_cast<const float*>(gates.ptr) + ((long)b * Tfull + t) * gates.ld;
        float zi = 0.f, zf = 0.f, zg = 0.f, zo = 0.f;
        if (j < H) {
            zi = g[j]; zf = g[H + j]; zg = g[2 * H + j]; zo = g[3 * H + j];
            for (int k = 0; k < H; ++k) {
                const float hk = hs[k];
                const half_t* wr = whh + (long)k * 4 * H;
                zi += hk * (float)wr[j];
                zf += hk * (float)wr[H + j];
                zg += hk * (float)wr[2 * H + j];
                zo += hk * (float)wr[3 * H + j];
            }
        }
        __syncthreads();
        if (j < H) {
            const float i_ = 1.f / (1.f + __expf(-zi)), f_ = 1.f / (1.f + __expf(-zf)), o_ = 1.f / (1.f + __expf(-zo));
            c = f_ * c + i_ * tanhf(zg);
            const float h = o_ * tanhf(c);
            hs[j] = h;
            reinterpret_cast<half_t*>(out.ptr)[((long)b * Tfull + t) * out.ld + j] = (half_t)h;
        }
        __syncthreads();
    }
    if (j < H)
        for (int t = T; t < Tfull; ++t) reinterpret_cast<half_t*>(out.ptr)[((long)b * Tfull + t) * out.ld + j] = (half_t)0.f;
}

// ------------------------------------------------------------------------------------------------ dispatch
// OP_WSCALE: per-image 1x1 conv weights = the tiled weight blob [Kp/kt][Np][kt] times the image's SE gate over k (rounded to fp16
// once, like the separate gate multiply rounds its products).  One thread per 8 consecutive k of one cout row.
__global__ __launch_bounds__(256) void wscale_kernel(const half_t* __restrict__ w, TView gate, half_t* __restrict__ out, int n_img,
                                                     int Kp, int Np, int kt) {
    const long per = (long)Kp * Np / 8;
    const long total = per * n_img;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / per);
        const long e = (i - (long)n * per) * 8;                  // element index inside the tiled blob
        const int k = (int)(e / ((long)Np * kt)) * kt + (int)(e % kt);
        const half8 wv = *reinterpret_cast<const half8*>(w + e);
        const half_t* g = reinterpret_cast<const half_t*>(gate.ptr) + (long)n * gate.ld;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)wv[j] * (k + j < gate.c ? (float)g[k + j] : 0.f));
        *reinterpret_cast<half8*>(out + (long)n * Kp * Np + e) = o;
    }
}

#ifdef VSE_DEV_BUILD
// dwconv_tile_kernel: tile rows by the map height (least dead rows, then the taller tile), LDS by the patch; VSE_E_UNSUPPORTED -> row kernel
template <int KW, int SW, int GM>
static int launch_dw_tile(const TView& in0, const TView& out, const TView& gate, int hilo, const float* wk, const float* bk, const int* p,
                          const float* f, const int* wl_out, hipStream_t st) {
    const int kh = p[P_KH], sh = p[P_SH];
    int trl = 3;
    long best = -1;
    for (int t = 3; t >= 1; --t) {
        const long rows = (long)((out.h + (1 << t) - 1) >> t) << t;
        if (best < 0 || rows < best) { best = rows; trl = t; }
    }
    const int TR = 1 << trl, CGB = 32 >> trl, IC = 31 * SW + KW, IR = (TR - 1) * sh + kh;
    const int colstride = CGB * 16 + (CGB == 4 ? 16 : CGB == 8 ? 32 : 0);
    const size_t lds = (size_t)(p[P_LO_RES] ? 2 : 1) * IR * IC * colstride;
    if (lds > 150 * 1024) return VSE_E_UNSUPPORTED;
    static VseDevOnce attr_once;          // (per device, thread-safe: common.h)
    if (!vse_dev_once(attr_once, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_tile_kernel<KW, SW, GM>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        }))
        return VSE_E_HIP;
    const int cg = in0.c >> 3, cgblocks = (cg + CGB - 1) / CGB, tiles_w = (out.w + 31) / 32, tiles_h = (out.h + TR - 1) / TR;
    const unsigned long long blocks = (unsigned long long)out.n * tiles_h * tiles_w * cgblocks;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_UNSUPPORTED;
    hipLaunchKernelGGL((dwconv_tile_kernel<KW, SW, GM>), dim3((unsigned)blocks), dim3(256), lds, st, in0, out, gate, hilo, wk, bk, kh, sh, p[P_PH], p[P_PW],
                       p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A], f[FS_POST_B], wl_out, trl, tiles_w, tiles_h, cgblocks);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
#endif

int launch_simple_op(const vse_op& op, const TView& in0, const TView& in1, const TView& in2, const TView& out,
                     const TView& out2, const char* wbase, const int* wl_in, const int* wl_out, hipStream_t st) {
    const int* p = op.p;
    const float* f = op.f;
    switch (op.kind) {
        case OP_DWCONV: {
            if ((in0.c & 7) || in0.esize != 2 || out.c != in0.c) return VSE_E_INVAL;
            const long items = (long)out.n * out.h * out.w * (in0.c >> 3);
            const float* wk = reinterpret_cast<const float*>(wbase + op.w_off);       // fp32 [kh * kw][C] (hi + lo summed by the compiler)
            const float* bk = reinterpret_cast<const float*>(wbase + op.b_off);
            const int kw = p[P_KW], sw = p[P_SW];
            TView gate = in1;
            const int gmode = (op.flags & F_RES) ? 2 : 1;
            if (p[P_LO_OUT] && ((p[P_LO_OUT] & 7) || out.ld < p[P_LO_OUT] + out.c || wl_out)) return VSE_E_INVAL;
            if (p[P_LO_RES] && ((p[P_LO_RES] & 7) || in0.ld < p[P_LO_RES] + in0.c || (op.flags & F_GATE))) return VSE_E_INVAL;
            // (the pair offsets of the output / of the input ride in the upper bits of `hilo`, in units of 8 channels)
            const int hilo = ((op.flags & F_HILO) ? 1 : 0) | ((p[P_LO_OUT] >> 3) << 1) | ((p[P_LO_RES] >> 3) << 13);
            if (!(op.flags & F_GATE)) gate.ptr = nullptr;
            else if (!in1.ptr || in1.c != in0.c || in1.n != in0.n || in1.esize != 2) return VSE_E_INVAL;
            // column-walk form (dwconv_col_kernel): plain fp16 tensors, no gate, square 3 x 3 / 5 x 5 'same' filters, horizontal stride 1
            static const bool dw_col = [] { const char* e = vse_dev_getenv("VSE_DW_COL"); return !(e && e[0] == '0'); }();
            if (dw_col && !gate.ptr && !p[P_LO_OUT] && !p[P_LO_RES] && kw == p[P_KH] && (kw == 3 || kw == 5) && sw == 1 && (p[P_SH] == 1 || p[P_SH] == 2)
                && p[P_PH] == kw / 2 && p[P_PW] == kw / 2 && out.w == in0.w && out.h == (in0.h + 2 * (kw / 2) - kw) / p[P_SH] + 1
                && in0.ld * (long)in0.w * in0.h < 0x7fffffffl) {
#ifndef VSE_DWCOL_OUTW
#define VSE_DWCOL_OUTW 4
#endif
                const int outw = VSE_DWCOL_OUTW, cpn = in0.c >> 1, owq = (out.w + outw - 1) / outw;
                const long cols = (long)out.n * owq * cpn;                     // one thread per (column strip, channel pair, row segment)
                int nseg = (int)((262144 + cols - 1) / cols);                   // >= ~256 k threads where the map allows it, >= 4 rows per segment
                if (nseg > (out.h + 3) / 4) nseg = (out.h + 3) / 4;
                if (nseg < 1) nseg = 1;
                const int rs = (out.h + nseg - 1) / nseg;
                nseg = (out.h + rs - 1) / rs;
                const long total = cols * nseg;
                const unsigned long long blocks = (unsigned long long)((total + 255) / 256);
                if (blocks > 0 && blocks <= 0x7fffffffull) {
#define DW_COL(KH_, SH_, OW_) hipLaunchKernelGGL((dwconv_col_kernel<KH_, SH_, OW_>), dim3((unsigned)blocks), dim3(256), 0, st, in0, out, wk, bk, rs, nseg, \
                                                 p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A], f[FS_POST_B], wl_out)
                    if (kw == 5 && p[P_SH] == 1) DW_COL(5, 1, VSE_DWCOL_OUTW);
                    else if (kw == 5) DW_COL(5, 2, VSE_DWCOL_OUTW);
                    else if (p[P_SH] == 1) DW_COL(3, 1, VSE_DWCOL_OUTW);
                    else DW_COL(3, 2, VSE_DWCOL_OUTW);
#undef DW_COL
                    break;
                }
            }
            if ((kw == 3 || kw == 5) && (sw == 1 || sw == 2)) {
#ifdef VSE_DEV_BUILD
                // VSE_DW_TILE: 0 = row kernel only, 1 = LDS-tile kernel for 5 x 5 filters, 2 = for 3 x 3 filters too
                static const int dw_tile = getenv("VSE_DW_TILE") ? atoi(getenv("VSE_DW_TILE")) : 0;
                if (dw_tile >= (kw == 5 ? 1 : 2)) {
                    int rc = VSE_E_UNSUPPORTED;
#define DW_TILE(KW_, SW_) do { \
                        if (!gate.ptr) rc = launch_dw_tile<KW_, SW_, 0>(in0, out, gate, hilo, wk, bk, p, f, wl_out, st); \
                        else if (gmode == 1) rc = launch_dw_tile<KW_, SW_, 1>(in0, out, gate, hilo, wk, bk, p, f, wl_out, st); \
                        else rc = launch_dw_tile<KW_, SW_, 2>(in0, out, gate, hilo, wk, bk, p, f, wl_out, st); } while (0)
                    if (kw == 3 && sw == 1) DW_TILE(3, 1);
                    else if (kw == 3) DW_TILE(3, 2);
                    else if (sw == 1) DW_TILE(5, 1);
                    else DW_TILE(5, 2);
#undef DW_TILE
                    if (rc == VSE_OK) break;
                    if (rc != VSE_E_UNSUPPORTED) return rc;
                }
#endif
                const long items4 = (long)out.n * out.h * ((out.w + 3) / 4) * (in0.c >> 3);
                const dim3 g4(grid_for(items4, 256)), b4(256);
#define DW_ROW(KW_, SW_) do { \
                    if (!gate.ptr) hipLaunchKernelGGL((dwconv_row_kernel<KW_, SW_, 0>), g4, b4, 0, st, in0, out, gate, hilo, wk, bk, p[P_KH], p[P_SH], \
                                                      p[P_PH], p[P_PW], p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A], f[FS_POST_B], wl_out); \
                    else if (gmode == 1) hipLaunchKernelGGL((dwconv_row_kernel<KW_, SW_, 1>), g4, b4, 0, st, in0, out, gate, hilo, wk, bk, p[P_KH], \
                                                            p[P_SH], p[P_PH], p[P_PW], p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A], f[FS_POST_B], wl_out); \
                    else hipLaunchKernelGGL((dwconv_row_kernel<KW_, SW_, 2>), g4, b4, 0, st, in0, out, gate, hilo, wk, bk, p[P_KH], p[P_SH], \
                                            p[P_PH], p[P_PW], p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A], f[FS_POST_B], wl_out); } while (0)
                if (kw == 3 && sw == 1) DW_ROW(3, 1);
                else if (kw == 3) DW_ROW(3, 2);
                else if (sw == 1) DW_ROW(5, 1);
                else DW_ROW(5, 2);
#undef DW_ROW
                break;
            }
            hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, out, gate, gmode, hilo, wk, bk, p[P_KH], p[P_KW],
                               p[P_SH], p[P_SW], p[P_PH], p[P_PW], p[P_ACT], f[FS_ACT_A], f[FS_ACT_B], f[FS_POST_A],
                               f[FS_POST_B], wl_out);
            break;
        }
        case OP_POOL: {
            if ((in0.c & 7) || out.c != in0.c) return VSE_E_INVAL;
            const long items = (long)out.n * out.h * out.w * (in0.c >> 3);
            // (is_max = 2: the fp32 form of the max, for A/B runs: VSE_POOL_PK=0)
            static const int pk_off = vse_dev_getenv("VSE_POOL_PK") && atoi(vse_dev_getenv("VSE_POOL_PK")) == 0;
            hipLaunchKernelGGL(pool_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, out, p[P_KH], p[P_KW],
                               p[P_SH], p[P_SW], p[P_PH], p[P_PW], p[P_POOL_MAX] ? (pk_off ? 2 : 1) : 0, p[P_POOL_EXCL], wl_in, wl_out);
            break;
        }
        case OP_GAP: {
            if ((in0.c & 7) || out.c != in0.c || in2.ptr == nullptr) return VSE_E_INVAL;
            const int splits = in2.h;   // scratch view [n, splits, 1, c] fp32
            if (wl_in != nullptr) {      // ragged batch: row-structured sums (one split per row), per-sample divisor
                if (splits != in0.h) return VSE_E_INVAL;
                hipLaunchKernelGGL(gap_rows_kernel, dim3(in0.n, (in0.c + 63) / 64, in0.h), dim3(256), 0, st, in0,
                                   reinterpret_cast<float*>(in2.ptr), wl_in);
                hipLaunchKernelGGL(gap_rows_finish_kernel, dim3(grid_for((long)out.n * out.c, 256)), dim3(256), 0, st,
                                   reinterpret_cast<const float*>(in2.ptr), out, in0.h, wl_in);
                break;
            }
            dim3 grid(in0.n, (in0.c + 63) / 64, splits);
            hipLaunchKernelGGL(gap_partial_kernel, grid, dim3(256), 0, st, in0, reinterpret_cast<float*>(in2.ptr), splits);
            const long items = (long)out.n * out.c;
            hipLaunchKernelGGL(gap_finish_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st,
                               reinterpret_cast<const float*>(in2.ptr), out, splits, 1.f / ((float)in0.h * in0.w));
            break;
        }
        case OP_SCALE: {
            const long items = (long)in0.n * in0.h * in0.w * (in0.c >> 3);
            hipLaunchKernelGGL(scale_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, in1, out,
                               (op.flags & F_RES) ? 1 : 0);
            break;
        }
        case OP_BINARY: {
            const long items = (long)in0.n * in0.h * in0.w * (in0.c >> 3);
            hipLaunchKernelGGL(binary_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, in1, out, p[0], p[1], p[2]);
            break;
        }
        case OP_RESIZE: {
            const long items = (long)out.n * out.h * out.w * (out.c >> 3);
            if (op.flags & (F_GATE | F_SRC2)) {
                // in0 = source A, in1 = its gate [N,1,1,C] (or none), p[0] = shift; F_SRC2: in2 = source B, out2 = ITS GATE (an input), p[1]
                TView gA = in1, inB = in2, gB = out2;
                if (!(op.flags & F_GATE)) gA.ptr = nullptr, gB.ptr = nullptr;
                if (!(op.flags & F_SRC2)) inB = in0, gB = gA;
                const int ca = in0.c, cb = (op.flags & F_SRC2) ? in2.c : 0;
                if ((ca & 7) || (cb & 7) || out.c != ca + cb || in0.esize != 2 || out.esize != 2) return VSE_E_INVAL;
                if ((op.flags & F_GATE) && (!in1.ptr || in1.c < ca || ((op.flags & F_SRC2) && (!out2.ptr || out2.c < cb)))) return VSE_E_INVAL;
                hipLaunchKernelGGL(resize_gate_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, gA, p[0], inB, gB, p[1], out, (op.flags & F_RES) ? 1 : 0);
                break;
            }
            if (in0.c < out.c) return VSE_E_INVAL;
            hipLaunchKernelGGL(resize_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, out, p[0]);
            break;
        }
        case OP_UNARY: {
            const bool vec = in0.esize == 2 && out.esize == 2 && !(out.c & 7) && !(in0.ld & 7) && !(out.ld & 7);
            if (vec) {
                const long items = (long)out.n * out.h * out.w * (out.c >> 3);
                hipLaunchKernelGGL(unary_vec_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, out, p[0],
                                   f[FS_ACT_A], f[FS_ACT_B], f[FS_PRE_A], f[FS_PRE_B], f[FS_POST_A], f[FS_POST_B], wl_out);
            } else {
                const long items = (long)out.n * out.h * out.w * out.c;
                hipLaunchKernelGGL(unary_scalar_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st, in0, out, p[0],
                                   f[FS_ACT_A], f[FS_ACT_B], f[FS_PRE_A], f[FS_PRE_B], f[FS_POST_A], f[FS_POST_B], wl_out);
            }
            break;
        }
        case OP_LAYERNORM: {
            if (in0.c > 128 || (in0.c & 7)) return VSE_E_UNSUPPORTED;
            const long rows = (long)in0.n * in0.h * in0.w;
            hipLaunchKernelGGL(layernorm_kernel, dim3(grid_for(rows * 16, 256)), dim3(256), 0, st, in0, out,
                               reinterpret_cast<const float*>(wbase + op.w_off), f[FS_EPS], wl_out);
            break;
        }
        case OP_ATTN: {
            const int heads = p[0], hd = p[1];
            if (hd > 16) return VSE_E_UNSUPPORTED;
            const size_t lds = (size_t)2 * in0.w * 16 * sizeof(float);
            if (lds > 160 * 1024) return VSE_E_UNSUPPORTED;
            hipLaunchKernelGGL(attn_kernel, dim3(in0.n * heads), dim3(256), lds, st, in0, out, heads, hd, f[FS_SCALE], wl_in);
            break;
        }
        case OP_SOFTMAX: {
            const long rows = (long)in0.n * in0.h * in0.w;
            const int wp = out2.ptr != nullptr ? 1 : 0;
            // the row in registers (one memory round trip) when rows are 16-byte aligned and fit 8 vectors per thread
            const int vw = 16 / in0.esize, nvt = ((p[0] + vw - 1) / vw + 255) / 256;
            if ((reinterpret_cast<uintptr_t>(in0.ptr) & 15) == 0 && ((long)in0.ld * in0.esize) % 16 == 0 && nvt <= 8 && p[0] > 0
                && (long)((p[0] + vw - 1) / vw) * vw <= in0.ld) {
                const dim3 g((unsigned)rows), b(256);
#define VSE_SM_LAUNCH(T, NV) hipLaunchKernelGGL((softmax_reg_kernel<T, NV>), g, b, 0, st, in0, out, out2, p[0], wp)
                if (in0.esize == 4) {
                    if (nvt <= 1) VSE_SM_LAUNCH(float, 1); else if (nvt <= 2) VSE_SM_LAUNCH(float, 2); else if (nvt <= 4) VSE_SM_LAUNCH(float, 4); else VSE_SM_LAUNCH(float, 8);
                } else {
                    if (nvt <= 1) VSE_SM_LAUNCH(half_t, 1); else if (nvt <= 2) VSE_SM_LAUNCH(half_t, 2); else if (nvt <= 4) VSE_SM_LAUNCH(half_t, 4); else VSE_SM_LAUNCH(half_t, 8);
                }
#undef VSE_SM_LAUNCH
                break;
            }
            if (in0.esize == 4)
                hipLaunchKernelGGL(softmax_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, in0, out, out2, p[0],
                                   out2.ptr != nullptr ? 1 : 0);
            else
                hipLaunchKernelGGL(softmax_kernel<half_t>, dim3((unsigned)rows), dim3(256), 0, st, in0, out, out2, p[0],
                                   out2.ptr != nullptr ? 1 : 0);
            break;
        }
        case OP_LSTM: {
            const int H = p[0];
            if (op.flags & F_LSTM_MFMA) {
                if (H != 256) return VSE_E_UNSUPPORTED;
                return launch_lstm_mfma(in0, in1, out, reinterpret_cast<const half_t*>(wbase + op.w_off), p[1] == 1 ? 1 : 0,
                                        p[1] == 2 ? 2 : 1, p[2] ? p[2] : 8, wl_in, st);
            }
            if (H > 256 || p[1] > 1) return VSE_E_UNSUPPORTED;
            hipLaunchKernelGGL(lstm_kernel, dim3(in0.n), dim3(256), H * sizeof(float), st, in0, out,
                               reinterpret_cast<const half_t*>(wbase + op.w_off), H, p[1], wl_in);
            break;
        }
        case OP_WSCALE: {
            const int Kp = p[0], Np = p[1], kt = p[2];
            if (Kp <= 0 || Np <= 0 || (kt != 32 && kt != 64) || Kp % kt || in0.esize != 2 || out.esize != 2) return VSE_E_INVAL;
            const long items = (long)in0.n * Kp * Np / 8;
            hipLaunchKernelGGL(wscale_kernel, dim3(grid_for(items, 256)), dim3(256), 0, st,
                               reinterpret_cast<const half_t*>(wbase + op.w_off), in0, reinterpret_cast<half_t*>(out.ptr), in0.n, Kp, Np, kt);
            break;
        }
        default:
            return VSE_E_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
