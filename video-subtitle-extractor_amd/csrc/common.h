// Shared device helpers for the vse HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vse_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_SWISH = 3, ACT_SIGMOID = 4, ACT_HSIGMOID = 5 };
enum { OP_CONV = 1, OP_DWCONV, OP_POOL, OP_GAP, OP_SCALE, OP_BINARY, OP_RESIZE, OP_UNARY, OP_LAYERNORM, OP_ATTN,
       OP_SOFTMAX, OP_LSTM, OP_WSCALE, OP_CHAIN };   // OP_CHAIN: 1x1 / depthwise conv chain with LDS-resident intermediates (chain.hip)
enum { F_LSTM_MFMA = 16384, F_ONECH = 32768, F_U8SRC = 65536, F_OGATE = 131072, F_DWPRE = 262144, F_HLSUM = 524288, F_TAIL2 = 1048576 };   // F_HLSUM: hi | lo weight rows in one 64-row stage, accumulator tiles added (see ir.py)   // F_DWPRE: depthwise conv fused in front of a 1x1 conv (conv_dwpw.hip)   // F_OGATE: in2 = per-image output gate (see ir.py)   // F_U8SRC: the stem conv resizes the uint8 BGR frames itself (see ir.py)   // F_ONECH: pixel-shuffle conv to one channel stores the fp32 map itself (see ir.py)
enum { F_LSTM_MFMA_ = 0 };   // OP_LSTM: W_hh^T in MFMA fragment order, H = 256 (lstm.hip); p[P_REVERSE] = 2: both directions, in1 = reverse gates
enum { F_RES = 1, F_PIXSHUF = 2, F_OUT_F32 = 4, F_PATCH = 8, F_DOT1 = 16, F_SRC2 = 32, F_UP2HEAD = 64, F_WK32 = 128, F_GATE = 256, F_STEM = 512, F_HILO = 1024, F_COL = 2048, F_PW = 4096, F_IMGW = 8192 };
// p[] slots (keep in sync with ir.py)
enum { P_KH = 0, P_KW, P_SH, P_SW, P_PH, P_PW, P_ACT, P_ACT2, P_COUT, P_KTOT, P_INSHIFT, P_RESSHIFT, P_CINP, P_DOTACT, P_IN2SHIFT, P_LO_OUT, P_LO_RES, P_LO_IN };
enum { P_POOL_MAX = 6, P_POOL_CEIL = 7, P_POOL_EXCL = 8 };
// ragged plans: 1 + width level of in0 / of the output (0 = no per-sample width); the run supplies widths[level][n]
enum { P_WLIN = 20, P_WLOUT = 21 };
enum { FS_ACT_A = 0, FS_ACT_B, FS_POST_A, FS_POST_B, FS_PRE_A, FS_PRE_B, FS_EPS, FS_SCALE };

__device__ __forceinline__ float vse_act(float x, int code, float a, float b) {
    switch (code) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case ACT_SWISH: return x / (1.f + __expf(-x));
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case ACT_HSIGMOID: return fminf(fmaxf(x * a + b, 0.f), 1.f);
        default: return x;
    }
}

// One activation code applied to N values: ONE uniform switch per call (per-element switches blow the epilogue up to
// thousands of scalar branches).
template <int N> __device__ __forceinline__ void vse_act_n(float (&v)[N], int code, float a, float b) {
    switch (code) {
        case ACT_RELU:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.f);
            break;
        case ACT_HSWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * fminf(fmaxf(v[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
            break;
        case ACT_SWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] / (1.f + __expf(-v[e]));
            break;
        case ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
            break;
        case ACT_HSIGMOID:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = fminf(fmaxf(v[e] * a + b, 0.f), 1.f);
            break;
        default: break;
    }
}

// acc[e] = fma((float)x[e], w[e], acc[e]) for the 8 fp16 values of x as EIGHT v_fma_mix_f32: the conversion rides in the multiply-add
// (op_sel_hi marks src0 as fp16, op_sel picks the half of the packed register) — hipcc emits v_cvt_f32_f16 + v_fma_f32 for the same
// source, and the depthwise kernels are VALU-bound (counters, round 4: the vector pipe 65-85 % busy at 2.4 TB/s of their bytes).
// Same value as fmaf((float)x[e], w[e], acc[e]): the fp16 -> fp32 conversion is exact, the multiply-add fused in fp32.
typedef unsigned vse_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float vse_fma_mix_lo(unsigned xp, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(xp), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float vse_fma_mix_hi(unsigned xp, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(xp), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ void vse_fma_h8(float (&acc)[8], const half8& x, const float4v& w0, const float4v& w1) {
    const vse_u32x4 u = __builtin_bit_cast(vse_u32x4, x);
    acc[0] = vse_fma_mix_lo(u[0], w0[0], acc[0]);
    acc[1] = vse_fma_mix_hi(u[0], w0[1], acc[1]);
    acc[2] = vse_fma_mix_lo(u[1], w0[2], acc[2]);
    acc[3] = vse_fma_mix_hi(u[1], w0[3], acc[3]);
    acc[4] = vse_fma_mix_lo(u[2], w1[0], acc[4]);
    acc[5] = vse_fma_mix_hi(u[2], w1[1], acc[5]);
    acc[6] = vse_fma_mix_lo(u[3], w1[2], acc[6]);
    acc[7] = vse_fma_mix_hi(u[3], w1[3], acc[7]);
}

// An EXPERIMENT switch (kernel selection for A/B runs, ablations): read from the environment only in a development build
// (-DVSE_DEV_BUILD); the product library has none of them.  The product's own switches are listed in INTEGRATION.md.
#include <stdlib.h>
static inline const char* vse_dev_getenv(const char* name) {
#ifdef VSE_DEV_BUILD
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Per-device launch state (a raised dynamic-LDS limit, the CU count): hipFuncSetAttribute acts on the CURRENT device and a process may
// hold contexts on several devices and launch from several host threads, so "done once" is kept per device id, under a mutex.
#include <mutex>
struct VseDevOnce {
    std::mutex m;
    unsigned long long done = 0;          // bit d: device d is set up (<= 64 devices per process)
};
template <class F>
static inline bool vse_dev_once(VseDevOnce& s, F&& setup) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    std::lock_guard<std::mutex> g(s.m);
    if ((s.done >> dev) & 1ull) return true;
    if (!setup()) return false;
    s.done |= 1ull << dev;
    return true;
}
static inline int vse_cu_count() {         // compute units of the current device (0 on error)
    static std::mutex m;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 0;
    std::lock_guard<std::mutex> g(m);
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

// Staging a table global -> LDS / registers with ALL of a thread's loads in flight before the first use (round 5).  The obvious
//     for (v = tid; v < n; v += 256) { if (ok(v)) x = src[v]; dst[v] = x; }
// compiles to load -> s_waitcnt vmcnt(0) -> ds_write per iteration — and a load under a condition to a branch around it besides: one
// memory round trip per 256 vectors, 3 .. 10 dependent round trips (3 - 8 us) in front of the first MFMA of a block that then computes
// for 2 us (conv_pw / conv_dwpw / conv_stem / chain_pw2 before this helper).  ld(v) must be UNCONDITIONAL (clamp the address, select the
// value in st); CH loads per thread and round.
template <int CH, int NT = 256, typename LD, typename ST>
__device__ __forceinline__ void stage_batched(int n, int tid, LD&& ld, ST&& st) {
    for (int b = 0; b < n; b += CH * NT) {
        decltype(ld(0)) t[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) t[k] = ld(min(b + k * NT + tid, n - 1));
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int v = b + k * NT + tid;
            if (v < n) st(v, t[k]);
        }
    }
}

// XCD-aware block order for STREAMING kernels whose neighbouring outputs share input rows (depthwise / pooling windows, up-sampling
// copies): the dispatcher places block b on XCD b % 8 and every XCD has a private L2, so with the plain order the blocks of
// vertically adjacent rows land on different XCDs and every input row is fetched from memory once per XCD that needs it (counters,
// round 4: 2.2 - 3.5 x the algorithmic bytes on the mobile detector's depthwise layers).  This bijection gives every XCD one CONTIGUOUS
// run of logical blocks (the formula of conv_mfma.hip), so the window overlap is served by the L2 that already holds the row.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblk) {
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// q = a / b and r = a % b for a flattened element index a >= 0 and an extent b > 0.  A 64-bit division by a runtime value expands to ~100
// instructions and the element-wise / window kernels do three or four per thread; indices that fit 32 bits (every tensor the models produce;
// one comparison, uniform in practice) take the 32-bit unsigned division instead.
__device__ __forceinline__ long vse_divmod(long a, int b, int& r) {
    if ((unsigned long)a <= 0xfffffffful) {
        const unsigned au = (unsigned)a, q = au / (unsigned)b;
        r = (int)(au - q * (unsigned)b);
        return (long)q;
    }
    r = (int)(a % b);
    return a / b;
}

// Resolved (pointer-carrying) tensor view handed to kernels.
struct TView {
    char* ptr;
    int n, h, w, c, ld, esize;
};

// Kernel launchers implemented in the .hip files; all enqueue on `st`.
struct ConvArgs {
    TView in, res, out;
    const half_t* w;      // tiled [K/64][Np][64]
    const float* bias;    // [Np]
    const half_t* zero;   // >= 64 bytes of device zeros
    int kh, kw, sh, sw, ph, pw, act, act2, Np, Kp, inshift, resshift, cinp, flags;
    float act_a, act_b, post_a, post_b;
    // F_DOT1: fused 1x1 projection to one channel
    const float* dotw;
    float dotb;
    int dotact;
    TView dot_out;
    // F_SRC2: second input source of a virtual channel concat
    TView in2;
    int in2shift;
    // ragged plans: per-sample output widths (device, [n]); output pixels at x >= wl_out[n] are written as zeros
    const int* wl_out;
    int lo_off;           // P_LO_OUT: channel offset of the lo half of an fp16 hi + lo pair output (0 = plain fp16)
    int res_lo_off;       // P_LO_RES: ... of the residual
    int in_lo_off;        // P_LO_IN: ... of the input (F_DWPRE)
    // F_U8SRC: the uint8 BGR frames the stem conv pre-processes itself (vse_plan_set_source)
    const uint8_t* u8src;
    int u8_h, u8_w;
    long u8_pitch, u8_fstride;
};
int launch_conv(const ConvArgs& a, hipStream_t st);
int conv_tile_bn(int Np);   // which conv_mfma_kernel instantiation (BN = 128 / 64 / 32) serves Np output channels
// wl_in / wl_out: per-sample widths of in0 / of the output in a ragged plan (device, [n]), else nullptr
int launch_lstm_mfma(const TView& gf, const TView& gr, const TView& out, const half_t* whh, int rev_single, int ndir, int waves,
                     const int* tl, hipStream_t st);
// OP_CHAIN (chain.hip): in0 = the chain's input tensor, out / out2 / out3 = the tensors it stores (out3 travels in the op's in2 slot)
int launch_chain(const vse_op& op, const TView& in0, const TView& out, const TView& out2, const TView& out3, const char* wbase, hipStream_t st);
int launch_simple_op(const vse_op& op, const TView& in0, const TView& in1, const TView& in2, const TView& out,
                     const TView& out2, const char* wbase, const int* wl_in, const int* wl_out, hipStream_t st);
