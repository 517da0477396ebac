// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950), NHWC fp16 in, fp32 accumulate.
//
// GEMM view:  D[cout][pixel] = sum_k  Wt[cout][k] * X[pixel][k],   k = (tap_y, tap_x, cin)
//   * "A" MFMA operand = weights (rows = cout), "B" operand = gathered activations (cols = pixel), so each
//     lane ends up holding 4 *consecutive output channels* of one pixel per accumulator quad -> 8-byte
//     NHWC stores, per-channel bias as a float4, residual loads of the same shape.
//   * v_mfma_f32_32x32x16_f16: lane l supplies row/col (l&31) and k-slice 8*(l>>5)..+7 of both operands.
//   * block = 256 threads = 4 waves arranged WM x WN over a BM(pixels) x BN(couts) tile, BK = 32 per step,
//     register-staged double-buffered LDS with 80-byte rows (16-byte pad => conflict-free ds_read_b128).
//   * activations are gathered as one 16-byte vector per (pixel, tap, 8-channel group): cin is padded to a
//     multiple of 8 by the compiler, so a vector never straddles taps; image borders, K padding and the
//     M tail are predicated to zero.  A nearest-upsampled input (FPN) is gathered with (y>>s, x>>s).
//   * weights arrive pre-tiled [K/32][Np][32] so a BN x 32 tile is one contiguous, fully coalesced chunk.
//   * epilogue: + bias (BN folded) -> activation -> scalar affine -> (+ residual, optionally upsampled)
//     -> activation2 -> fp16 (or fp32) store; 2x2/stride-2 transposed conv = same GEMM with a
//     pixel-shuffle store.
#include "common.h"

struct ConvParams {
    const half_t* in;
    const half_t* w;
    const float* bias;
    const half_t* res;
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
};

#define LDS_ROW 40   // halfs per LDS row (32 data + 8 pad)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NA = BM / 64;                 // 16-byte activation vectors per thread per K step
    constexpr int NB = (BN * 4 + 255) / 256;    // 16-byte weight vectors per thread per K step
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile shape");

    __shared__ __attribute__((aligned(16))) half_t As[2][BM][LDS_ROW];
    __shared__ __attribute__((aligned(16))) half_t Bs[2][BN][LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (each XCD has a private 4 MiB L2).  Give every
    // XCD one CONTIGUOUS range of logical tiles (bijective for any tile count) and walk the N tiles of one pixel
    // tile first, so the kh x kw halo rows shared by neighbouring pixel tiles and the re-read of the activation
    // tile by the other cout tiles hit the same L2 instead of being fetched once per XCD.  Speed only: any
    // placement computes the same result.
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned mtile = logical / p.ntn, ntile = logical - mtile * p.ntn;
    const long m0 = (long)mtile * BM;
    const int n0 = ntile * BN;

    // ---- per-thread gather state -------------------------------------------------------------------
    const int kv = tid & 3;
    int ih0[NA], iw0[NA];
    long pbase[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const long m = m0 + (tid >> 2) + 64 * j;
        if (m < p.M) {
            const int ow = (int)(m % p.OW);
            const long t = m / p.OW;
            const int oh = (int)(t % p.OH);
            const long n = t / p.OH;
            ih0[j] = oh * p.sh - p.ph;
            iw0[j] = ow * p.sw - p.pw;
            pbase[j] = n * p.Hs * p.Ws;
        } else {
            ih0[j] = -(1 << 28);
            iw0[j] = 0;
            pbase[j] = 0;
        }
    }
    int kc = kv * 8, dy = 0, dx = 0;
    while (kc >= p.cinp) {
        kc -= p.cinp;
        if (++dx == p.kw) { dx = 0; ++dy; }
    }

    half8 ra[NA], rb[NB];
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int ih = ih0[j] + dy, iw = iw0[j] + dx;
            const bool ok = (dy < p.kh) && (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.W);
            if (ok) {
                const long pix = pbase[j] + (long)(ih >> p.inshift) * p.Ws + (iw >> p.inshift);
                ra[j] = *reinterpret_cast<const half8*>(p.in + pix * p.in_ld + kc);
            } else {
                ra[j] = zero8;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int v = tid + 256 * j;
            const int row = v >> 2;
            const int n = n0 + row;
            if (row < BN && n < p.Np) {
                rb[j] = *reinterpret_cast<const half8*>(p.w + ((long)kt * p.Np + n) * 32 + (v & 3) * 8);
            } else {
                rb[j] = zero8;
            }
        }
        // advance the (tap, channel) cursor by one K step
        kc += 32;
        while (kc >= p.cinp) {
            kc -= p.cinp;
            if (++dx == p.kw) { dx = 0; ++dy; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            *reinterpret_cast<half8*>(&As[buf][(tid >> 2) + 64 * j][kv * 8]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int v = tid + 256 * j;
            if ((v >> 2) < BN) *reinterpret_cast<half8*>(&Bs[buf][v >> 2][(v & 3) * 8]) = rb[j];
        }
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 8;
    int cur = 0;
    for (int kt = 0; kt < p.nk; ++kt) {
        const bool more = (kt + 1 < p.nk);
        if (more) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 wf[TN], xf[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wf[j] = *reinterpret_cast<const half8*>(&Bs[cur][wn * WTN + j * 32 + frow][ks * 16 + fk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                xf[i] = *reinterpret_cast<const half8*>(&As[cur][wm * WTM + i * 32 + frow][ks * 16 + fk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    const bool pixshuf = p.flags & F_PIXSHUF;
    const bool has_res = p.flags & F_RES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * WTM + i * 32 + frow;
        if (m >= p.M) continue;
        int ow = 0, oh = 0;
        long n = 0;
        if (pixshuf || (has_res && p.resshift)) {
            ow = (int)(m % p.OW);
            const long t = m / p.OW;
            oh = (int)(t % p.OH);
            n = t / p.OH;
        }
        long res_pix = m;
        if (has_res && p.resshift)
            res_pix = (n * p.res_hs + (oh >> p.resshift)) * p.res_ws + (ow >> p.resshift);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = n0 + wn * WTN + j * 32 + q * 8 + (lane >> 5) * 4;
                if (c0 >= p.Np) continue;
                const float4v b4 = *reinterpret_cast<const float4v*>(p.bias + c0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][q * 4 + e] + b4[e];
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
    }
}

int conv_tile_bn(int Np) {
    // tile selection: minimise padded-N waste, prefer the widest tile on ties
    auto padded = [&](int bn) { return ((Np + bn - 1) / bn) * bn; };
    if (Np <= 32) return 32;
    if (padded(64) < padded(128)) return 64;
    return 128;
}

int launch_conv(const ConvArgs& a, hipStream_t st) {
    ConvParams p;
    p.in = reinterpret_cast<const half_t*>(a.in.ptr);
    p.w = a.w;
    p.bias = a.bias;
    p.res = reinterpret_cast<const half_t*>(a.res.ptr);
    p.out = a.out.ptr;
    p.Hs = a.in.h;
    p.Ws = a.in.w;
    p.H = a.in.h << a.inshift;
    p.W = a.in.w << a.inshift;
    p.in_ld = a.in.ld;
    p.cinp = a.cinp;
    p.inshift = a.inshift;
    p.kh = a.kh; p.kw = a.kw; p.sh = a.sh; p.sw = a.sw; p.ph = a.ph; p.pw = a.pw;
    p.OH = (p.H + 2 * a.ph - a.kh) / a.sh + 1;
    p.OW = (p.W + 2 * a.pw - a.kw) / a.sw + 1;
    p.M = (long)a.in.n * p.OH * p.OW;
    p.Np = a.Np;
    p.nk = a.Kp / 32;
    p.out_ld = a.out.ld;
    p.out_f32 = (a.flags & F_OUT_F32) ? 1 : 0;
    p.res_ld = a.res.ld;
    p.resshift = a.resshift;
    p.res_hs = a.res.h;
    p.res_ws = a.res.w;
    p.act = a.act; p.act2 = a.act2;
    p.act_a = a.act_a; p.act_b = a.act_b; p.post_a = a.post_a; p.post_b = a.post_b;
    p.flags = a.flags;
    p.coutp = (a.flags & F_PIXSHUF) ? a.Np / 4 : a.Np;
    if (a.in.esize != 2 || (a.in.ld & 7) || (a.cinp & 7) || a.in.c != a.cinp) return VSE_E_INVAL;
    if ((a.flags & F_RES) && (a.res.esize != 2 || (a.res.ld & 3))) return VSE_E_INVAL;
    if ((a.out.ld & 3) || (a.Np & 7)) return VSE_E_INVAL;
    // sanity on the output view: [n, OH(*2), OW(*2)]
    const int mul = (a.flags & F_PIXSHUF) ? 2 : 1;
    if (a.out.h != p.OH * mul || a.out.w != p.OW * mul || a.out.n != a.in.n) return VSE_E_INVAL;

    const int bn = conv_tile_bn(a.Np);
    dim3 block(256);
    const int bm = bn == 32 ? 256 : 128;
    p.ntn = (unsigned)((a.Np + bn - 1) / bn);
    const unsigned long long tiles = (unsigned long long)((p.M + bm - 1) / bm) * p.ntn;
    if (tiles == 0 || tiles > 0x7fffffffull) return VSE_E_INVAL;
    dim3 grid((unsigned)tiles);
    if (bn == 128) hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 2, 2>), grid, block, 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_mfma_kernel<128, 64, 2, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_mfma_kernel<256, 32, 4, 1>), grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
