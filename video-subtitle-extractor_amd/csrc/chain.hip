// chain_kernel — a CHAIN of 1x1 convs and depthwise k x k convs (the inverted-residual / PP-LCNetV3 units of the mobile detectors, the
// DB head's two transposed convs) evaluated tile by tile with every intermediate tensor in LDS (OP_CHAIN, ir.py).
//
// Why: the reference's DEFAULT mode runs the mobile models (backend/tools/paddle_model_config.py:53-58).  Layer by
// layer they are HBM-bound byte shuffling: expand 1x1 -> depthwise -> project 1x1 writes and re-reads the widest tensors of the net
// three times at 272 x 480 ... 68 x 120, and every one of those round trips also ROUNDS the tensor to fp16 — the roundings
// tools/act_rounding_study.py traced the detector's box residue (IoU 0.955-0.961 on 3 of 183 boxes) to.  Here a block owns an output
// tile, walks the stages back to front once on the host (compiler.py: regions with their halos) and front to back on the device:
//
//   stage input/outputs in LDS   1x1 conv (PW) input: channel-minor fp16 as a hi + lo PAIR (x = hi + lo to ~22 bits): the MFMA B operand
//                                is one ds_read_b128 per 16-deep K slice; pixel stride 2 Cp + 16 bytes (an odd number of 16-byte slots:
//                                conflict-free); depthwise (DW) input: planar fp32 [channel][region pixel] (consecutive lanes =
//                                consecutive pixels: conflict-free for the MFMA D layout that writes it and the taps that read it).
//   PW                           v_mfma_f32_32x32x16_f16, three passes W_hi x_hi + W_lo x_hi + W_hi x_lo into one fp32 accumulator tile
//                                (the fp16 pair arithmetic the engine's F_HILO layers already use for weights, extended to the
//                                activations): nothing between a chain's input and its output is rounded to 11 bits.  Weight fragments
//                                are staged once per block in exactly lane order (1 KiB per (cout tile, K slice, pass)).
//   DW                           fp32 VALU from the planar buffer, weights as broadcast 16-byte LDS reads (per-channel records in the LDS
//                                image), 8 channels per lane -> one hi + lo 16-byte pair per pixel for the next PW.
//   zero padding                 a producer whose consumer is a depthwise conv writes ZEROS at region pixels outside the image, so the
//                                taps of the next stage see what the reference's padding gives them.
//   HBM                          the chain input once (with the halo of all later depthwise stages; L2 serves the overlap between
//                                neighbouring tiles, XCD-contiguous tile order) and each tensor some op outside the chain reads, once.
//
// Bound: HBM (algorithmic bytes = chain input + stored outputs); the matrix work is 2-6 % of the MFMA peak at these channel counts.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "conv_common.h"

#ifndef VSE_CHAIN_ABL
#define VSE_CHAIN_ABL 0      // timing-only ablations (tools/ablate_chain.sh; results are garbage): 1 no global stores, 2 no depthwise taps,
#endif                       // 4 no MFMAs, 8 no activation / hi-lo split in the PW epilogue (bit mask)

namespace {

enum { CH_MAGIC = 0x43484e31, CH_HDR = 16, CH_BUF = 16, CH_STAGE = 28, CH_MAX_STAGES = 8, CH_MAX_BUFS = 10 };
enum { B_KIND = 0, B_OFF_HI, B_OFF_LO, B_C, B_CP, B_STRIDE, B_TH, B_AH, B_EH, B_TW, B_AW, B_EW, B_P, B_HIMG, B_WIMG };
enum { S_TYPE = 0, S_IN, S_OUT, S_RES, S_CIN, S_COUT, S_NKS, S_NCT, S_K, S_S, S_PAD, S_ACT, S_ACT_A, S_ACT_B, S_POST_A, S_POST_B,
       S_WLDS, S_BLDS, S_DWW, S_DWB, S_GOUT, S_MASK, S_HASLO, S_ACT2, S_SHUF };

struct GView {          // a global NHWC fp16 tensor; lo_off != 0: the lo half of a hi + lo pair sits lo_off channels behind the hi half
    half_t* ptr;
    int ld, c, lo_off, h, w;
};

struct ChainArgs {
    const int* desc;            // header + buffers + stages (device, inside the weight blob)
    const char* blob;           // base of the chain's blob (LDS image and depthwise weights at offsets given by the header)
    GView in, out[3];
    float* out_f32;             // S_SHUF store: the 1-channel fp32 map [n][4 H][4 W]
    int n_img;
    int desc_words;             // header + buffers + stages
    unsigned long long* trace;  // -DVSE_CHAIN_TRACE builds: s_memtime stamps of block 0's first tiles
};

// Block barrier that orders LDS traffic ONLY.  __syncthreads() is a fence + barrier: hipcc puts s_waitcnt vmcnt(0) in front of it, so
// every stage boundary waited for the tile's global STORES to reach memory and for the next tile's prefetch LOADS to return
// (s_memtime trace, tools/trace_chain.sh: 12 k + 17 k cycles per tile of a two-stage chain, against ~3 k of work).  Nothing global
// needs ordering inside the kernel: inputs reach LDS through registers (the compiler waits for those), outputs are write-only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int fdiv(int p, float inv) { return __float2int_rz(((float)p + 0.5f) * inv); }

__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (half_t)v[j];
        lo[j] = (half_t)(v[j] - (float)hi[j]);
    }
}

template <int K>
__device__ __forceinline__ void dw_taps(const float* __restrict__ plane, int ew_in, const float* w, float& acc) {
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) acc = fmaf(w[dy * K + dx], plane[dy * ew_in + dx], acc);
}

#ifndef VSE_CHAIN_LB
#define VSE_CHAIN_LB 3        // blocks per CU the register budget is sized for (146 VGPRs: 3 without spills; 4 spills 10 VGPRs)
#endif
__global__ __launch_bounds__(256, VSE_CHAIN_LB) void chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // The descriptor (<= 400 words) is copied into LDS once per block and read from there.  Read from global memory, hipcc turned the
    // per-stage reads into VECTOR loads + v_readfirstlane (the stage index is a loop variable): two or three DEPENDENT global round
    // trips at every stage start, each also draining the in-order vmcnt queue of the tile's stores and prefetch loads — 10-15 k of the
    // ~30 k cycles a two-stage tile took (s_memtime trace, tools/trace_chain.sh).
    __shared__ int sdesc[CH_HDR + CH_MAX_BUFS * CH_BUF + CH_MAX_STAGES * CH_STAGE];
    for (int i = threadIdx.x; i < a.desc_words; i += 256) sdesc[i] = a.desc[i];
    lds_barrier();
    const int* D = sdesc;
    const int nstages = D[1], nbufs = D[2];
    const int tiles_h = D[8], tiles_w = D[9];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PERSISTENT blocks: block b walks tiles v(b), v(b) + G, v(b) + 2 G, ... (G = gridDim.x; v = the XCD-aware bijective order of
    // conv_mfma.hip, so the G tiles of one round are neighbours and each XCD gets a contiguous run of them).  The weight image is
    // staged once per block, the descriptor stays in the scalar cache, and the first LU x 256 input vectors of the NEXT tile are
    // loaded into registers while this tile's stages run — a tile's global-load latency (2-3 us under load, a third of an
    // un-pipelined tile) is off the critical path.
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const unsigned vbid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned ntiles = (unsigned)a.n_img * tiles_h * tiles_w;

    const int* BF = D + CH_HDR;
    const int* ST = D + CH_HDR + nbufs * CH_BUF;

    // ---- the LDS image of the weights (batched loads: one round trip per 4 x 16 bytes per thread) --------------------------------
    {
        const int nb16 = D[3] >> 4;
        const int4* __restrict__ wsrc = reinterpret_cast<const int4*>(a.blob + D[4]);
        for (int i0 = tid; i0 < nb16; i0 += 256 * 4) {
            int4 wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = wsrc[min(i0 + u * 256, nb16 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < nb16) reinterpret_cast<int4*>(lds)[i0 + u * 256] = wv[u];
        }
    }

    // ---- chain input: region of buffer 0 from global NHWC fp16 (zeros outside the image) -------------------------------------------
    // Loads come from CLAMPED addresses (an `if` around a load makes hipcc branch and wait per element) and are selected afterwards.
    constexpr int LU = 2;
    const int b0_kind = BF[B_KIND], b0_C = BF[B_C], b0_Cp = BF[B_CP], b0_stride = BF[B_STRIDE], b0_P = BF[B_P], b0_ew = BF[B_EW];
    const int b0_hi = BF[B_OFF_HI], b0_lo = BF[B_OFF_LO], b0_th = BF[B_TH], b0_ah = BF[B_AH], b0_tw = BF[B_TW], b0_aw = BF[B_AW];
    const int b0_H = BF[B_HIMG], b0_W = BF[B_WIMG];
    const float b0_inv_p = 1.0f / (float)b0_P, b0_inv_ew = 1.0f / (float)b0_ew;
    const int b0_ng = (b0_kind == 0 ? b0_Cp : b0_C) >> 3, b0_total = b0_P * b0_ng;
    const int in_c = a.in.c, in_ld = a.in.ld, in_lo = a.in.lo_off;
    // item `it` of tile (ni, tyi, txi): -> address of its 16 bytes (clamped), pixel, channel group, whether it is real data
    auto in_item = [&](int it, int ni, int tyi, int txi, int& pix, int& g, bool& ok) -> const half_t* {
        const int itc = it < b0_total ? it : b0_total - 1;
        g = fdiv(itc, b0_inv_p);
        pix = itc - g * b0_P;                                         // pixel fastest: consecutive lanes = consecutive pixels
        const int ry = fdiv(pix, b0_inv_ew), rx = pix - ry * b0_ew;
        const int iy = tyi * b0_th - b0_ah + ry, ix = txi * b0_tw - b0_aw + rx;
        ok = it < b0_total && iy >= 0 && iy < b0_H && ix >= 0 && ix < b0_W && g * 8 < in_c;
        if (it >= b0_total) pix = -1;
        const int cy = min(max(iy, 0), b0_H - 1), cx = min(max(ix, 0), b0_W - 1), cg = min(g * 8, in_c - 8);
        return a.in.ptr + (((size_t)ni * b0_H + cy) * b0_W + cx) * in_ld + cg;
    };
    auto in_store = [&](int pix, int g, bool ok, const half8& hv, const half8& lv) {
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const half8 vh = ok ? hv : z, vl = (ok && in_lo) ? lv : z;
        if (b0_kind == 0) {
            *reinterpret_cast<half8*>(lds + b0_hi + pix * b0_stride + g * 16) = vh;
            if (b0_lo >= 0) *reinterpret_cast<half8*>(lds + b0_lo + pix * b0_stride + g * 16) = vl;
        } else {
            float* pl = reinterpret_cast<float*>(lds + b0_hi) + (size_t)(g * 8) * b0_stride + pix;
#pragma unroll
            for (int j = 0; j < 8; ++j) pl[j * b0_stride] = (float)vh[j] + (float)vl[j];
        }
    };
    auto decode = [&](unsigned t, int& ni, int& tyi, int& txi) {
        txi = t % tiles_w; t /= tiles_w;
        tyi = t % tiles_h;
        ni = t / tiles_h;
    };
    half8 pf[LU];                                  // the next tile's first LU x 256 input vectors (hi halves), in flight across the stages
    unsigned tile = vbid;
    if (tile < ntiles) {
        int ni, tyi, txi;
        decode(tile, ni, tyi, txi);
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            int pix, g; bool ok;
            pf[u] = *reinterpret_cast<const half8*>(in_item(tid + u * 256, ni, tyi, txi, pix, g, ok));
        }
    }
#ifdef VSE_CHAIN_TRACE
    int trace_i = 0;
#define CH_STAMP() do { if (a.trace && bid == 8 && tid == 0 && trace_i < 60) a.trace[trace_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#define CH_FINE(slot) do { if (a.trace && bid == 8 && tid == 0 && tile == vbid) a.trace[64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_STAMP() do { } while (0)
#define CH_FINE(slot) do { } while (0)
#endif
  for (; tile < ntiles; tile += nblk) {
    int n, ty, tx;
    decode(tile, n, ty, tx);
    CH_STAMP();
    {
#pragma unroll
        for (int u = 0; u < LU; ++u) {                 // the prefetched batch
            int pix, g; bool ok;
            const half_t* q = in_item(tid + u * 256, n, ty, tx, pix, g, ok);
            half8 lv = pf[u];
            if (in_lo && pix >= 0) lv = *reinterpret_cast<const half8*>(q + in_lo);
            if (pix >= 0) in_store(pix, g, ok, pf[u], lv);
        }
        for (int it0 = tid + 256 * LU; it0 < b0_total; it0 += 256 * LU) {      // regions with more than LU x 256 vectors: the rest, batched
            half8 hv[LU], lv[LU];
            int pixs[LU], gs[LU];
            bool oks[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const half_t* q = in_item(it0 + u * 256, n, ty, tx, pixs[u], gs[u], oks[u]);
                hv[u] = *reinterpret_cast<const half8*>(q);
                lv[u] = *reinterpret_cast<const half8*>(q + in_lo);       // (in_lo = 0: the same line again; selected away in in_store)
            }
#pragma unroll
            for (int u = 0; u < LU; ++u)
                if (pixs[u] >= 0) in_store(pixs[u], gs[u], oks[u], hv[u], lv[u]);
        }
    }
    lds_barrier();
    CH_STAMP();
    if (tile + nblk < ntiles) {
        int ni, tyi, txi;
        decode(tile + nblk, ni, tyi, txi);
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            int pix, g; bool ok;
            pf[u] = *reinterpret_cast<const half8*>(in_item(tid + u * 256, ni, tyi, txi, pix, g, ok));
        }
    }

    for (int si = 0; si < nstages; ++si) {
        // every descriptor word the stage needs is read HERE, into scalar registers: a scalar load inside the loops below could not
        // be hoisted past their stores (it may alias them as far as the compiler knows) and costs a ~200-cycle wait each
        CH_FINE(si * 8 + 4);
        // the stage record and its three buffer records: TWO lane-indexed LDS reads, then v_readlane per field (~60 separate
        // ds_read + v_readfirstlane pairs cost 1.2-3 k cycles per stage and tile in the s_memtime trace)
        const int fs = ST[si * CH_STAGE + min(lane, CH_STAGE - 1)];
#define SF(k) __builtin_amdgcn_readlane(fs, (k))
        const int s_type = SF(S_TYPE), s_in = SF(S_IN), ob = SF(S_OUT), s_res = SF(S_RES);
        const int act = SF(S_ACT), act2 = SF(S_ACT2), gout = SF(S_GOUT), cout = SF(S_COUT), cin = SF(S_CIN);
        const int s_mask = SF(S_MASK), s_shuf = SF(S_SHUF), haslo = SF(S_HASLO), nks = SF(S_NKS), nct = SF(S_NCT);
        const int s_wlds = SF(S_WLDS), s_blds = SF(S_BLDS), K = SF(S_K), S = SF(S_S);
        const float act_a = __int_as_float(SF(S_ACT_A)), act_b = __int_as_float(SF(S_ACT_B));
        const float post_a = __int_as_float(SF(S_POST_A)), post_b = __int_as_float(SF(S_POST_B));
        const bool post = post_a != 1.0f || post_b != 0.0f;
        const int fb = BF[(lane < 16 ? s_in : (lane < 32 ? ob : (s_res >= 0 ? s_res : s_in))) * CH_BUF + (lane & 15)];
#define BI(k) __builtin_amdgcn_readlane(fb, (k))
#define BO(k) __builtin_amdgcn_readlane(fb, 16 + (k))
#define BR(k) __builtin_amdgcn_readlane(fb, 32 + (k))
        const int bi_hi = BI(B_OFF_HI), bi_lo = BI(B_OFF_LO), bi_stride = BI(B_STRIDE), bi_P = BI(B_P), bi_ew = BI(B_EW);
        const int bi_th = BI(B_TH), bi_ah = BI(B_AH), bi_tw = BI(B_TW), bi_aw = BI(B_AW), bi_H = BI(B_HIMG), bi_W = BI(B_WIMG);
        const int bo_kind = BO(B_KIND);
        const bool to_lds = bo_kind != 2;
        const int bo_hi = BO(B_OFF_HI), bo_lo = BO(B_OFF_LO), bo_stride = BO(B_STRIDE), bo_cp = BO(B_CP);
        const int bo_P = BO(B_P), bo_ew = BO(B_EW), bo_th = BO(B_TH), bo_ah = BO(B_AH), bo_tw = BO(B_TW), bo_aw = BO(B_AW);
        const int bo_H = BO(B_HIMG), bo_W = BO(B_WIMG);
        const int br_hi = BR(B_OFF_HI), br_lo = BR(B_OFF_LO), br_stride = BR(B_STRIDE), br_ew = BR(B_EW);
        const int br_dy = BR(B_AH) - bi_ah, br_dx = BR(B_AW) - bi_aw;
        const GView gv = a.out[gout >= 0 ? gout : 0];
        if (s_type == 0) {
            // ================= 1x1 conv: D[cout][pixel] = W[cout][k] X[k][pixel], three hi / lo passes ==================
            const int P = bi_P, ew = bi_ew, stride = bi_stride;
            const int npt = (P + 31) >> 5;
            const char* xhi = lds + bi_hi;
            const char* xlo = lds + (bi_lo >= 0 ? bi_lo : bi_hi);
            const char* wfr = lds + s_wlds;
            const float* bias = reinterpret_cast<const float*>(lds + s_blds);
            const int y0 = ty * bi_th - bi_ah, x0 = tx * bi_tw - bi_aw, H = bi_H, W = bi_W;
            const float inv_ew = 1.0f / (float)ew;
            const int h = lane >> 5;
            const size_t lo_pass = (size_t)nct * nks * 1024;
            for (int it = wave; it < npt * nct; it += 4) {
                const int pt = it / nct, ct = it - pt * nct;           // (uniform per wave)
                if (it == wave) CH_FINE(si * 8 + 0);
                const int pix_raw = pt * 32 + (lane & 31);
                const int pix = pix_raw < P ? pix_raw : P - 1;
                float16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const char* xb = xhi + pix * stride + h * 16;
                const char* xl = xlo + pix * stride + h * 16;
                const char* wb = wfr + ((size_t)ct * nks * 64 + lane) * 16;
                if (VSE_CHAIN_ABL & 4) {
                } else if (haslo) {
                    for (int ks = 0; ks < nks; ++ks) {
                        const half8 bh = *reinterpret_cast<const half8*>(xb + ks * 32);
                        const half8 bl = *reinterpret_cast<const half8*>(xl + ks * 32);
                        const half8 ah = *reinterpret_cast<const half8*>(wb + ks * 1024);
                        const half8 al = *reinterpret_cast<const half8*>(wb + ks * 1024 + lo_pass);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                    }
                } else {

                    for (int ks = 0; ks < nks; ++ks) {
                        const half8 bh = *reinterpret_cast<const half8*>(xb + ks * 32);
                        const half8 ah = *reinterpret_cast<const half8*>(wb + ks * 1024);
                        const half8 al = *reinterpret_cast<const half8*>(wb + ks * 1024 + lo_pass);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                    }
                }
                if (it == wave) CH_FINE(si * 8 + 1);
                // ---- epilogue: lane = pixel (lane & 31); registers 8g + j = channel ct*32 + 16g + 8h + j (conv_wrow order) ----
                const int ry = fdiv(pix, inv_ew), rx = pix - ry * ew;
                const int iy = y0 + ry, ix = x0 + rx;
                const bool inimg = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const bool live = pix_raw < P;
                const int oy = ry - bi_ah, ox = rx - bi_aw;                 // position inside the tile's owned pixels
                const bool store = gout >= 0 && inimg && live && oy >= 0 && oy < bi_th && ox >= 0 && ox < bi_tw;
                const int rpix = (ry + br_dy) * br_ew + rx + br_dx;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c0 = ct * 32 + 16 * g + 8 * h;
                    if (c0 >= cout) continue;
                    float v[8];
                    const float4v b0 = *reinterpret_cast<const float4v*>(bias + c0), b1 = *reinterpret_cast<const float4v*>(bias + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[8 * g + j] + b0[j]; v[4 + j] = acc[8 * g + 4 + j] + b1[j]; }
                    if (!(VSE_CHAIN_ABL & 8)) vse_act_n<8>(v, act, act_a, act_b);
                    if (post) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = v[j] * post_a + post_b;
                    }
                    if (s_res >= 0) {
                        // the residual: a channel-minor buffer of the same resolution, possibly with a wider halo
                        const half8 rh = *reinterpret_cast<const half8*>(lds + br_hi + rpix * br_stride + c0 * 2);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)rh[j];
                        if (br_lo >= 0) {
                            const half8 rl = *reinterpret_cast<const half8*>(lds + br_lo + rpix * br_stride + c0 * 2);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)rl[j];
                        }
                        if (act2 == ACT_RELU) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                    }
                    if (s_mask && !inimg) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = 0.f;
                    }
                    if (!live) continue;
                    if (to_lds) {
                        if (bo_kind == 1) {
                            float* pl = reinterpret_cast<float*>(lds + bo_hi) + (size_t)c0 * bo_stride + pix;
#pragma unroll
                            for (int j = 0; j < 8; ++j) pl[j * bo_stride] = v[j];
                        } else {
                            half8 hi, lo;
                            split8(v, hi, lo);
                            *reinterpret_cast<half8*>(lds + bo_hi + pix * bo_stride + c0 * 2) = hi;
                            if (bo_lo >= 0) *reinterpret_cast<half8*>(lds + bo_lo + pix * bo_stride + c0 * 2) = lo;
                        }
                    }
                    if (store && !(VSE_CHAIN_ABL & 1)) {
                        if (s_shuf) {
                            // head tail: channel 4 r + c = output (4 iy + r, 4 ix + c) of the 4 x 4 block of input pixel (iy, ix); this
                            // lane's run c0 .. c0 + 7 = rows c0 / 4 and c0 / 4 + 1 of the block -> fp32 map [n][4 H][4 W]
                            float* om = a.out_f32 + ((size_t)n * 4 * H + 4 * iy) * (4 * (size_t)W) + 4 * ix;
                            const int row0 = (c0 >> 2);
#pragma unroll
                            for (int rr = 0; rr < 2; ++rr)
                                *reinterpret_cast<float4v*>(om + (size_t)(row0 + rr) * 4 * W) = float4v{v[4 * rr], v[4 * rr + 1], v[4 * rr + 2], v[4 * rr + 3]};
                        } else if (c0 < gv.c) {
                            half_t* gp = gv.ptr + (((size_t)n * H + iy) * W + ix) * gv.ld + c0;
                            half8 hi, lo;
                            split8(v, hi, lo);
                            *reinterpret_cast<half8*>(gp) = hi;
                            if (gv.lo_off) *reinterpret_cast<half8*>(gp + gv.lo_off) = lo;
                        }
                    }
                }
                if (it == wave) CH_FINE(si * 8 + 2);
                // zero K padding of a channel-minor output (Cp > cout): garbage there would meet zero weights, but NaN * 0 = NaN
                if (bo_kind == 0 && live && ct == nct - 1 && bo_cp > cout && h == 0) {
                    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int c0 = cout; c0 < bo_cp; c0 += 8) {
                        *reinterpret_cast<half8*>(lds + bo_hi + pix * bo_stride + c0 * 2) = z;
                        if (bo_lo >= 0) *reinterpret_cast<half8*>(lds + bo_lo + pix * bo_stride + c0 * 2) = z;
                    }
                }
            }
        } else {
            // ================= depthwise k x k, stride s: planar fp32 in, 8 channels per lane out ==============================
            // weights + bias: records of DWREC(K) floats per channel in the LDS image ([k*k weights, bias, zero padding]), read as
            // broadcast 16-byte vectors (every lane the same address): no scalar-cache round trip inside the loop
            const int C = cin;
            const int ew_in = bi_ew, pstride = bi_stride;
            const int Po = bo_P, ewo = bo_ew;
            const float inv_ewo = 1.0f / (float)ewo;
            const float* pin = reinterpret_cast<const float*>(lds + bi_hi);
            const float* wl = reinterpret_cast<const float*>(lds + s_wlds);
            const int y0 = ty * bo_th - bo_ah, x0 = tx * bo_tw - bo_aw, H = bo_H, W = bo_W;
            const int nchunk = (Po + 63) >> 6, ncg = C >> 3;
            for (int it = wave; it < nchunk * ncg; it += 4) {
                const int cg = it / nchunk, chunk = it - cg * nchunk;
                if (it == wave) CH_FINE(si * 8 + 0);
                const int pix_raw = chunk * 64 + lane;
                const int pix = pix_raw < Po ? pix_raw : Po - 1;
                const int ry = fdiv(pix, inv_ewo), rx = pix - ry * ewo;
                const float* base = pin + (size_t)(cg * 8) * pstride + (ry * S) * ew_in + rx * S;
                float v[8];
                if (K == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4v* wr = reinterpret_cast<const float4v*>(wl + (cg * 8 + j) * 12);
                        const float4v w0 = wr[0], w1 = wr[1], w2 = wr[2];
                        const float w[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
                        float acc = w2[1];
                        if (!(VSE_CHAIN_ABL & 2)) dw_taps<3>(base + (size_t)j * pstride, ew_in, w, acc);
                        v[j] = acc;
                    }
                } else {
                    // 5 x 5: the record is read row by row (25 weights + bias in registers at once cost 28 VGPRs per channel in flight)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float* wr = wl + (cg * 8 + j) * 28;
                        const float* pl = base + (size_t)j * pstride;
                        float acc = wr[25];
                        if (!(VSE_CHAIN_ABL & 2)) {
#pragma unroll
                            for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
                                for (int dx = 0; dx < 5; ++dx) acc = fmaf(wr[dy * 5 + dx], pl[dy * ew_in + dx], acc);
                            }
                        }
                        v[j] = acc;
                    }
                }
                if (it == wave) CH_FINE(si * 8 + 1);
                vse_act_n<8>(v, act, act_a, act_b);
                if (post) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] * post_a + post_b;
                }
                if (it == wave) CH_FINE(si * 8 + 2);
                if (pix_raw >= Po) continue;
                half8 hi, lo;
                split8(v, hi, lo);
                if (to_lds) {
                    *reinterpret_cast<half8*>(lds + bo_hi + pix * bo_stride + cg * 16) = hi;
                    if (bo_lo >= 0) *reinterpret_cast<half8*>(lds + bo_lo + pix * bo_stride + cg * 16) = lo;
                    if (cg == ncg - 1 && bo_cp > C) {
                        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int c0 = C; c0 < bo_cp; c0 += 8) {
                            *reinterpret_cast<half8*>(lds + bo_hi + pix * bo_stride + c0 * 2) = z;
                            if (bo_lo >= 0) *reinterpret_cast<half8*>(lds + bo_lo + pix * bo_stride + c0 * 2) = z;
                        }
                    }
                }
                if (gout >= 0 && !(VSE_CHAIN_ABL & 1)) {
                    const int iy = y0 + ry, ix = x0 + rx;
                    const int oy = ry - bo_ah, ox = rx - bo_aw;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W && oy >= 0 && oy < bo_th && ox >= 0 && ox < bo_tw && cg * 8 < gv.c) {
                        half_t* gp = gv.ptr + (((size_t)n * H + iy) * W + ix) * gv.ld + cg * 8;
                        *reinterpret_cast<half8*>(gp) = hi;
                        if (gv.lo_off) *reinterpret_cast<half8*>(gp + gv.lo_off) = lo;
                    }
                }
            }
        }
        CH_FINE(si * 8 + 5);
        lds_barrier();
        CH_STAMP();
    }
  }   // tiles of this block
}

// REGISTER form of a two-stage 1x1 -> 1x1 chain (round 4; the DB head's tail: transposed conv 2x2 s2 + relu -> transposed conv 2x2 s2 +
// sigmoid, lowered by compiler.py try_lower_head_tail to PW(c0 -> 4 c1) -> PW(4 c1 -> 16) with the 4 x 4 pixel-shuffle fp32 store).
// Both stages act on the SAME pixel, so no tile, no halo and no LDS buffer is needed: the accumulator tile of stage A — lane = pixel,
// register 8 g + j = channel 32 ct + 16 g + 8 h + j, the conv_wrow order every conv kernel stores in — IS the B operand layout of the
// k slice 2 ct + g of stage B.  A wave takes 32 pixels, runs stage A one 32-channel tile at a time (bias, activation, fp16 hi + lo split
// in registers) and feeds each tile's two slices straight into stage B's accumulator; only the weight image is staged (once per block).
// It reads the SAME blob as chain_kernel (descriptor words + LDS image of MFMA fragments) and performs the same operations in the same
// order — bit-identical results — at ~0.1 ms instead of 0.5 ms per 64 x 136 x 240 on the mobile detectors (the generic kernel pays a
// descriptor-driven stage skeleton, an LDS round trip per stage and block barriers for what is 36 MFMAs per 32 pixels).
struct Pw2Args {
    const int* desc;
    const char* blob;
    const half_t* in;
    float* out_f32;
    long M;                     // N * H * W input pixels
    int H, W, in_ld, in_c, in_lo;
};

__global__ __launch_bounds__(256, 3) void chain_pw2_kernel(const Pw2Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int* D = a.desc;
    const int nbufs = D[2];
    const int* SA = D + CH_HDR + nbufs * CH_BUF;
    const int* SB = SA + CH_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int nb16 = D[3] >> 4;
        const int4* __restrict__ wsrc = reinterpret_cast<const int4*>(a.blob + D[4]);
        // (all of a thread's loads in flight before its first LDS write: stage_batched, common.h)
        stage_batched<8>(nb16, tid, [&](int i) { return wsrc[i]; }, [&](int i, int4 x) { reinterpret_cast<int4*>(lds)[i] = x; });
    }
    const int nksA = SA[S_NKS], nctA = SA[S_NCT], coutA = SA[S_COUT], actA = SA[S_ACT], hasloA = SA[S_HASLO];
    const float actA_a = __int_as_float(SA[S_ACT_A]), actA_b = __int_as_float(SA[S_ACT_B]);
    const int nksB = SB[S_NKS], coutB = SB[S_COUT], actB = SB[S_ACT], hasloB = SB[S_HASLO];
    const float actB_a = __int_as_float(SB[S_ACT_A]), actB_b = __int_as_float(SB[S_ACT_B]);
    const char* wA = lds + SA[S_WLDS];
    const char* wB = lds + SB[S_WLDS];
    const float* biasA = reinterpret_cast<const float*>(lds + SA[S_BLDS]);
    const float* biasB = reinterpret_cast<const float*>(lds + SB[S_BLDS]);
    const size_t loA = (size_t)nctA * nksA * 1024, loB = (size_t)nksB * 1024;         // (stage B has one cout tile)
    __syncthreads();
    const int h = lane >> 5, fx = lane & 31;
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const long mraw = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + wave * 64 + i * 32 + fx;
        const long m = mraw < a.M ? mraw : 0;
        // stage A's input: this pixel's channels, 8 per lane half and K slice, zero behind the real channels (the chain kernel's LDS
        // copy zero-fills them the same way)
        half8 bh[4], bl[4];
        const half_t* src = a.in + m * a.in_ld + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = z8;
            bl[ks] = z8;
            if (ks < nksA && ks * 16 + h * 8 < a.in_c) {
                bh[ks] = *reinterpret_cast<const half8*>(src + ks * 16);
                if (hasloA) bl[ks] = *reinterpret_cast<const half8*>(src + ks * 16 + a.in_lo);
            }
        }
        float16v acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int ct = 0; ct < nctA; ++ct) {
            float16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const char* wb = wA + ((size_t)ct * nksA * 64 + lane) * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks >= nksA) break;
                const half8 ah = *reinterpret_cast<const half8*>(wb + ks * 1024);
                const half8 al = *reinterpret_cast<const half8*>(wb + ks * 1024 + loA);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
                if (hasloA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ks2 = 2 * ct + g;
                if (ks2 >= nksB) continue;
                const int c0 = ct * 32 + 16 * g + 8 * h;
                half8 hi = z8, lo = z8;                       // (channels behind stage A's couts: the zero K padding of the chain's buffer)
                if (c0 < coutA) {
                    float v[8];
                    const float4v b0 = *reinterpret_cast<const float4v*>(biasA + c0), b1 = *reinterpret_cast<const float4v*>(biasA + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[8 * g + j] + b0[j]; v[4 + j] = acc[8 * g + 4 + j] + b1[j]; }
                    vse_act_n<8>(v, actA, actA_a, actA_b);
                    split8(v, hi, lo);
                }
                const char* w2 = wB + ((size_t)ks2 * 64 + lane) * 16;
                const half8 ah = *reinterpret_cast<const half8*>(w2);
                const half8 al = *reinterpret_cast<const half8*>(w2 + loB);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hi, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, hi, acc2, 0, 0, 0);
                if (hasloB) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, lo, acc2, 0, 0, 0);
            }
        }
        // stage B's 16 channels = the 4 x 4 block of output pixels of this input pixel: channel 4 r + c -> (4 iy + r, 4 ix + c); this
        // lane half's run 8 h .. 8 h + 7 = rows 2 h and 2 h + 1 of the block (register group 0 of the accumulator tile)
        const int c0 = 8 * h;
        if (mraw < a.M && c0 < coutB) {
            const unsigned mu = (unsigned)m, t = mu / (unsigned)a.W, n = t / (unsigned)a.H;
            const int ix = (int)(mu - t * (unsigned)a.W), iy = (int)(t - n * (unsigned)a.H);
            float v[8];
            const float4v b0 = *reinterpret_cast<const float4v*>(biasB + c0), b1 = *reinterpret_cast<const float4v*>(biasB + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc2[j] + b0[j]; v[4 + j] = acc2[4 + j] + b1[j]; }
            vse_act_n<8>(v, actB, actB_a, actB_b);
            float* om = a.out_f32 + ((size_t)n * 4 * a.H + 4 * iy) * (4 * (size_t)a.W) + 4 * ix;
            const int row0 = c0 >> 2;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                *reinterpret_cast<float4v*>(om + (size_t)(row0 + rr) * 4 * a.W) = float4v{v[4 * rr], v[4 * rr + 1], v[4 * rr + 2], v[4 * rr + 3]};
        }
    }
}

}  // namespace

// Host side: the descriptor (device memory, inside the weight blob) was written by compiler.py (Compiler.emit_chain); the words the
// launcher needs travel in the op record: p[0] tiles_h, p[1] tiles_w, p[2] dynamic LDS bytes, p[3] stages, p[4] buffers,
// p[10..13] channel offset of the lo half of the input / of outputs 0..2 (0 = a plain fp16 tensor).
int launch_chain(const vse_op& o, const TView& in0, const TView& out, const TView& out2, const TView& out3, const char* wbase,
                 hipStream_t st) {
    const int nstages = o.p[3], nbufs = o.p[4];
    if (nstages < 1 || nstages > CH_MAX_STAGES || nbufs < 1 || nbufs > CH_MAX_BUFS) return VSE_E_INVAL;
    // p[5] = 1 (compiler.py try_lower_head_tail): a 1x1 -> 1x1 chain with the pixel-shuffle map store — the register form
    // (chain_pw2_kernel; VSE_HEAD_PW2=0 keeps the generic kernel for A/B runs)
    static const bool pw2_on = !(vse_dev_getenv("VSE_HEAD_PW2") && atoi(vse_dev_getenv("VSE_HEAD_PW2")) == 0);
    // (an image above 64 KiB — the ResNet detector's 64 -> 4 x 64 -> 16 tail — stays on the generic kernel)
    if (o.p[5] == 1 && pw2_on && nstages == 2 && o.p[6] > 0 && o.p[6] <= 64 * 1024 && in0.c <= 64) {
        Pw2Args b;
        b.desc = reinterpret_cast<const int*>(wbase + o.w_off);
        b.blob = wbase + o.w_off;
        b.in = reinterpret_cast<const half_t*>(in0.ptr);
        b.out_f32 = reinterpret_cast<float*>(out.ptr);
        b.M = (long)in0.n * in0.h * in0.w;
        b.H = in0.h; b.W = in0.w; b.in_ld = in0.ld; b.in_c = in0.c; b.in_lo = o.p[10];
        const int img_bytes = o.p[6];
        if (b.M <= 0 || b.M >= 0x7fffffffl || in0.esize != 2 || (in0.c & 7)) return VSE_E_INVAL;
        const unsigned blocks = (unsigned)((b.M + 255) / 256);
        hipLaunchKernelGGL(chain_pw2_kernel, dim3(blocks), dim3(256), (size_t)img_bytes, st, b);
        return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
    }
    ChainArgs a;
    a.desc = reinterpret_cast<const int*>(wbase + o.w_off);
    a.blob = wbase + o.w_off;
    auto gv = [](const TView& t, int lo_off) {
        GView g;
        g.ptr = reinterpret_cast<half_t*>(t.ptr); g.ld = t.ld; g.c = t.c; g.lo_off = lo_off; g.h = t.h; g.w = t.w;
        return g;
    };
    a.in = gv(in0, o.p[10]);
    a.out[0] = gv(out, o.p[11]);
    a.out[1] = gv(out2, o.p[12]);
    a.out[2] = gv(out3, o.p[13]);
    a.out_f32 = reinterpret_cast<float*>(out.ptr);
    a.n_img = in0.n;
    a.desc_words = CH_HDR + nbufs * CH_BUF + nstages * CH_STAGE;
    const int tiles_h = o.p[0], tiles_w = o.p[1];
    const unsigned long long blocks = (unsigned long long)in0.n * tiles_h * tiles_w;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    const int lds_bytes = o.p[2];
    if (lds_bytes > 158 * 1024) return VSE_E_UNSUPPORTED;
    static VseDevOnce attr_once;          // (per device, thread-safe: common.h)
    if (!vse_dev_once(attr_once, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) == hipSuccess;      // (+ 1.6 KiB of static LDS: the descriptor)
        }))
        return VSE_E_HIP;
    // persistent grid: as many blocks as the chip holds at this LDS size (160 VGPRs: three 4-wave blocks per CU at most), a multiple of 8
    const int n_cu = vse_cu_count();
    if (!n_cu) return VSE_E_HIP;
    const int per_cu = std::max(1, std::min(VSE_CHAIN_LB, (160 * 1024) / (lds_bytes + 2048)));
    unsigned grid = (unsigned)std::min<unsigned long long>(blocks, (unsigned long long)n_cu * per_cu);
    if (grid > 8) grid &= ~7u;
#ifdef VSE_CHAIN_TRACE
    static unsigned long long* trace_dev = nullptr;
    if (!trace_dev) (void)hipMalloc(&trace_dev, 128 * sizeof(unsigned long long));
    (void)hipMemsetAsync(trace_dev, 0, 128 * sizeof(unsigned long long), st);
    a.trace = trace_dev;
#else
    a.trace = nullptr;
#endif
    hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), lds_bytes, st, a);
#ifdef VSE_CHAIN_TRACE
    {
        (void)hipStreamSynchronize(st);
        unsigned long long h[128];
        (void)hipMemcpy(h, trace_dev, sizeof h, hipMemcpyDeviceToHost);
        const int per = nstages + 2;       // stamps per tile: start, input stored, after each stage
        fprintf(stderr, "[chain trace] %d stages, grid %u, lds %d, tiles %llu: block 8, s_memtime ticks (shader clocks) per phase of its first tiles:", nstages, grid, lds_bytes, blocks);
        for (int t = 0; t < 4 && h[(t + 1) * per - 1]; ++t) {
            fprintf(stderr, "  | tile %d:", t);
            for (int q = 1; q < per; ++q) fprintf(stderr, " %llu", h[t * per + q] - h[t * per + q - 1]);
            if (h[(t + 1) * per]) fprintf(stderr, " (gap %llu)", h[(t + 1) * per] - h[(t + 1) * per - 1]);
        }
        fprintf(stderr, "  || wave 0, first tile, per stage (stage start->first item, item core, item epilogue, end of first item->loop done):");
        for (int q = 0; q < nstages; ++q) fprintf(stderr, " [%llu %llu %llu %llu]", h[64 + q * 8] - h[64 + q * 8 + 4], h[64 + q * 8 + 1] - h[64 + q * 8], h[64 + q * 8 + 2] - h[64 + q * 8 + 1], h[64 + q * 8 + 5] - h[64 + q * 8 + 2]);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
