// BiLSTM recurrence of the CRNN recogniser (V2/ch_rec: two bidirectional layers, 256 hidden units; the reference reaches it
// through paddleocr's TextRecognizer behind backend/tools/ocr.py:27, graph op `rnn` of backend/models/V2/ch_rec).
//
// The input projection x.W_ih^T + b of ALL time steps is one MFMA GEMM upstream (OP_CONV, fp32 out).  What is left per time step is
//     z[b, 4H] = gates_x[b, t] + h[b, H] . W_hh^T[H, 4H]
// — an MFMA GEMM over the BATCH (the round-2 kernel ran it per sample with scalar FMAs and re-read the 512 KiB W_hh per sample
// and step).  Design:
//   * one block per (direction, tile of 32 samples), 8 waves; wave w owns hidden units 32w .. 32w+31 and ALL FOUR gates of them
//     (four 32 x 32 accumulator tiles), so the cell update i/f/g/o -> c -> h is lane-local: the four gate values of (unit, sample)
//     sit in the same lane and register index of the four tiles.  c stays in registers for the whole sequence.
//   * W_hh^T is the MFMA A operand, packed by the compiler in fragment order [wave][k-slice][gate][lane][8]: a wave streams its own
//     64 KiB per step straight from L2 into VGPRs with contiguous 1 KiB wave loads (no wave shares rows with another, so there is
//     nothing to stage in LDS); the step time is that stream (~512 KiB per block and step), shared by the 32 samples of the tile.
//   * h is the B operand, kept in LDS as fp16 hi + lo (h = hi + lo to ~22 bits: two MFMAs per weight fragment, which is free
//     next to the weight stream) — the recurrence then carries fp32-grade state like the oracle, not fp16 roundings over T steps.
//   * the accumulators are INITIALISED with gates_x (16-byte fp32 loads), so the projection is never held twice.
//   * both directions in one launch (blockIdx.y); a sample's length may be its own (ragged batches): the reverse pass starts at
//     ITS last step, finished samples idle, outputs behind a sample's end are zeros.
// A sample's column of the product depends on that sample alone -> results do not depend on the batch composition.
#include "common.h"

#define LSTM_H 256
#define LSTM_LDH (LSTM_H + 8)      // padded LDS row (halfs)

__global__ __launch_bounds__(512, 2) void lstm_mfma_kernel(TView gates_f, TView gates_r, TView out, const half_t* __restrict__ whh,
                                                           int rev_single, int ndir, const int* __restrict__ tl) {
    __shared__ half_t hbuf[2][32][LSTM_LDH];      // [hi / lo][sample][hidden unit]
    __shared__ int s_tmax;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kq = lane >> 5;
    const int dir = blockIdx.y;
    const int rev = ndir == 2 ? dir : rev_single;
    const int B = gates_f.n, Tfull = gates_f.w, gld = gates_f.ld;       // (the launcher checks that both directions agree)
    const int b = blockIdx.x * 32 + n;
    const int T = b < B ? (tl != nullptr ? min(max(tl[b], 0), Tfull) : Tfull) : 0;
    if (threadIdx.x == 0) s_tmax = 0;
    for (int i = threadIdx.x; i < 2 * 32 * LSTM_LDH; i += blockDim.x) (&hbuf[0][0][0])[i] = (half_t)0.f;
    __syncthreads();
    if (wave == 0 && kq == 0) atomicMax(&s_tmax, T);
    __syncthreads();
    const int tmax = s_tmax;
    const half8* wfrag = reinterpret_cast<const half8*>(whh) + (size_t)dir * (8 * 4 * 16 * 64) + (size_t)wave * (4 * 16 * 64) + lane;
    const float* gbase = reinterpret_cast<const float*>((ndir == 2 && dir == 1) ? gates_r.ptr : gates_f.ptr);
    half_t* obase = reinterpret_cast<half_t*>(out.ptr) + (ndir == 2 ? dir * LSTM_H : 0);
    const int u0 = 32 * wave + 4 * kq;            // this lane's units: u0 + 8q + e
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    // The weight stream is the same 64 fragments (k-slice major, gate minor) every step and does not depend on h: a rolling
    // queue of WQ fragments stays in flight ACROSS the step boundary, so the loads of the next step's first slices overlap
    // this step's cell arithmetic and barriers (round-3 first version, loads issued 4 slices at a time inside the step: 17 us
    // per step; the stream itself needs ~3.5 us).
    constexpr int WQ = 8;                         // two k-slices x four gates
    half8 wq[WQ];
#pragma unroll
    for (int f = 0; f < WQ; ++f) wq[f] = wfrag[(size_t)f * 64];
    auto fast_tanh = [](float x) { return 2.f / (1.f + __expf(-2.f * x)) - 1.f; };
    for (int step = 0; step < tmax; ++step) {
        const bool active = step < T;
        const int t = rev ? T - 1 - step : step;
        // gate pre-activations of this step (x . W_ih^T + b, fp32): issued now, consumed after the MFMAs
        float4v gx[4][4];
        {
            const float* gp = gbase + ((long)b * Tfull + t) * gld + 128 * wave + 4 * kq;      // channel order [wave][gate][unit in wave]
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gx[g][q] = float4v{0.f, 0.f, 0.f, 0.f};
                    if (active) gx[g][q] = *reinterpret_cast<const float4v*>(gp + g * 32 + 8 * q);
                }
        }
        float16v acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
#pragma unroll 1
        for (int s2 = 0; s2 < 8; ++s2) {          // (not unrolled: hipcc would hoist the whole step's loads and spill)
            const half8* wnext = wfrag + (size_t)(((s2 + 1) & 7) * 8) * 64;      // wraps into the next step's stream
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = 2 * s2 + u;
                const half8 bh = *reinterpret_cast<const half8*>(&hbuf[0][n][s * 16 + kq * 8]);
                const half8 bl = *reinterpret_cast<const half8*>(&hbuf[1][n][s * 16 + kq * 8]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const half8 a = wq[u * 4 + g];
                    wq[u * 4 + g] = wnext[(size_t)(u * 4 + g) * 64];
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[g], 0, 0, 0);
                }
            }
        }
        __syncthreads();                          // every wave has read h(t-1)
        if (active) {
            half_t* orow = obase + ((long)b * Tfull + t) * out.ld + u0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                half4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const float zi = acc[0][i] + gx[0][q][e], zf = acc[1][i] + gx[1][q][e], zg = acc[2][i] + gx[2][q][e],
                                zo = acc[3][i] + gx[3][q][e];
                    const float i_ = 1.f / (1.f + __expf(-zi)), f_ = 1.f / (1.f + __expf(-zf)), o_ = 1.f / (1.f + __expf(-zo));
                    c[i] = f_ * c[i] + i_ * fast_tanh(zg);
                    const float h = o_ * fast_tanh(c[i]);
                    const half_t hi = (half_t)h;
                    hbuf[0][n][u0 + 8 * q + e] = hi;
                    hbuf[1][n][u0 + 8 * q + e] = (half_t)(h - (float)hi);
                    o4[e] = hi;
                }
                *reinterpret_cast<half4*>(orow + 8 * q) = o4;
            }
        }
        __syncthreads();                          // h(t) complete
    }
    if (b < B) {
        // ragged batch: steps behind the sample's own end hold zeros (what the next layer's masked producer would have written)
        for (int t = T; t < Tfull; ++t) {
            half_t* orow = obase + ((long)b * Tfull + t) * out.ld + u0;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<half4*>(orow + 8 * q) = half4{0, 0, 0, 0};
        }
    }
}

// 16-wave form (p[2] = 16): the step time of the 8-wave kernel is set by the bytes in flight, not by the stream or the MFMAs
// (8 waves x 8 KiB of fragment loads against ~1.75 us of L2 latency = 37 B/ns per CU -> 14 us for 512 KiB).  Here a block has 16
// waves (4 per SIMD, <= 128 VGPRs): wave w owns hidden units 16w .. 16w+15; its two accumulator tiles hold [gate i | gate f] and
// [gate g | gate o] of those units in rows 0-15 | 16-31, so the four gates of a (unit, sample) are still in one lane
// (registers 4q'+e and 4(q'+2)+e of the two tiles).  Twice the waves = twice the loads in flight, half the MFMAs and half the cell
// arithmetic per wave.  Fragment (= stream) order: [wave 16][k-slice 16][tile 2][lane][8].
__global__ __launch_bounds__(1024, 4) void lstm_mfma16_kernel(TView gates_f, TView gates_r, TView out, const half_t* __restrict__ whh,
                                                              int rev_single, int ndir, const int* __restrict__ tl) {
    __shared__ half_t hbuf[2][32][LSTM_LDH];      // [hi / lo][sample][hidden unit]
    __shared__ int s_tmax;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kq = lane >> 5;
    const int dir = blockIdx.y;
    const int rev = ndir == 2 ? dir : rev_single;
    const int B = gates_f.n, Tfull = gates_f.w, gld = gates_f.ld;
    const int b = blockIdx.x * 32 + n;
    const int T = b < B ? (tl != nullptr ? min(max(tl[b], 0), Tfull) : Tfull) : 0;
    if (threadIdx.x == 0) s_tmax = 0;
    for (int i = threadIdx.x; i < 2 * 32 * LSTM_LDH; i += blockDim.x) (&hbuf[0][0][0])[i] = (half_t)0.f;
    __syncthreads();
    if (wave == 0 && kq == 0) atomicMax(&s_tmax, T);
    __syncthreads();
    const int tmax = s_tmax;
    const half8* wfrag = reinterpret_cast<const half8*>(whh) + (size_t)dir * (16 * 16 * 2 * 64) + (size_t)wave * (16 * 2 * 64) + lane;
    const float* gbase = reinterpret_cast<const float*>((ndir == 2 && dir == 1) ? gates_r.ptr : gates_f.ptr);
    half_t* obase = reinterpret_cast<half_t*>(out.ptr) + (ndir == 2 ? dir * LSTM_H : 0);
    const int u0 = 16 * wave + 4 * kq;            // this lane's units: u0 + 8q' + e, q' = 0, 1
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.f;
#ifndef LSTM16_WQ
#define LSTM16_WQ 2       // fragments in flight: 4 spills 15 VGPRs at the 128-register budget of 4 waves per SIMD and is 10 % SLOWER (2.46 vs 2.22 ms per layer, tools/ab_lstm_wq.sh, round 5); 8 spills ~25 and is 12 % slower still
#endif
    constexpr int WQ = LSTM16_WQ, SPC = WQ / 2, NCH = 16 / SPC;      // fragments in flight = SPC k-slices x two tiles; chunks per step
    half8 wq[WQ];
#pragma unroll
    for (int f = 0; f < WQ; ++f) wq[f] = wfrag[(size_t)f * 64];
    auto fast_tanh = [](float x) { return 2.f / (1.f + __expf(-2.f * x)) - 1.f; };
    // gate pre-activations x . W_ih^T + b of the step (fp32, 16-byte pieces of 32 different rows per wave instruction: ~3 us of
    // exposed latency per step when they were loaded at the step's start — tools/ablate_lstm.sh).  They are loaded ONE STEP AHEAD,
    // into the registers the cell arithmetic has just finished reading: the loads fly during the rest of the cell phase, the
    // barrier and the next step's MFMAs.
    float4v gx[4][2];
    auto load_gx = [&](int step_) {
        const bool act = step_ < T;
        const int t_ = rev ? T - 1 - step_ : step_;
        const float* gp = gbase + ((long)b * Tfull + t_) * gld + 64 * wave + 4 * kq;          // channel order [wave][gate][unit in wave]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gx[g][q] = float4v{0.f, 0.f, 0.f, 0.f};
#ifndef LSTM_ABL_NO_GX
                if (act) gx[g][q] = *reinterpret_cast<const float4v*>(gp + g * 16 + 8 * q);
#endif
            }
    };
    load_gx(0);
    for (int step = 0; step < tmax; ++step) {
        const bool active = step < T;
        const int t = rev ? T - 1 - step : step;
        float16v acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
#pragma unroll 1
        for (int s4 = 0; s4 < NCH; ++s4) {
            const half8* wnext = wfrag + (size_t)(((s4 + 1) & (NCH - 1)) * WQ) * 64;
#pragma unroll
            for (int u = 0; u < SPC; ++u) {
                const int s = SPC * s4 + u;
                const half8 bh = *reinterpret_cast<const half8*>(&hbuf[0][n][s * 16 + kq * 8]);
                const half8 bl = *reinterpret_cast<const half8*>(&hbuf[1][n][s * 16 + kq * 8]);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const half8 a = wq[u * 2 + g];
#ifndef LSTM_ABL_NO_WLOAD
                    wq[u * 2 + g] = wnext[(size_t)(u * 2 + g) * 64];
#endif
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc[g], 0, 0, 0);
#ifndef LSTM_ABL_NO_LO
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[g], 0, 0, 0);
#endif
                }
            }
        }
        __syncthreads();                          // every wave has read h(t-1)
        // z = h . W_hh^T + gate pre-activations, then the next step's pre-activations are requested before any arithmetic
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][4 * q + e] += gx[0][q][e];
                acc[0][4 * q + e + 8] += gx[1][q][e];
                acc[1][4 * q + e] += gx[2][q][e];
                acc[1][4 * q + e + 8] += gx[3][q][e];
            }
        __builtin_amdgcn_sched_barrier(0);
        load_gx(step + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            half_t* orow = obase + ((long)b * Tfull + t) * out.ld + u0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                half4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const float zi = acc[0][i], zf = acc[0][i + 8], zg = acc[1][i], zo = acc[1][i + 8];
#ifdef LSTM_ABL_NO_CELL
                    c[i] += zi + zf; const float h = zg + zo + c[i];
#else
                    const float i_ = 1.f / (1.f + __expf(-zi)), f_ = 1.f / (1.f + __expf(-zf)), o_ = 1.f / (1.f + __expf(-zo));
                    c[i] = f_ * c[i] + i_ * fast_tanh(zg);
                    const float h = o_ * fast_tanh(c[i]);
#endif
                    const half_t hi = (half_t)h;
                    hbuf[0][n][u0 + 8 * q + e] = hi;
                    hbuf[1][n][u0 + 8 * q + e] = (half_t)(h - (float)hi);
                    o4[e] = hi;
                }
                *reinterpret_cast<half4*>(orow + 8 * q) = o4;
            }
        }
        __syncthreads();                          // h(t) complete
    }
    if (b < B) {
        for (int t = T; t < Tfull; ++t) {
            half_t* orow = obase + ((long)b * Tfull + t) * out.ld + u0;
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<half4*>(orow + 8 * q) = half4{0, 0, 0, 0};
        }
    }
}

// in0 = forward (or the only) direction's gate pre-activations fp32 [B,1,T,4H], in1 = the reverse direction's when ndir == 2;
// out = [B,1,T,ndir*H] fp16; whh = fragment-ordered W_hh^T of direction 0 then direction 1.
int launch_lstm_mfma(const TView& gf, const TView& gr, const TView& out, const half_t* whh, int rev_single, int ndir, int waves,
                     const int* tl, hipStream_t st) {
    if (gf.esize != 4 || gf.c != 4 * LSTM_H || out.esize != 2 || (out.ld & 3) || (gf.ld & 3) || out.c != ndir * LSTM_H) return VSE_E_INVAL;
    if (ndir == 2 && (gr.esize != 4 || gr.c != 4 * LSTM_H || gr.n != gf.n || gr.w != gf.w || gr.ld != gf.ld)) return VSE_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(out.ptr) & 7) || (reinterpret_cast<uintptr_t>(gf.ptr) & 15)) return VSE_E_INVAL;
    if (waves == 16)
        hipLaunchKernelGGL(lstm_mfma16_kernel, dim3((gf.n + 31) / 32, ndir), dim3(1024), 0, st, gf, gr, out, whh, rev_single, ndir, tl);
    else if (waves == 8)
        hipLaunchKernelGGL(lstm_mfma_kernel, dim3((gf.n + 31) / 32, ndir), dim3(512), 0, st, gf, gr, out, whh, rev_single, ndir, tl);
    else
        return VSE_E_INVAL;
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
