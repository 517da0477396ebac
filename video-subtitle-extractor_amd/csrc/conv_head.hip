// DB head on the LOW-RESOLUTION grid ("conv_head_up2_kernel", F_UP2HEAD).
//
// PP-OCRv4's server detector ends in   y = dot(relu(bn(conv3x3(concat[u, up2(x)]))), w) -> sigmoid   at full resolution,
// where x is a 64-channel map at HALF resolution (nearest x2 upsampled on the fly) and u a 1-channel full-resolution
// map.  Under nearest x2 upsampling the 3x3 window of an output pixel of parity (a, b) = (row & 1, col & 1) touches
// only 2 x 2 distinct pixels of x: low-res offsets {a-1, a} x {b-1, b}.  The compiler therefore folds the 3x3 taps into
// four 2x2-tap weight sets W_ab[r][s] (sums of the original taps, in fp64), and this kernel computes, for a tile of
// 8 x 32 LOW-RES pixels, the 4 parities x 64 couts from ONE staging of the low-res halo patch: 2.25x fewer MACs than
// the full-resolution 3x3 conv (exact in real arithmetic), no upsampled tensor, no 64-channel output tensor.
//
//   block   = 512 threads = 8 waves: wave -> 2 low-res rows (wave >> 1) x 32 couts (wave & 1), 4 parities:
//             acc[4][2] accumulator tiles (128 VGPRs)
//   LDS     = x halo patch, both 32-channel chunks (10 x 34 pixels, staged once) 48 KiB + 4-stage weight ring
//             (stage = 4 taps x 64 couts x 32 channels = 16 KiB) 64 KiB + u weights 4 KiB + u tile 2.4 KiB
//             + cross-wave partial dots 8 KiB + epilogue constants
//   K loop  = 8 steps (2 chunks x 4 parities, fully unrolled: the parity selects the accumulator set), 16 MFMAs per
//             wave between barriers; the 6 activation fragments a step needs serve its 8 (row, tap) pairs
//   u       = one extra K = 16 MFMA slice per parity: the 9 full-resolution taps of the 1-channel map are gathered
//             from the LDS u tile (ds_read_u16) into an activation fragment
//   epilogue= bias -> activation -> dot with the 1x1 projection over this wave's 32 couts -> cross-half shuffle ->
//             LDS exchange between the two cout halves -> activation (sigmoid) -> one 8-byte store per lane and row
//             parity (columns 2j, 2j+1)
// weights stream (compiler.py head_up2_weights): [chunk][parity a*2+b][tap r*2+s][64][32] fp16, then [64][32] for u
// (k = 3*dy + dx < 9, rest zero).
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

#define HT_ROWS 8
#define HT_COLS 32
#define HPW (HT_COLS + 2)
#define HPH (HT_ROWS + 2)
#define HP (HPW * HPH)               // 340 patch pixels
#define HPPIX 384                    // staged (whole wave instructions: 8 waves x 3 x 16 pixels)
#define UTW 68                       // u tile row pitch (66 used)
#define UTH 18

__global__ __launch_bounds__(512) void conv_head_up2_kernel(const ConvParams p) {
    constexpr int PATCHC_HALFS = HPPIX * 32;                 // one 32-channel chunk of the patch
    constexpr int WSTAGE_HALFS = 4 * 64 * 32;
    constexpr int NSTEP = 8;
    __shared__ __attribute__((aligned(16))) half_t lds[2 * PATCHC_HALFS + 4 * WSTAGE_HALFS + 64 * 32 + UTH * UTW + 2 * 4096 + 2 * 128];
    half_t* const patch0 = lds;
    half_t* const ring0 = lds + 2 * PATCHC_HALFS;
    half_t* const uw0 = ring0 + 4 * WSTAGE_HALFS;
    half_t* const ut0 = uw0 + 64 * 32;
    float* const part0 = reinterpret_cast<float*>(ut0 + UTH * UTW);      // [8 waves][4 parities][2 rows][32 px]
    float* const sbias = part0 + 2048;
    float* const sdotw = sbias + 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave >> 1, wco = wave & 1;

    // XCD-aware bijective block order (see conv_mfma.hip)
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * HT_ROWS, ox0 = tx * HT_COLS;        // low-res origin of the tile
    const int Hl = p.in2_hs, Wl = p.in2_ws;                  // low-res map; the output is 2Hl x 2Wl

    // u tile: full-res rows 2*oy0-1 .. 2*oy0+16, cols 2*ox0-1 .. 2*ox0+64 of channel 0, zero outside the map.  The (plain)
    // loads are issued FIRST, all three per thread back to back, and land in LDS after the one vmcnt(0) below: a load ->
    // ds_write loop after the DMAs made hipcc wait for the whole DMA queue and then for two more memory round trips.
    half_t uval[3];
    {
        const half_t* ub = p.in + img * (long)(2 * Hl) * (2 * Wl) * p.in_ld;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + 512 * k;
            const int uy = idx / 66, ux = idx - uy * 66;
            const int fy = 2 * oy0 - 1 + uy, fx2 = 2 * ox0 - 1 + ux;
            uval[k] = (half_t)0.f;
            if (idx < UTH * 66 && fy >= 0 && fy < 2 * Hl && fx2 >= 0 && fx2 < 2 * Wl) uval[k] = ub[((long)fy * (2 * Wl) + fx2) * p.in_ld];
        }
    }
    // ---- prologue: everything but the weight stream is staged once --------------------------------------------------
    const int kv = (lane & 3) ^ ((lane >> 4) & 3);           // logical k-vector this lane fetches (source-side swizzle)
    conv_stage_consts<true>(sbias, p.bias, p.zero, 0, 64, p.Np, wave, lane);            // wave 0
    conv_stage_consts<true>(sdotw, p.dotw, p.zero, 0, 64, p.Np, wave - 4, lane);        // wave 4
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = 16 * (wave + 8 * j) + (lane >> 2);
            const int py = q / HPW, px = q - py * HPW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = (q < HP) && (iy >= 0) && (iy < Hl) && (ix >= 0) && (ix < Wl);
            const half_t* src = ok ? p.in2 + ((img * Hl + iy) * Wl + ix) * (long)p.in2_ld + c * 32 + kv * 8 : p.zero;
            glds16_asm(src, patch0 + c * PATCHC_HALFS + (wave + 8 * j) * 16 * 32);
        }
    const half_t* const wu = p.w + (long)NSTEP * WSTAGE_HALFS;
    if (wave < 4) glds16_asm(wu + (wave * 16 + (lane >> 2)) * 32 + kv * 8, uw0 + wave * 16 * 32);
    // weight ring: thread -> cout row (tid>>2)&63; waves 0-3 fetch taps 0 and 2 of a stage, waves 4-7 taps 1 and 3
    const half_t* wptr = p.w + ((long)(wave >> 2) * 64 + ((tid >> 2) & 63)) * 32 + kv * 8;
    auto issue_w = [&](int s) __attribute__((always_inline)) {
        half_t* st = ring0 + (s & 3) * WSTAGE_HALFS;
        glds16_asm(wptr, st + (wave >> 2) * 64 * 32 + (wave & 3) * 16 * 32);
        glds16_asm(wptr + 2 * 64 * 32, st + (2 + (wave >> 2)) * 64 * 32 + (wave & 3) * 16 * 32);
        wptr += WSTAGE_HALFS;
    };
    issue_w(0);
    issue_w(1);
    issue_w(2);
    float16v acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

    const int fx = lane & 31, fj = lane >> 5;
    const int wr = wco * 32 + conv_wrow(fx);                 // weight row (cout) this lane supplies
    const unsigned woffb = (unsigned)(wr * 64 + ((fj ^ ((wr >> 2) & 3)) << 4));
    const char* const ring_b = reinterpret_cast<const char*>(ring0);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = tid + 512 * k;
        const int uy = idx / 66, ux = idx - uy * 66;
        if (idx < UTH * 66) ut0[uy * UTW + ux] = uval[k];
    }
    __syncthreads();

    // ---- K loop -----------------------------------------------------------------------------------------------------
    auto step = [&](auto c_c, auto par_c) __attribute__((always_inline)) {
        constexpr int C = decltype(c_c)::value, PAR = decltype(par_c)::value;
        constexpr int S = C * 4 + PAR, A = PAR >> 1, B = PAR & 1;
        if constexpr (S > 0) {
            __builtin_amdgcn_sched_barrier(0);     // keep the barrier behind the previous step's fragment reads
            // stage S has landed: only the stages issued after it (S+1, S+2 when they exist) may be outstanding
            if constexpr (S + 2 < NSTEP) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (S + 1 < NSTEP) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if constexpr (S + 3 < NSTEP) issue_w(S + 3);
        const char* const pb = reinterpret_cast<const char*>(patch0 + C * PATCHC_HALFS);
        const unsigned wsb = (unsigned)(S & 3) * (WSTAGE_HALFS * 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // activation fragments: patch rows 2*wrow + A + {0,1,2}, cols fx + B + {0,1}
            half8 xf[3][2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const unsigned q = (unsigned)((2 * wrow + A + rr) * HPW + fx + B + s2);
                    xf[rr][s2] = *reinterpret_cast<const half8*>(pb + ((q << 6) + (((2 * ks + fj) ^ ((q >> 2) & 3)) << 4)));
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const half8 wf = *reinterpret_cast<const half8*>(ring_b + wsb + (r * 2 + s2) * (64 * 64) + (woffb ^ (ks << 5)));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[PAR][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf[i + r][s2], acc[PAR][i], 0, 0, 0);
                }
        }
    };
    step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});

    // ---- the 1-channel full-resolution source: one K = 16 slice per parity ---------------------------------------------
    {
        const half8 wfu = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(uw0) + woffb);   // k 0..15 of row wr
#pragma unroll
        for (int par = 0; par < 4; ++par) {
            const int a = par >> 1, b = par & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // lane (fx, fj) supplies k = 8*fj + e: tap (dy, dx) = (k / 3, k % 3) of output pixel (2*row + a, 2*fx + b)
                const int uy0 = 2 * (2 * wrow + i) + a, ux0 = 2 * fx + b;      // tile coords of tap (0, 0)
                half8 xu = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (fj == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xu[e] = ut0[(uy0 + e / 3) * UTW + ux0 + e % 3];
                } else {
                    xu[0] = ut0[(uy0 + 2) * UTW + ux0 + 2];                        // k = 8: tap (2, 2)
                }
                acc[par][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfu, xu, acc[par][i], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    {
        float dbias[16], dw[16];
        conv_epilogue_consts(sbias, wco * 32, lane, dbias);
        conv_epilogue_consts(sdotw, wco * 32, lane, dw);
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float part = conv_epilogue_dot(p, acc[par][i], dbias, dw);
                part += __shfl_xor(part, 32);
                if (fj == 0) part0[((wave * 4 + par) * 2 + i) * 32 + fx] = part;
            }
    }
    __syncthreads();
    if (wco == 0) {
        // lanes 0-31: row parity a = 0, lanes 32-63: a = 1; both column parities of a pixel -> one 8-byte store
        const int a = fj;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + 2 * wrow + i, ox = ox0 + fx;
            if (oy >= Hl || ox >= Wl) continue;
            float z[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int par = a * 2 + b;
                const float s0 = part0[((wave * 4 + par) * 2 + i) * 32 + fx] + part0[(((wave + 1) * 4 + par) * 2 + i) * 32 + fx];
                z[b] = vse_act(s0 + p.dotb, p.dotact, 0.f, 0.f);
            }
            const long m = ((img * (2 * Hl) + 2 * oy + a) * (long)(2 * Wl) + 2 * ox) * p.dot_ld;
            if (p.dot_f32) {
                float* o = reinterpret_cast<float*>(p.dot_out) + m;
                if (p.dot_ld == 1) *reinterpret_cast<float2*>(o) = make_float2(z[0], z[1]);
                else { o[0] = z[0]; o[p.dot_ld] = z[1]; }
            } else {
                half_t* o = reinterpret_cast<half_t*>(p.dot_out) + m;
                o[0] = (half_t)z[0]; o[p.dot_ld] = (half_t)z[1];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Resident-weight form (end of round 3).  conv_head_up2_kernel streams 128 KiB of folded weights + 48 KiB of patch through
// LDS-DMA per 256-pixel tile — 5.7 GB per 64 frames at the ~3 TB/s such streams reach on this chip = the kernel's 1.9 ms;
// its matrix pipe is busy a quarter of the time.  Here a PERSISTENT block (one per CU) owns ONE output row parity a: the
// 64 KiB of weights of parities (a, 0), (a, 1) are loaded ONCE per block and stay in LDS; the block walks 16 x 32 low-res
// tiles, and the only stream left is the halo patch (2 x 39 KiB per tile, the next tile's chunks prefetched behind the
// current tile's compute): 2.25x fewer staged bytes per output, three barriers per tile instead of eight, 14 fragment reads
// per 16 MFMAs instead of 20.
//   wave   = 4 low-res rows (wave >> 1) x 32 couts (wave & 1) x 2 column parities: acc[2][4] (128 VGPRs)
//   LDS    = weights 64 KiB + two patch chunks 80 KiB + u weights 4 KiB + u tile 4.5 KiB + dot exchange 4 KiB + constants
//   order  = per parity: chunk 0 (ks, r, s), chunk 1, u slice — conv_head_up2_kernel's, so the bits are identical
//   stream = chunk 0 of tile t+1 is issued during chunk 1 of tile t (SIMD partners at different points), chunk 1 of t+1
//            behind the barrier that ends tile t's reads; counted waits: vmcnt(5) (chunk 0 landed, chunk 1 may fly) at the
//            end of a tile, vmcnt(0) in front of chunk 1.  The u tile's plain loads are issued in front of the chunk-0 DMAs
//            and consumed behind that barrier, so the compiler's own wait for them drains nothing that is not needed anyway.
#ifdef VSE_TRACE
#include <stdio.h>
#include <vector>
#define HTR(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define HACC(acc_, a_, b_) acc_ += (b_) - (a_)
#else
#define HTR(v) do { } while (0)
#define HACC(acc_, a_, b_) do { } while (0)
#endif
#define RT_ROWS 16
#define RPH (RT_ROWS + 2)
#define RHP (HPW * RPH)              // 612 patch pixels
#define RPJ 5                        // patch DMA instructions per wave and chunk (40 x 16 pixels)
#define RUTH (2 * RT_ROWS + 2)       // u tile rows
#define RUPT 5                       // u values per thread (34 x 66 <= 5 x 512)

__global__ __launch_bounds__(512, 2) void conv_head_up2r_kernel(const ConvParams p) {
    constexpr int WST = 4 * 64 * 32;                         // halfs per (chunk, parity) weight stage: 4 taps x 64 couts x 32 ch
    constexpr int PCH = 8 * RPJ * 16 * 32;                   // halfs per patch chunk buffer (40 KiB)
    __shared__ __attribute__((aligned(16))) half_t lds[4 * WST + 2 * PCH + 64 * 32 + RUTH * UTW + 2 * 1024 + 2 * 128];
    half_t* const wres = lds;                                // [chunk][b] stages
    half_t* const patch0 = lds + 4 * WST;
    half_t* const uw0 = patch0 + 2 * PCH;
    half_t* const ut0 = uw0 + 64 * 32;
    float* const part0 = reinterpret_cast<float*>(ut0 + RUTH * UTW);     // [4 wrow][2 b][4 rows][32 px]: the upper cout half's dots
    float* const sbias = part0 + 1024;
    float* const sdotw = sbias + 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave >> 1, wco = wave & 1;
    const int fx = lane & 31, fj = lane >> 5;
    const int Hl = p.in2_hs, Wl = p.in2_ws;

    // ---- this block's work: row parity a, tiles x0 + slot, x0 + slot + na, ... of its XCD's contiguous tile range ----------
    const unsigned G = gridDim.x, bid = blockIdx.x, xcd = bid & 7, bslot = bid >> 3;
    const int a = (int)(bslot & 1);
    const unsigned T = p.ntiles, q8 = T >> 3, r8 = T & 7;
    const unsigned x0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned tend = x0 + q8 + (xcd < r8 ? 1u : 0u);
    const unsigned na = (G >> 3) >> 1;                       // blocks per XCD and parity (the launcher makes G a multiple of 16)
    unsigned t = x0 + (bslot >> 1);
    if (t >= tend) return;
    int oy0, ox0;
    long img;
    auto decode = [&](unsigned tt) {
        const unsigned tx = tt % (unsigned)p.tiles_w, r = tt / (unsigned)p.tiles_w;
        ox0 = (int)tx * HT_COLS;
        oy0 = (int)(r % (unsigned)p.tiles_h) * RT_ROWS;
        img = (long)(r / (unsigned)p.tiles_h);
    };
    decode(t);

    const int kv = (lane & 3) ^ ((lane >> 4) & 3);           // logical k-vector this lane fetches (source-side swizzle)
    auto issue_patch = [&](int c, bool live) __attribute__((always_inline)) {
        half_t* dstb = patch0 + c * PCH;
#pragma unroll
        for (int j = 0; j < RPJ; ++j) {
            const int q = 16 * (wave + 8 * j) + (lane >> 2);
            const int py = q / HPW, px = q - py * HPW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = live && (q < RHP) && (iy >= 0) && (iy < Hl) && (ix >= 0) && (ix < Wl);
            const half_t* src = ok ? p.in2 + ((img * Hl + iy) * Wl + ix) * (long)p.in2_ld + c * 32 + kv * 8 : p.zero;
            glds16_asm(src, dstb + (wave + 8 * j) * 16 * 32);
        }
    };
    half_t uval[RUPT];
    unsigned uok = 0;                                        // bit k: uval[k] lies inside the map
    auto load_u = [&]() __attribute__((always_inline)) {
        const half_t* ub = p.in + img * (long)(2 * Hl) * (2 * Wl) * p.in_ld;
#pragma unroll
        for (int k = 0; k < RUPT; ++k) {
            const int idx = tid + 512 * k;
            const int uy = idx / 66, ux = idx - uy * 66;
            const int fy = 2 * oy0 - 1 + uy, fx2 = 2 * ox0 - 1 + ux;
            // unconditional loads from clamped addresses, zero selected afterwards: a load under a condition makes hipcc branch around
            // it and wait vmcnt(0) per element — five dependent round trips, each draining the DMA queue
            const bool ok = idx < RUTH * 66 && fy >= 0 && fy < 2 * Hl && fx2 >= 0 && fx2 < 2 * Wl;
            const int cy = min(max(fy, 0), 2 * Hl - 1), cx = min(max(fx2, 0), 2 * Wl - 1);
            uval[k] = ub[((long)cy * (2 * Wl) + cx) * p.in_ld];      // raw: the select waits for the load, so it happens in store_u
            uok = ok ? (uok | (1u << k)) : (uok & ~(1u << k));
        }
        asm volatile("" ::: "memory");
    };
    auto store_u = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < RUPT; ++k) {
            const int idx = tid + 512 * k;
            const int uy = idx / 66, ux = idx - uy * 66;
            if (idx < RUTH * 66) ut0[uy * UTW + ux] = ((uok >> k) & 1u) ? uval[k] : (half_t)0.f;
        }
    };

    // ---- prologue: constants, the block's 64 KiB of weights, the first tile ----------------------------------------------
    load_u();
    conv_stage_consts<true>(sbias, p.bias, p.zero, 0, 64, p.Np, wave, lane);            // wave 0
    conv_stage_consts<true>(sdotw, p.dotw, p.zero, 0, 64, p.Np, wave - 4, lane);        // wave 4
    {
        const half_t* wl = p.w + ((long)(wave >> 2) * 64 + ((tid >> 2) & 63)) * 32 + kv * 8;
#pragma unroll
        for (int st = 0; st < 4; ++st) {                      // st = chunk * 2 + b  <-  stream stage chunk * 4 + a * 2 + b
            const half_t* src = wl + (long)((st >> 1) * 4 + a * 2 + (st & 1)) * WST;
            half_t* dst = wres + st * WST;
            glds16_asm(src, dst + (wave >> 2) * 64 * 32 + (wave & 3) * 16 * 32);
            glds16_asm(src + 2 * 64 * 32, dst + (2 + (wave >> 2)) * 64 * 32 + (wave & 3) * 16 * 32);
        }
        const half_t* const wu = p.w + 8L * WST;
        if (wave < 4) glds16_asm(wu + (wave * 16 + (lane >> 2)) * 32 + kv * 8, uw0 + wave * 16 * 32);
    }
    issue_patch(0, true);
    issue_patch(1, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_u();
    __syncthreads();

    const int wr = wco * 32 + conv_wrow(fx);                 // weight row (cout) this lane supplies
    const unsigned woffb = (unsigned)(wr * 64 + ((fj ^ ((wr >> 2) & 3)) << 4));
    const char* const wres_b = reinterpret_cast<const char*>(wres);
    const half8 wfu = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(uw0) + woffb);   // k 0..15 of row wr
    const bool early = wave < 4;                             // SIMD partners (w, w + 4) issue the next tile's stream at different points

#ifdef VSE_TRACE
    unsigned long long tr_c0 = 0, tr_b1 = 0, tr_c1 = 0, tr_u = 0, tr_dot = 0, tr_b2 = 0, tr_tail = 0, tr_b3 = 0, tr_n = 0;
    const unsigned long long tr_begin = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        HTR(t0);
        float16v acc[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][i][r] = 0.f;
        // one (chunk, column parity): 32 MFMAs, 28 fragment reads
        auto compute = [&](int c, auto b_c) __attribute__((always_inline)) {
            constexpr int B = decltype(b_c)::value;
            const char* const pb = reinterpret_cast<const char*>(patch0 + c * PCH);
            const unsigned wsb = (unsigned)(c * 2 + B) * (WST * 2);
            unsigned q0 = (unsigned)((4 * wrow + a) * HPW + fx + B);      // recomputed per call: hoisted out of the tile loop, the 80 fragment
            asm volatile("" : "+v"(q0));                                 // addresses of a tile would live (spilled) across everything
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 xf[5][2];
#pragma unroll
                for (int rr = 0; rr < 5; ++rr)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const unsigned q = q0 + (unsigned)(rr * HPW + s2);
                        xf[rr][s2] = *reinterpret_cast<const half8*>(pb + ((q << 6) + (((2 * ks + fj) ^ ((q >> 2) & 3)) << 4)));
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const half8 wf = *reinterpret_cast<const half8*>(wres_b + wsb + (r * 2 + s2) * (64 * 64) + (woffb ^ (ks << 5)));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[B][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf[i + r][s2], acc[B][i], 0, 0, 0);
                    }
            }
        };
        compute(0, std::integral_constant<int, 0>{});
        compute(0, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        HTR(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // own part of chunk 1 landed (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();                       // chunk 1 visible; nobody reads chunk 0 any more
        asm volatile("" ::: "memory");
        HTR(t2);

        // the next tile: its geometry replaces this tile's behind the epilogue
        const int c_oy0 = oy0, c_ox0 = ox0;
        const long c_img = img;
        const unsigned tn = t + na;
        const bool have_next = tn < tend;
        if (have_next) decode(tn);
        if (early && have_next) { load_u(); issue_patch(0, true); }
        __builtin_amdgcn_sched_barrier(0);
        compute(1, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!early && have_next) { load_u(); issue_patch(0, true); }
        __builtin_amdgcn_sched_barrier(0);
        compute(1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        HTR(t3);

        // ---- the 1-channel full-resolution source: one K = 16 slice per column parity and row ----------------------------
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int ub0 = (2 * (4 * wrow + i) + a) * UTW + 2 * fx + b;          // u-tile index of tap (0, 0)
                asm volatile("" : "+v"(ub0));
                half8 xu = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (fj == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xu[e] = ut0[ub0 + (e / 3) * UTW + e % 3];
                } else {
                    xu[0] = ut0[ub0 + 2 * UTW + 2];                                // k = 8: tap (2, 2)
                }
                acc[b][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfu, xu, acc[b][i], 0, 0, 0);
            }

        __builtin_amdgcn_sched_barrier(0);
        HTR(t4);
        // ---- epilogue: dots over this wave's 32 couts; the upper cout half hands its dots to the lower one ----------------
        float part[2][4];
        float dbias[16], dw[16];                            // read back per tile: held across the K loop they cost 32 VGPRs (spills)
        conv_epilogue_consts(sbias, wco * 32, lane, dbias);
        conv_epilogue_consts(sdotw, wco * 32, lane, dw);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = conv_epilogue_dot(p, acc[b][i], dbias, dw);
                v += __shfl_xor(v, 32);
                part[b][i] = v;
                if (wco == 1 && fj == 0) part0[((wrow * 2 + b) * 4 + i) * 32 + fx] = v;
            }
        __builtin_amdgcn_sched_barrier(0);
        HTR(t5);
        __syncthreads();                                    // dots visible; chunk 1 and the u tile are free
        HTR(t6);
        if (have_next) {
            store_u();
            issue_patch(1, true);
        }
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // own part of the next tile's chunk 0 landed; its chunk 1 (the 5 youngest) may fly
        if (wco == 0) {
            // lanes 0-31: rows 0, 1 of the wave's four, lanes 32-63: rows 2, 3; both column parities of a pixel -> one 8-byte store
            // (no array indexed by fj: hipcc moves such an array to scratch memory, and scratch traffic sits in the vmcnt queue)
            const float own[2][2] = {{fj == 0 ? part[0][0] : part[0][2], fj == 0 ? part[0][1] : part[0][3]},
                                     {fj == 0 ? part[1][0] : part[1][2], fj == 0 ? part[1][1] : part[1][3]}};
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * fj + ii;
                const int oy = c_oy0 + 4 * wrow + i, ox = c_ox0 + fx;
                if (oy < Hl && ox < Wl) {
                    float z[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        z[b] = vse_act(own[b][ii] + part0[((wrow * 2 + b) * 4 + i) * 32 + fx] + p.dotb, p.dotact, 0.f, 0.f);
                    const long m = ((c_img * (2 * Hl) + 2 * oy + a) * (long)(2 * Wl) + 2 * ox) * p.dot_ld;
                    if (p.dot_f32) {
                        float* o = reinterpret_cast<float*>(p.dot_out) + m;
                        if (p.dot_ld == 1) *reinterpret_cast<float2*>(o) = make_float2(z[0], z[1]);
                        else { o[0] = z[0]; o[p.dot_ld] = z[1]; }
                    } else {
                        half_t* o = reinterpret_cast<half_t*>(p.dot_out) + m;
                        o[0] = (half_t)z[0]; o[p.dot_ld] = (half_t)z[1];
                    }
                }
            }
        }
        HTR(t7);
        HACC(tr_c0, t0, t1); HACC(tr_b1, t1, t2); HACC(tr_c1, t2, t3); HACC(tr_u, t3, t4); HACC(tr_dot, t4, t5); HACC(tr_b2, t5, t6); HACC(tr_tail, t6, t7);
#ifdef VSE_TRACE
        ++tr_n;
#endif
        if (!have_next) break;
        t = tn;
        __syncthreads();                                    // the next tile's chunk 0 and u tile visible; the dot exchange is free
        HTR(t8);
        HACC(tr_b3, t7, t8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef VSE_TRACE
    if (lane == 0 && p.trace && (wave == 0 || wave == 4)) {
        unsigned long long* o = p.trace + ((unsigned long long)blockIdx.x * 2 + (wave >> 2)) * 10;
        o[0] = __builtin_amdgcn_s_memtime() - tr_begin; o[1] = tr_c0; o[2] = tr_b1; o[3] = tr_c1; o[4] = tr_u; o[5] = tr_dot; o[6] = tr_b2; o[7] = tr_tail; o[8] = tr_b3; o[9] = tr_n;
    }
#endif
}

int launch_conv_head_up2(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    // u = in0: [n, 2Hl, 2Wl, 8-channel padded, 1 real]; x = in2: [n, Hl, Wl, 64], upsampled by 2
    if (!(p.flags & F_DOT1) || !(p.flags & F_SRC2) || p.in2_shift != 1 || p.inshift != 0) return VSE_E_INVAL;
    if (p.kh != 3 || p.kw != 3 || p.ph != 1 || p.pw != 1 || p.sh != 1 || p.sw != 1) return VSE_E_INVAL;
    if (p.cinp != 72 || p.nv0 != 1 || p.Np > 64 || (p.flags & (F_RES | F_PIXSHUF))) return VSE_E_INVAL;
    if (p.H != 2 * p.in2_hs || p.W != 2 * p.in2_ws || !p.dotw || !p.dot_out || !p.zero) return VSE_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(p.dot_out) & 7) || ((2 * p.in2_ws * p.dot_ld) & 1)) return VSE_E_INVAL;
    p.tiles_h = (p.in2_hs + HT_ROWS - 1) / HT_ROWS;
    p.tiles_w = (p.in2_ws + HT_COLS - 1) / HT_COLS;
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    static const int resident = [] { const char* e = getenv("VSE_HEAD_RESIDENT"); return e && e[0] ? atoi(e) : 1; }();
    if (resident) {
        // resident-weight form: 16 x 32 tiles, persistent blocks, grid a multiple of 16 (8 XCDs x 2 row parities)
        p.tiles_h = (p.in2_hs + RT_ROWS - 1) / RT_ROWS;
        const unsigned long long tiles = (unsigned long long)n_img * p.tiles_h * p.tiles_w;
        if (tiles == 0 || tiles > 0x7fffffffull) return VSE_E_INVAL;
        p.ntiles = (unsigned)tiles;
        const int cus = vse_cu_count() ? vse_cu_count() : 256;          // (per device: common.h)
        unsigned long long want = 2 * ((tiles + 7) / 8) * 8;                 // two blocks (row parities) per tile slot, whole XCD rounds
        if (want > (unsigned long long)cus) want = (unsigned long long)cus;
        const unsigned grid = (unsigned)(want < 16 ? 16 : want / 16 * 16);
#ifdef VSE_TRACE
        static unsigned long long* trace_dev = nullptr;
        if (!trace_dev) (void)hipMalloc(&trace_dev, 256 * 20 * sizeof(unsigned long long));
        (void)hipMemsetAsync(trace_dev, 0, 256 * 20 * sizeof(unsigned long long), st);
        p.trace = grid <= 256 ? trace_dev : nullptr;
#endif
        hipLaunchKernelGGL(conv_head_up2r_kernel, dim3(grid), dim3(512), 0, st, p);
#ifdef VSE_TRACE
        if (p.trace) {
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> h((size_t)grid * 20);
            (void)hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
            for (int g = 0; g < 2; ++g) {
                double d[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                size_t nb = 0;
                for (size_t b = 0; b < grid; ++b) {
                    const unsigned long long* t = &h[(b * 2 + g) * 10];
                    if (!t[9]) continue;
                    for (int i = 0; i < 10; ++i) d[i] += (double)t[i];
                    ++nb;
                }
                if (nb) fprintf(stderr, "[head trace] grid %u wave %d: per block total %.0f ticks over %.1f tiles; per tile: chunk 0 %.0f, wait+barrier %.0f, chunk 1 (+ next chunk 0 issue) %.0f, "
                                "u slice %.0f, dots %.0f, barrier %.0f, u store + chunk 1 issue + wait + stores %.0f, barrier %.0f\n", grid, 4 * g, d[0] / nb, d[9] / nb, d[1] / d[9],
                                d[2] / d[9], d[3] / d[9], d[4] / d[9], d[5] / d[9], d[6] / d[9], d[7] / d[9], d[8] / d[9]);
            }
        }
#endif
        return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
    }
    hipLaunchKernelGGL(conv_head_up2_kernel, dim3((unsigned)blocks), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
