// Implicit-GEMM convolution, scalar-addressed variant ("conv_gemm_kernel"), for layers whose K tiles never straddle a
// filter tap (cinp % BK == 0) and whose input is read at its own resolution.  Same math and epilogue as
// conv_mfma_kernel (conv_mfma.hip) — results are bit-identical — but built around what bounds an implicit GEMM on
// gfx950: the L2 -> LDS fill path (~56 B/clk/CU) and the instruction stream next to the MFMAs.
//
//   * every LDS-DMA is `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`: the per-lane part of the address
//     (row of the tile, 16-byte k-vector) is a 32-bit VGPR computed ONCE per block; the per-K-step part (filter tap
//     offset + channel offset, or the weight tile offset) is one SGPR updated by SALU.  No 64-bit vector address
//     arithmetic, no divergent branches in the loop.
//   * zero padding / M tail / cout tail come from the buffer descriptor's bounds check: an invalid lane's offset gets
//     bit 31 set, the load is out of range and the DMA writes zeros (no zero page, no select on pointers).  The
//     validity of (row, tap) is one bit of a per-row tap mask: v_bfe_u32 + v_lshl_or_b32 per load.
//   * BK = 64 configurations fetch whole 128-byte lines per row (one wave instruction = 8 rows x 128 B) and tiles of
//     256 pixels x 128/256 couts raise the flops per fetched byte over the 128 x 128 x 32 tile of conv_mfma_kernel.
//   * the ring is unrolled by stage, so every ds_read_b128 is `per-thread VGPR + immediate`.
//
// Eligibility and the tile configuration are decided in launch_conv_gemm(); everything else stays on conv_mfma_kernel.

#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

#ifndef VSE_GEMM_NT_A
#define VSE_GEMM_NT_A 0   // 1: non-temporal hint (aux = 2) on the ACTIVATION stream of the implicit GEMM (read once per launch by one block,
                          // while the weight tiles are re-read by every block and should keep the L2); A/B: tools/ab.sh conv_gemm VSE_GEMM_NT_A
#endif
#ifndef VSE_GEMM_ASM
#define VSE_GEMM_ASM 0    // 1: LDS-DMAs as asm statements (counted lgkmcnt waits survive; measured 1-3 % SLOWER here: the kernel is bound by the DMA stream, the asm form adds issue slots); 0: builtins.  A/B on one box: tools/ab.sh
#endif
#ifndef VSE_ABLATE
#define VSE_ABLATE 0      // 1: no s_barrier  2: no fragment ds_reads  3: no DMA in the loop  4: no MFMA  5: no epilogue   (timing experiments only)
#endif

// BM x BN block tile (pixels x couts), WM x WN waves, BKT-deep K tiles in an ST-stage LDS ring.
// MASK = 0: 1x1, no padding, K % 64 == 0 -> every (row, k) of a valid row is a real element, no tap mask.
// MASK = 1: up to 31 taps, one validity bit per tap and row.
// blocks per CU the LDS ring allows -> waves per SIMD the register allocation has to leave room for
constexpr int gemm_waves_per_simd(int bm, int bn, int bkt, int st, int nw) {
    const int rpi = 64 / (bkt / 8);
    const int bnr = (bn + rpi * nw - 1) / (rpi * nw) * (rpi * nw);
    const int blocks = (160 * 1024) / (st * (bm + bnr) * bkt * 2 + bn * 4);
    const int w = blocks * nw / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <int BM, int BN, int WM, int WN, int BKT, int ST, int MASK>
__global__ __launch_bounds__(64 * WM * WN, gemm_waves_per_simd(BM, BN, BKT, ST, WM * WN))
void conv_gemm_kernel(const ConvParams p) {
    constexpr int NW = WM * WN;                 // waves per block
    constexpr int KV = BKT / 8;                 // 16-byte k-vectors per LDS row (4 or 8)
    constexpr int RPI = 64 / KV;                // tile rows one wave instruction (1 KiB) covers
    constexpr int ROWB = BKT * 2;               // bytes per LDS row
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int KS = BKT / 16;                // MFMA k sub-steps per tile
    constexpr int BNR = (BN + RPI * NW - 1) / (RPI * NW) * (RPI * NW);   // weight rows staged (whole wave instructions)
    constexpr int NA = BM / (RPI * NW);         // LDS-DMA instructions per wave per stage, activations
    constexpr int NB = BNR / (RPI * NW);        // ... weights
    constexpr int LPT = NA + NB;
    constexpr int STAGE_HALFS = (BM + BNR) * BKT;
    static_assert(BKT == 32 || BKT == 64, "BK");
    static_assert(ST >= 2 && ST <= 5 && (ST - 2) * LPT <= 16, "ring depth");
    static_assert(TM >= 1 && TN >= 1 && NA >= 1 && NB >= 1 && BM % (RPI * NW) == 0 && BNR % (RPI * NW) == 0, "tile shape");
    static_assert(ST * STAGE_HALFS * 2 + BN * 4 <= 160 * 1024, "LDS");

    __shared__ __attribute__((aligned(16))) half_t lds[ST * STAGE_HALFS + BN * 2];   // ring + BN floats of bias
    float* const sbias = reinterpret_cast<float*>(lds + ST * STAGE_HALFS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order, identical to conv_mfma_kernel
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const unsigned mtile = logical / p.ntn, ntile = logical - mtile * p.ntn;
    // F_IMGW (per-image weights): M tiles are aligned to images, so that one tile has one weight matrix
    long m0 = (long)mtile * BM;
    long mend = p.M;                                 // rows at or beyond it do not exist
    const half_t* wsrc = p.w;
    if (p.flags & F_IMGW) {
        const unsigned img = mtile / (unsigned)p.tiles_img, ti = mtile - img * (unsigned)p.tiles_img;
        m0 = (long)img * p.hw_img + (long)ti * BM;
        mend = (long)(img + 1) * p.hw_img;
        wsrc += (long)img * p.wimg_stride;
    }
    const int n0 = ntile * BN;

    // LDS image: row-major [row][BKT halfs], 16-byte slots XOR-swizzled by the row so that the MFMA fragment reads
    // (32 consecutive rows, one logical slot) are bank-conflict free: BK=32: slot ^ (row>>2 & 3); BK=64: slot ^ (row>>1 & 7)
    auto swz = [](int r) { return BKT == 32 ? (r >> 2) & 3 : (r >> 1) & 7; };

    // ---- block-uniform descriptors ------------------------------------------------------------------------
    // activations: base = address of tap (0,0), channel 0 of the block's first output pixel (may lie before the tensor
    // for padded borders; it is only ever used with offsets that land inside).  Input pixel index is monotone in m
    // (checked at launch: kh >= 2*ph + 1, kw >= 2*pw), so every row's offset from this base is >= 0.
    long pix0;
    {
        const int ow = (int)(m0 % p.OW);
        const long t = m0 / p.OW;
        const int oh = (int)(t % p.OH);
        const long n = t / p.OH;
        pix0 = (n * p.Hs + (long)oh * p.sh - p.ph) * p.Ws + (long)ow * p.sw - p.pw;
    }
#if VSE_GEMM_ASM
    const rsrc4_t rsA = make_rsrc4(p.in + pix0 * p.in_ld);
    const rsrc4_t rsW = make_rsrc4(wsrc + (long)n0 * ((p.flags & F_WK32) ? 32 : 64));
#else
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + pix0 * p.in_ld), 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(wsrc + (long)n0 * ((p.flags & F_WK32) ? 32 : 64)), 0, 0x7fffffff, 0x00020000);
#endif

    // ---- per-thread, loop-invariant offsets ------------------------------------------------------------------
    // wave instruction j of this wave covers tile rows (j*NW + wave)*RPI .. +RPI-1; lane l -> row + l/KV, physical
    // 16-byte slot l%KV, i.e. logical k-vector (l%KV) ^ swz(row)  (source-side swizzle; the LDS image is lane-linear)
    const int rsub = lane / KV;
    unsigned voffA[NA], ntap[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int r = (j * NW + wave) * RPI + rsub;
        const int kv = (lane % KV) ^ swz(r);
        const long m = m0 + r;
        voffA[j] = OOB;
        ntap[j] = 0xffffffffu;
        if (m < mend) {
            const int ow = (int)(m % p.OW);
            const long t = m / p.OW;
            const int oh = (int)(t % p.OH);
            const long n = t / p.OH;
            const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
            const long pix = (n * p.Hs + ih0) * p.Ws + iw0;
            voffA[j] = (unsigned)((pix - pix0) * p.in_ld * 2 + kv * 16);
            if constexpr (MASK) {
                unsigned okm = 0;
                for (int dy = 0; dy < p.kh; ++dy)
                    for (int dx = 0; dx < p.kw; ++dx)
                        okm |= (unsigned)(ih0 + dy >= 0 && ih0 + dy < p.H && iw0 + dx >= 0 && iw0 + dx < p.W) << (dy * p.kw + dx);
                ntap[j] = ~okm;
            }
        }
    }
    const bool wk32 = p.flags & F_WK32;
    unsigned voffW[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r = (j * NW + wave) * RPI + rsub;
        const int kv = (lane % KV) ^ swz(r);
        // weights tiled [Kp/64][Np][64], or (F_WK32, BK 32 only) [Kp/32][Np][32]: a wave DMA then covers 1 KiB of
        // CONTIGUOUS memory = 8 whole 128-byte lines instead of 16 half lines (half the L2 requests of the weight stream)
        // (BK 64 over 32-deep tiles: k-vectors 4..7 of a row come from the next tile, Np * 64 bytes further)
        voffW[j] = !(r < BN && n0 + r < p.Np) ? OOB
                 : wk32 ? (unsigned)((kv >> 2) * p.Np * 64 + r * 64 + (kv & 3) * 16) : (unsigned)(r * 128 + kv * 16);
    }
    const unsigned wstep = (unsigned)p.Np * (wk32 ? 64u : 128u);       // bytes per weight K tile

    // ---- block-uniform K walk (SALU) ---------------------------------------------------------------------------
    int kc = 0, dx = 0, dy = 0, tap = 0;
    auto issue = [&](int kt, int st) {
        half_t* base = lds + st * STAGE_HALFS;
        const int soffA = ((dy * p.Ws + dx) * p.in_ld + kc) * 2;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            unsigned off = voffA[j];
            if constexpr (MASK) off |= __builtin_amdgcn_ubfe(ntap[j], (unsigned)tap, 1u) << 31;   // v_bfe_u32 + v_lshl_or_b32
#if VSE_GEMM_ASM
            bufdma16_asm(rsA, off, soffA, base + (j * NW + wave) * RPI * BKT);
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsv_t)(base + (j * NW + wave) * RPI * BKT), 16, (int)off, soffA, 0, VSE_GEMM_NT_A ? 2 : 0);
#endif
        }
        const int soffW = wk32 ? (int)((unsigned)kt * (BKT / 32) * wstep)
                         : BKT == 64 ? (int)((unsigned)kt * wstep) : (int)((unsigned)(kt >> 1) * wstep + (unsigned)(kt & 1) * 64u);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#if VSE_GEMM_ASM
            bufdma16_asm(rsW, voffW[j], soffW, base + BM * BKT + (j * NW + wave) * RPI * BKT);
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (ldsv_t)(base + BM * BKT + (j * NW + wave) * RPI * BKT), 16, (int)voffW[j], soffW, 0, 0);
#endif
        kc += BKT;
        if (kc >= p.cinp) {
            kc = 0;
            ++tap;
            if (++dx == p.kw) { dx = 0; ++dy; }
        }
        if (kt + 1 == p.nkh) { kc = 0; dx = 0; dy = 0; tap = 0; }      // F_HILO: second pass (lo weight tiles), same activations
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    conv_stage_consts<VSE_GEMM_ASM != 0>(sbias, p.bias, p.zero, n0, BN, p.Np, wave, lane);
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < p.nk) issue(s, s);

    // fragment addresses inside a stage: row*ROWB + swizzled 16-byte slot; tiles of one wave are 32 rows apart, which
    // keeps swz(row), so tile i / stage S are immediates on top of one per-thread VGPR per k sub-step and operand
    const int frow = lane & 31, fj = lane >> 5;
    const char* xptr[KS];
    const char* wptr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int rx = wm * WTM + frow, rw = wn * WTN + conv_wrow(frow);
        xptr[ks] = (const char*)lds + rx * ROWB + (((ks * 2 + fj) ^ swz(rx)) << 4);
        wptr[ks] = (const char*)lds + BM * ROWB + rw * ROWB + (((ks * 2 + fj) ^ swz(rw)) << 4);
    }

    int kt = 0;
    auto step = [&](auto stage_c) {
        constexpr int S = decltype(stage_c)::value;
        // WAR on the ring slot refilled below: it was read during step kt-1, and those ds_reads were retired by the
        // lgkmcnt waits in front of that step's last MFMAs — keep this step's barrier behind them (no hoisting)
        __builtin_amdgcn_sched_barrier(0);
        // tile kt has landed once only the loads of the ST-2 tiles behind it can be outstanding (vmcnt retires in order)
        const int rem = p.nk - 1 - kt;
        if (rem >= ST - 2) wait_vm<(ST - 2) * LPT>();
        else if (ST >= 5 && rem == 2) wait_vm<2 * LPT>();
        else if (ST >= 4 && rem == 1) wait_vm<LPT>();
        else wait_vm<0>();
#if VSE_ABLATE != 1
        __builtin_amdgcn_s_barrier();              // every thread's part of tile kt is in LDS; compute(kt-1) is over
#endif
        asm volatile("" ::: "memory");
#if VSE_ABLATE != 3
        if (kt + ST - 1 < p.nk) issue(kt + ST - 1, (S + ST - 1) % ST);
#endif
        // fragment reads in groups of KG k sub-steps (<= 16 ds_read_b128 in flight), each followed by its MFMAs
        // (16-wave tiles run 4 waves per SIMD = 128 VGPRs: at most 8 fragments = 32 VGPRs in flight beside the accumulators)
        constexpr int FCAP = NW >= 16 ? 8 : 16;
        constexpr int KG = (KS * (TM + TN) <= FCAP) ? KS : (KS / 2 * (TM + TN) <= FCAP ? KS / 2 : 1);
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KG) {
            half8 wf[KG][TN], xf[KG][TM];
#pragma unroll
            for (int ks = 0; ks < KG; ++ks) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#if VSE_ABLATE == 2
                    wf[ks][j] = half8{(half_t)lane, 0, 0, 0, 0, 0, 0, 0};
#else
                    wf[ks][j] = *reinterpret_cast<const half8*>(wptr[k0 + ks] + (S * STAGE_HALFS + j * 32 * BKT) * 2);
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
#if VSE_ABLATE == 2
                    xf[ks][i] = half8{(half_t)kt, 0, 0, 0, 0, 0, 0, 0};
#else
                    xf[ks][i] = *reinterpret_cast<const half8*>(xptr[k0 + ks] + (S * STAGE_HALFS + i * 32 * BKT) * 2);
#endif
            }
#pragma unroll
            for (int ks = 0; ks < KG; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#if VSE_ABLATE == 4
                        acc[i][j][0] += (float)wf[ks][j][0] + (float)xf[ks][i][0];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_group_barrier(0x100, KG * (TM + TN), 0);   // this group's ds_reads ...
            __builtin_amdgcn_sched_group_barrier(0x008, KG * TM * TN, 0);     // ... then its MFMAs
        }
    };
    // whole ring revolutions without exits (keeps the accumulators in place), then a tail of at most ST-1 steps
    for (; kt + ST <= p.nk;) {
        step(std::integral_constant<int, 0>{}); ++kt;
        step(std::integral_constant<int, 1>{}); ++kt;
        if constexpr (ST >= 3) { step(std::integral_constant<int, 2>{}); ++kt; }
        if constexpr (ST >= 4) { step(std::integral_constant<int, 3>{}); ++kt; }
        if constexpr (ST >= 5) { step(std::integral_constant<int, 4>{}); ++kt; }
    }
    if (kt < p.nk) {
        step(std::integral_constant<int, 0>{}); ++kt;
        if constexpr (ST >= 3) {
            if (kt < p.nk) { step(std::integral_constant<int, 1>{}); ++kt; }
        }
        if constexpr (ST >= 4) {
            if (kt < p.nk) { step(std::integral_constant<int, 2>{}); ++kt; }
        }
        if constexpr (ST >= 5) {
            if (kt < p.nk) { step(std::integral_constant<int, 3>{}); ++kt; }
        }
    }

    // ---- epilogue (shared with conv_mfma_kernel) ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * WTM + i * 32 + (lane & 31);
        if (m >= mend) continue;
        const int ow = (int)(m % p.OW);
        const long t = m / p.OW;
        const int oh = (int)(t % p.OH);
        const long n = t / p.OH;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#if VSE_ABLATE == 5
            if (acc[i][j][0] == 12345.678f)
#endif
            {
                float bias[16];
                conv_epilogue_consts(sbias, wn * WTN + j * 32, lane, bias);
                conv_epilogue_tile(p, acc[i][j], bias, m, n, oh, ow, n0 + wn * WTN + j * 32, lane);
            }
    }
}

// 0 = not eligible; 1 = masked variant; 2 = unmasked 1x1 variant
int conv_gemm_mode(int kh, int kw, int sh, int sw, int ph, int pw, int cinp, int Kp, int inshift, int flags) {
    (void)sh; (void)sw;
    if (inshift || (flags & (F_PATCH | F_DOT1 | F_SRC2))) return 0;
    if (cinp % 32) return 0;
    if (kh * kw > 31) return 0;
    if (kh < 2 * ph + 1 || kw < 2 * pw) return 0;
    if (kh == 1 && kw == 1 && ph == 0 && pw == 0 && Kp == cinp) return 2;
    return 1;
}

// Tile configurations (BM x BN, waves, BK, stages).  Index = the `cfg` field of vse_plan_op_variant() (bench.py mirrors
// the names).  Measured and dropped on MI355X (tools/bench_conv.py): BK = 64 rings with 2-3 stages (fewer
// bytes in flight per CU, -5..-25 %), 4-stage 128 x 128 (2 blocks/CU, -10 %), 512 x 128, 8-wave 256 x 256 (VGPR
// spills), and a persistent one-block-per-slot variant of every shape (-5..-15 %: the hardware already overlaps
// one block's store tail with its neighbours' K loops, and stores share vmcnt with the LDS-DMAs); 4-stage rings for
// the 16-wave tiles (128 / 112 KiB, 0..-12 %: more bytes in flight do not help, tools/ubench/fill.hip shows why: the
// L2 takes ~1 request per channel clock, i.e. ~32 B/clk/CU of 64-byte row segments chip-wide, HBM streams at ~10 B/clk/CU,
// and a stream that mixes both gets ~15 B/clk/CU), BK 64 / 2 stages for 256 x 256 (8x slower: spills).
//   0: 128 x 128,  4 waves (2 x 2), BK 32, 3 stages  (48 KiB LDS, 3 blocks/CU)  — the conv_mfma_kernel shape
//   1: 256 x  64,  4 waves (4 x 1), BK 32, 3 stages  (60 KiB, 2 blocks/CU)
//   2: 256 x  32,  4 waves (4 x 1), BK 32, 3 stages
//   6: 256 x 128,  8 waves (4 x 2), BK 32, 3 stages  (72 KiB, 2 blocks/CU)
//  16: 256 x 256, 16 waves (4 x 4), BK 32, 3 stages  (96 KiB, 1 block/CU)
//  17: 256 x 192, 16 waves (8 x 2), BK 32, 3 stages  (96 KiB, 1 block/CU; weight rows staged as 256)
//  18 / 19: the same two tiles with BK 64 and 2 stages (128 KiB) for layers with cin % 64 == 0
struct GemmCfg { int bm, bn, bk; };
static const GemmCfg kCfg[] = {{128, 128, 32}, {256, 64, 32}, {256, 32, 32}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {256, 128, 32},
                               {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
                               {256, 256, 32}, {256, 192, 32}, {256, 256, 64}, {256, 192, 64}};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

// Which configuration serves a layer.  The kernel is bound by the L2 -> LDS fill (ablation: MFMAs and fragment reads
// are free, the DMA stream and the store tail are not), so the choice minimises fetched bytes: one cout tile when the
// couts fit 192 / 256 (the activation tile is then fetched once instead of 2-3 times), 256-pixel tiles (weights re-read
// half as often), as long as the grid still fills the 256 CUs.  Zero-padded couts cost MFMA issue slots only.
int conv_gemm_config(int Np, int cinp, long M) {
#ifdef VSE_DEV_BUILD
    const char* e = getenv("VSE_GEMM_CFG");          // experiments: force a configuration where it is legal
    if (e && e[0]) {
        const int c = atoi(e);
        if (c >= 0 && c < kNumCfg && kCfg[c].bm && cinp % kCfg[c].bk == 0) return c;
    }
#endif
    auto ntn = [&](int bn) { return (long)((Np + bn - 1) / bn); };
    // the 16-wave tiles run 64-deep K tiles in a 2-stage ring where the channels allow it (whole 128-byte lines per
    // activation row and half the barriers; A/B on one box: -2..-4 %); the 8-wave 256 x 128 tile loses its second block per CU
    auto deep = [&](int c) { return cinp % 64 == 0 ? c + 2 : c; };          // 16 -> 18, 17 -> 19
    if (Np <= 32) return 2;
    if (Np <= 64) return 1;
    const long mt = (M + 255) / 256;
    if (Np <= 128) return mt >= 256 ? 6 : 0;
    if (Np <= 192) return mt >= 192 ? deep(17) : (mt >= 128 ? 6 : 0);
    const double w256 = (double)ntn(256) * 256 / Np, w192 = (double)ntn(192) * 192 / Np;
    const bool pick256 = ntn(256) < ntn(192) || (ntn(256) == ntn(192) && w256 <= w192);
    if (pick256 && mt * ntn(256) >= 192 && w256 <= 1.34) return deep(16);
    if (mt * ntn(192) >= 192 && w192 <= 1.34) return deep(17);
    if (mt * ntn(256) >= 192 && w256 <= 1.34) return deep(16);
    return mt * ntn(128) >= 512 ? 6 : 0;
}

template <int BM, int BN, int WM, int WN, int BKT, int ST>
static void launch_cfg(const ConvParams& p, int mode, dim3 grid, hipStream_t st) {
    dim3 block(64 * WM * WN);
    if (mode == 1) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, BKT, ST, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, BKT, ST, 0>), grid, block, 0, st, p);
}

int launch_conv_gemm(ConvParams& p, int Kp, hipStream_t st) {
    const int mode = conv_gemm_mode(p.kh, p.kw, p.sh, p.sw, p.ph, p.pw, p.cinp, Kp, p.inshift, p.flags);
    if (!mode) return VSE_E_UNSUPPORTED;
    p.nkh = Kp / ((p.flags & F_WK32) ? 32 : 64);                         // (K tiles of one weight pass, as conv_smallm.hip walks them)
    if (conv_smallm_ok(p, mode)) return launch_conv_smallm(p, st);       // a handful of pixels (SE gates): conv_smallm.hip
    // 32-bit offsets: the rows of one block span at most BM output pixels (+ one image seam), the tap walk kh rows
    const double span = ((double)512 * p.sw + (512.0 / p.OW + 3) * p.sh * p.Ws + (double)p.kh * p.Ws + p.kw) * p.in_ld * 2;
    if (span > 1.9e9 || (double)Kp * p.Np * 2 > 1.9e9) return VSE_E_UNSUPPORTED;
    const int c = conv_gemm_config(p.Np, p.cinp, p.M);
    const GemmCfg& g = kCfg[c];
    p.ntn = (unsigned)((p.Np + g.bn - 1) / g.bn);
    p.nk = Kp / g.bk;
    p.nkh = p.nk;
    if (p.flags & F_HILO) p.nk *= 2;
    unsigned long long tiles = (unsigned long long)((p.M + g.bm - 1) / g.bm) * p.ntn;
    if (p.flags & F_IMGW) {
        if (mode != 2 || p.hw_img <= 0 || p.M % p.hw_img) return VSE_E_UNSUPPORTED;      // unmasked 1x1 only
        p.tiles_img = (p.hw_img + g.bm - 1) / g.bm;
        tiles = (unsigned long long)(p.M / p.hw_img) * p.tiles_img * p.ntn;
    }
    if (tiles == 0 || tiles > 0x7fffffffull) return VSE_E_INVAL;
    dim3 grid((unsigned)tiles);
    switch (c) {
        case 1: launch_cfg<256, 64, 4, 1, 32, 3>(p, mode, grid, st); break;
        case 2: launch_cfg<256, 32, 4, 1, 32, 3>(p, mode, grid, st); break;
        case 6: launch_cfg<256, 128, 4, 2, 32, 3>(p, mode, grid, st); break;
        case 16: launch_cfg<256, 256, 4, 4, 32, 3>(p, mode, grid, st); break;
        case 17: launch_cfg<256, 192, 8, 2, 32, 3>(p, mode, grid, st); break;
        case 18: launch_cfg<256, 256, 4, 4, 64, 2>(p, mode, grid, st); break;
        case 19: launch_cfg<256, 192, 8, 2, 64, 2>(p, mode, grid, st); break;
        default: launch_cfg<128, 128, 2, 2, 32, 3>(p, mode, grid, st); break;
    }
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
