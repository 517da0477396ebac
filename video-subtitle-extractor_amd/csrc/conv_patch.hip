// k x k stride-1 convolution with an LDS-RESIDENT INPUT PATCH (gfx950 / CDNA4).
//
// The implicit-GEMM kernel (conv_mfma.hip) re-fetches its activation tile from L2 for every filter tap; PMC showed
// it bound by the L2 -> L1 -> LDS fill path (TA busy 50-70 %, L1 hit rate < 20 %, MFMA busy < 20 %).  Here a block
// owns an 8 x 32 tile of output pixels of ONE image and, per 32-channel chunk, pulls the (8+kh-1) x (32+kw-1) halo
// patch into LDS ONCE; all kh*kw taps then read their MFMA B-fragments from that patch at shifted pixel addresses.
// Activation bytes crossing L2->LDS drop by ~kh*kw / halo overhead (9x9: 81 taps / 2.5 = 32x; 3x3: 9 / 1.33 = 6.8x);
// what is left to stream per (chunk, tap) step is the [BN][32] weight tile (4-8 KiB), through a 4-deep LDS-DMA ring.
//
//   block  = 512 threads = 8 waves: 4 (pixel rows 2w,2w+1) x 2 (cout halves); wave tile = 64 px x BN/2 couts
//   LDS    = 2 x 40 KiB patch (double-buffered across channel chunks) + 4 x 8 KiB weight ring = 112 KiB, one object
//   sync   = ONE raw s_barrier per (chunk, tap) step; LDS-DMA completion by counted s_waitcnt vmcnt(N):
//            per step every thread issues exactly 1 weight DMA (+5 patch DMAs at tap 0, zero-page dummies included),
//            so N is a literal: 7 at taps 1,2 (the patch DMAs of the NEXT chunk are younger than the stage waited
//            for), 2 otherwise (look-ahead 3 steps)
//   banks  = 64-byte rows; 16-byte slot s of row/pixel q holds logical k-vector s ^ ((q >> 2) & 3), applied on the
//            DMA source side; 16 consecutive pixels at ANY alignment then hit 16 distinct bank groups (shifted taps
//            stay conflict-free)
//   weights are packed [chunk][tap][Np][32] by the compiler for this kernel (F_PATCH) so the stream is sequential.
#include "conv_common.h"

#define PTH 8
#define PTW 32
#define PNPL 5            // patch DMAs per thread per chunk: 5 * 512 threads * 16 B = 640 pixels * 64 B
#define PPIX 640
#define PRING 4
#define PATCH_HALFS (PPIX * 32)
#define WSTAGE_HALFS (128 * 32)

template <int BN>
__global__ __launch_bounds__(512) void conv_patch_kernel(const ConvParams p) {
    constexpr int TN = BN / 64;                     // 32-cout MFMA tiles per wave
    __shared__ __attribute__((aligned(16))) half_t lds[2 * PATCH_HALFS + PRING * WSTAGE_HALFS];   // the ONLY LDS object
    half_t* const patch0 = lds;
    half_t* const ring0 = lds + 2 * PATCH_HALFS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = wave >> 1, wco = wave & 1;

    // XCD-aware bijective block order (see conv_mfma.hip): contiguous logical range per XCD, cout tiles innermost
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int nt = t % p.ntn;  t /= p.ntn;
    const int tx = t % p.tiles_w;  t /= p.tiles_w;
    const int ty = t % p.tiles_h;
    const long img = t / p.tiles_h;
    const int oy0 = ty * PTH, ox0 = tx * PTW, n0 = nt * BN;

    const int PW = PTW + p.kw - 1, PH = PTH + p.kh - 1, P = PW * PH;
    const int taps = p.kh * p.kw;
    const int nchunks = p.cinp >> 5;
    const int total = nchunks * taps;

    // ---- DMA source state ---------------------------------------------------------------------------------
    const int kv = (lane & 3) ^ ((lane >> 4) & 3);        // logical k-vector this lane fetches (source-side swizzle)
    long poff[PNPL];
    bool pok[PNPL];
#pragma unroll
    for (int j = 0; j < PNPL; ++j) {
        const int q = 16 * (wave + 8 * j) + (lane >> 2);    // patch pixel index; wave instruction covers 16 pixels
        const int py = q / PW, px = q - py * PW;
        const int iy = oy0 - p.ph + py, ix = ox0 - p.pw + px;
        pok[j] = (q < P) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        poff[j] = ((img * p.Hs + (iy >> p.inshift)) * p.Ws + (ix >> p.inshift)) * (long)p.in_ld + kv * 8;
    }
    const int wr = tid >> 2;                               // weight row (cout within the block tile), 0..127
    const bool wok = (wr < BN) && (n0 + wr < p.Np);
    const half_t* wsrc = p.w + (long)(n0 + wr) * 32 + kv * 8;

    auto issue_patch = [&](int cc, int buf) {
        half_t* base = patch0 + buf * PATCH_HALFS;
        const bool live = cc < nchunks;
#pragma unroll
        for (int j = 0; j < PNPL; ++j)
            glds16((live && pok[j]) ? p.in + poff[j] + cc * 32 : p.zero, base + (wave + 8 * j) * 16 * 32);
    };
    auto issue_w = [&](int s) {
        glds16((wok && s < total) ? wsrc + (long)s * p.Np * 32 : p.zero, ring0 + (s & (PRING - 1)) * WSTAGE_HALFS + wave * 16 * 32);
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------
    const int fx = lane & 31, fj = lane >> 5;
    int woff[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wco * (BN / 2) + j * 32 + fx;
            woff[j][ks] = r * 32 + (((ks * 2 + fj) ^ ((r >> 2) & 3)) << 3);
        }
    const int qb0 = (2 * wpx) * PW + fx, qb1 = qb0 + PW;

    float16v acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue_patch(0, 0);
    issue_w(0);
    issue_w(1);
    issue_w(2);

    int s = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const half_t* pbuf = patch0 + (cc & 1) * PATCH_HALFS;
        int tapoff = 0, dx = 0;                            // dy*PW + dx
        for (int tap = 0; tap < taps; ++tap, ++s) {
            if (tap == 1 || tap == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tap == 0) issue_patch(cc + 1, (cc + 1) & 1);
            issue_w(s + 3);
            const half_t* wst = ring0 + (s & (PRING - 1)) * WSTAGE_HALFS;
            const int q0 = qb0 + tapoff, q1 = qb1 + tapoff;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 wf[TN], xf[2];
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8*>(wst + woff[j][ks]);
                xf[0] = *reinterpret_cast<const half8*>(pbuf + q0 * 32 + (((ks * 2 + fj) ^ ((q0 >> 2) & 3)) << 3));
                xf[1] = *reinterpret_cast<const half8*>(pbuf + q1 * 32 + (((ks * 2 + fj) ^ ((q1 >> 2) & 3)) << 3));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            if (++dx == p.kw) { dx = 0; tapoff += PW - p.kw + 1; } else { ++tapoff; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the zero-page dummies before LDS is released

    // ---- epilogue ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wpx + i, ox = ox0 + fx;
        if (oy >= p.OH || ox >= p.OW) continue;
        const long m = (img * p.OH + oy) * p.OW + ox;
#pragma unroll
        for (int j = 0; j < TN; ++j) conv_epilogue_tile(p, acc[i][j], m, img, oy, ox, n0 + wco * (BN / 2) + j * 32, lane);
    }
}

int launch_conv_patch(const ConvParams& pin, int n_img, hipStream_t st) {
    ConvParams p = pin;
    if (p.sh != 1 || p.sw != 1 || p.kh * p.kw < 3 || (p.cinp & 31) || (p.flags & F_PIXSHUF)) return VSE_E_INVAL;
    if ((PTH + p.kh - 1) * (PTW + p.kw - 1) > PPIX) return VSE_E_UNSUPPORTED;
    const int bn = p.Np <= 64 ? 64 : 128;
    p.ntn = (unsigned)((p.Np + bn - 1) / bn);
    p.tiles_h = (p.OH + PTH - 1) / PTH;
    p.tiles_w = (p.OW + PTW - 1) / PTW;
    const unsigned long long blocks = (unsigned long long)n_img * p.tiles_h * p.tiles_w * p.ntn;
    if (blocks == 0 || blocks > 0x7fffffffull) return VSE_E_INVAL;
    if (bn == 64) hipLaunchKernelGGL((conv_patch_kernel<64>), dim3((unsigned)blocks), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_patch_kernel<128>), dim3((unsigned)blocks), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? VSE_OK : VSE_E_HIP;
}
