// Micro-benchmark: how fast can one CU fill LDS with buffer_load ... lds (LDS-DMA), as a function of the access shape
// (bytes per row segment), the ring depth, the waves per block / blocks per CU and the source (HBM stream vs L2-resident)?
// Emulates the activation stream of conv_gemm_kernel: a tile = BM rows of `ld` bytes; step kt loads SEG bytes of every row
// at column kt*SEG; the ring holds ST stages; one barrier per step (optional).  No MFMAs, no LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill tools/ubench/fill.hip && build/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* ldsv_t;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int NW, int SEG, int ST, bool BAR>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const char* src, int ld, int nk, int tiles_per_block, long tile_stride,
                                                       int resident_mod, unsigned* sink, const char* wsrc = nullptr) {
    constexpr int LPR = SEG / 16;            // lanes per row segment
    constexpr int RPI = 64 / LPR;            // rows per wave instruction
    constexpr int NA = BM / (RPI * NW);
    static_assert(NA >= 1 && BM % (RPI * NW) == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned voff[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) voff[j] = (unsigned)(((j * NW + wave) * RPI + lane / LPR) * ld + (lane % LPR) * 16);
    for (int t = 0; t < tiles_per_block; ++t) {
        long tile = (long)blockIdx.x * tiles_per_block + t;
        if (resident_mod) tile %= resident_mod;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + tile * tile_stride), 0, 0x7fffffff, 0x00020000);
        const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wsrc ? wsrc : src + tile * tile_stride), 0, 0x7fffffff, 0x00020000);
        auto issue = [&](int kt, int st) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds((wsrc && j >= NA / 2) ? rw : rs, (ldsv_t)(lds + st * BM * SEG + (j * NW + wave) * RPI * SEG), 16,
                                                         (int)voff[j], kt * SEG, 0, 0);
        };
        for (int s = 0; s < ST - 1; ++s) issue(s, s);
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + ST - 2 < nk) wait_vm<(ST - 2) * NA>(); else wait_vm<0>();
            if (BAR) __builtin_amdgcn_s_barrier();
            int nst = st + ST - 1; if (nst >= ST) nst -= ST;
            if (kt + ST - 1 < nk) issue(kt + ST - 1, nst);
            if (++st == ST) st = 0;
        }
        wait_vm<0>();
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    if (sink && lds[threadIdx.x * 16] == 77 && lane == 99) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int BM, int NW, int SEG, int ST, bool BAR>
void run(const char* name, const char* buf, size_t bufbytes, int ld, int blocks_per_cu, bool resident, const char* wbuf = nullptr) {
    const int nk = ld / SEG;
    const long tile_stride = (long)BM * ld;
    const int grid = 256 * blocks_per_cu;
    long tiles_total = (long)(bufbytes / tile_stride);
    int tpb = (int)(tiles_total / grid);
    if (tpb > 64) tpb = 64;
    if (tpb < 1) { printf("%s: buffer too small\n", name); return; }
    const int lds_bytes_min = ST * BM * SEG;
    // force the residency: pad dynamic LDS so that exactly blocks_per_cu blocks fit (160 KiB per CU)
    int lds_bytes = 160 * 1024 / blocks_per_cu - 1024;
    if (lds_bytes < lds_bytes_min) { printf("%s: LDS does not fit\n", name); return; }
    auto k = fill_kernel<BM, NW, SEG, ST, BAR>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int rmod = resident ? 8 : 0;     // 8 tiles shared by everybody -> L2 hits
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), lds_bytes, 0, buf, ld, nk, tpb, tile_stride, rmod, (unsigned*)nullptr, wbuf);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)grid * tpb * BM * nk * SEG;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s BM %3d NW %2d SEG %3d ST %d bar %d  blocks/CU %d  %s  ld %5d: %7.3f ms  %6.2f TB/s  %5.1f B/cyc/CU\n", name, BM, NW, SEG,
           ST, (int)BAR, blocks_per_cu, resident ? "L2 " : "HBM", ld, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char* buf; CK(hipMalloc(&buf, bytes + (64 << 20))); CK(hipMemset(buf, 1, bytes + (64 << 20)));
    // mixed: half of every stage streams from HBM (activations), half re-reads one resident tile (weights)
    for (int ld : {1792, 2432}) {
        printf("mixed HBM + L2 (bytes counted: both halves)\n");
        run<512, 16, 64, 3, true>("512 rows: 256 HBM + 256 L2", buf, bytes, ld, 1, false, buf + ((size_t)2 << 30));
        run<512, 16, 128, 2, true>("  BK64 2 stages", buf, bytes, ld, 1, false, buf + ((size_t)2 << 30));
        run<512, 8, 64, 3, true>("  8 waves", buf, bytes, ld, 1, false, buf + ((size_t)2 << 30));
        run<256, 8, 64, 3, true>("256 rows: 128 HBM + 128 L2, 2 blocks/CU", buf, bytes, ld, 2, false, buf + ((size_t)2 << 30));
        run<384, 8, 64, 3, true>("384 rows: 192 HBM + 192 L2, 2 blocks/CU", buf, bytes, ld, 2, false, buf + ((size_t)2 << 30));
    }
    for (int resident = 0; resident < 0; ++resident) {
        const bool r = resident;
        for (int ld : {1792, 512, 4224}) {
            run<256, 16, 64, 3, true>("gemm 256x256 A-stream (16 waves)", buf, bytes, ld, 1, r);
            run<256, 16, 64, 4, true>("  4 stages", buf, bytes, ld, 1, r);
            run<256, 16, 128, 3, true>("  BK64", buf, bytes, ld, 1, r);
            run<256, 16, 128, 2, true>("  BK64 2 stages", buf, bytes, ld, 1, r);
            run<256, 16, 256, 2, true>("  BK128 2 stages", buf, bytes, ld, 1, r);
            run<256, 16, 64, 3, false>("  no barrier", buf, bytes, ld, 1, r);
            run<256, 8, 64, 3, true>("256 rows, 8 waves, 2 blocks/CU", buf, bytes, ld, 2, r);
            run<256, 8, 128, 3, true>("  BK64", buf, bytes, ld, 2, r);
            run<128, 4, 64, 3, true>("128 rows, 4 waves, 3 blocks/CU", buf, bytes, ld, 3, r);
            run<128, 4, 64, 3, true>("128 rows, 4 waves, 4 blocks/CU", buf, bytes, ld, 4, r);
            run<128, 4, 128, 3, true>("  BK64 4 blocks/CU", buf, bytes, ld, 4, r);
            run<512, 16, 64, 3, true>("512 rows (A+B volume of 256x256)", buf, bytes, ld, 1, r);
            run<512, 16, 128, 2, true>("  BK64 2 stages", buf, bytes, ld, 1, r);
        }
    }
    return 0;
}
