// Counter calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an
// absolute"): kernels that read / write a KNOWN number of bytes with the access shapes of this repo's HBM-bound kernel families, to be run
// under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes).  tools/counter_calibration.py divides the known bytes by
// the counters and writes profiles/rNN_counter_calibration.json; tools/summarize_profiles.py applies the factor per kernel family.
//
//   calib_stream_read        16 B per lane, consecutive lanes = consecutive 16-byte vectors (the guide's calibrated case: expect 2.0)
//   calib_pix_read<KS>       conv_pw / conv_dwpw centre tap: lane (pixel fx, k-half fj) reads 16 B of pixel m at channel 16 ks + 8 fj,
//                            pixel stride = 32 KS bytes; every byte of the tensor is read exactly once (no neighbourhood)
//   calib_pix_read3x3<KS, XCD>  conv_dwpw: the same lane gathers the clamped 3 x 3 neighbourhood (9 loads per slice); the tensor is still
//                            only `bytes` large: anything FETCH_SIZE reports above the centre-tap figure is re-fetching (L2 misses).
//                            XCD = the product kernels' XCD-contiguous block order (xcd_block), false = plain blockIdx order
//   calib_dwrow_read<C8, XCD>   dwconv_row_kernel: lane = (8-channel group g fastest, quad of 4 output pixels), 6 x 16 B per row, 3 rows
//   calib_stream_write       16 B per lane streaming stores
//   calib_pix_write<C8>      the conv epilogue's store: lane (pixel, half) writes two 16-byte runs at channel 16 g + 8 half of its pixel
//
// Tensors are sized >= 0.5 GB (twice the 256 MiB Infinity Cache) so that nothing is served on-die.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/counter_calib tools/ubench/counter_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the product kernels' XCD-contiguous block order (csrc/common.h xcd_block): hardware block b runs on XCD b & 7; give every XCD one
// contiguous run of logical blocks so that vertically adjacent rows meet in one L2
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblk) {
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

__device__ __forceinline__ float sum8(half8 v) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)v[e];
    return s;
}

__global__ __launch_bounds__(256) void calib_stream_read(const half8* __restrict__ src, long nvec, float* sink) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) s += sum8(src[i]);
    if (s == 1234.5f) sink[0] = s;
}

// M pixels of C = 16 KS channels; a block = 4 waves x 64 pixels (two 32-pixel tiles per wave), as conv_dwpw_kernel
template <int KS>
__global__ __launch_bounds__(256) void calib_pix_read(const half_t* __restrict__ src, long M, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fx = lane & 31, fj = lane >> 5;
    float s = 0.f;
    for (int i = 0; i < 2; ++i) {
        const long m = (long)blockIdx.x * 256 + wave * 64 + i * 32 + fx;
        if (m >= M) continue;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s += sum8(*reinterpret_cast<const half8*>(src + m * (16 * KS) + ks * 16 + fj * 8));
    }
    if (s == 1234.5f) sink[0] = s;
}

template <int KS, bool XCD>
__global__ __launch_bounds__(256) void calib_pix_read3x3(const half_t* __restrict__ src, int N, int H, int W, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fx = lane & 31, fj = lane >> 5;
    const long M = (long)N * H * W;
    float s = 0.f;
    const long blk = XCD ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    for (int i = 0; i < 2; ++i) {
        const long m = blk * 256 + wave * 64 + i * 32 + fx;
        if (m >= M) continue;
        const unsigned mu = (unsigned)m, t = mu / (unsigned)W, n = t / (unsigned)H;
        const int ow = (int)(mu - t * W), oh = (int)(t - n * H);
        const half_t* img = src + (long)n * H * W * (16 * KS);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int cy = min(max(oh + dy, 0), H - 1);
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int cx = min(max(ow + dx, 0), W - 1);
                    s += sum8(*reinterpret_cast<const half8*>(img + ((long)cy * W + cx) * (16 * KS) + ks * 16 + fj * 8));
                }
            }
    }
    if (s == 1234.5f) sink[0] = s;
}

// dwconv_row_kernel<3, 1>: thread = (channel group g fastest, quad of 4 output pixels, row, image); 6 input pixels x 3 rows, stride 1, pad 1
template <int C8, bool XCD>
__global__ __launch_bounds__(256) void calib_dwrow_read(const half_t* __restrict__ src, int N, int H, int W, float* sink) {
    const int owq = (W + 3) / 4;
    const long total = (long)N * H * owq * C8;
    float s = 0.f;
    for (long i = (long)(XCD ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int g = (int)(i % C8);
        long t = i / C8;
        const int q = (int)(t % owq);
        t /= owq;
        const int oh = (int)(t % H);
        const long n = t / H;
        for (int dy = -1; dy <= 1; ++dy) {
            const int ih = oh + dy;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int iw = q * 4 - 1 + c;
                if (iw >= 0 && iw < W) s += sum8(*reinterpret_cast<const half8*>(src + ((n * H + ih) * W + iw) * (long)(8 * C8) + g * 8));
            }
        }
    }
    if (s == 1234.5f) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_stream_write(half8* __restrict__ dst, long nvec) {
    const half8 v = {1, 2, 3, 4, 5, 6, 7, 8};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) dst[i] = v;
}

// conv_epilogue_tile's store shape for a layer with 8 C8 couts: per 32-cout accumulator tile lane (pixel fx, half fj) stores the two
// runs 32 j + 16 g + 8 fj (g = 0, 1) of its pixel
template <int C8>
__global__ __launch_bounds__(256) void calib_pix_write(half_t* __restrict__ dst, long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fx = lane & 31, fj = lane >> 5;
    const half8 v = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < 2; ++i) {
        const long m = (long)blockIdx.x * 256 + wave * 64 + i * 32 + fx;
        if (m >= M) continue;
        for (int j = 0; j < (C8 + 3) / 4; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c0 = j * 32 + g * 16 + fj * 8;
                if (c0 < 8 * C8) *reinterpret_cast<half8*>(dst + m * (8 * C8) + c0) = v;
            }
    }
}

int main() {
    const size_t cap = (size_t)1 << 30;      // 1 GiB arena
    char* buf;
    float* sink;
    CK(hipMalloc(&buf, cap));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, cap));
    CK(hipDeviceSynchronize());
    printf("{\"kernels\": {\n");
    const int reps = 3;
    auto report = [&](const char* name, double bytes, const char* kind, bool last = false) {
        printf(" \"%s\": {\"known_bytes\": %.0f, \"kind\": \"%s\"}%s\n", name, bytes, kind, last ? "" : ",");
    };
    // ---- reads ----
    {
        const long nvec = (long)(cap / 16);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_stream_read, dim3(256 * 16), dim3(256), 0, 0, (const half8*)buf, nvec, sink);
        report("calib_stream_read", (double)cap, "read");
    }
#define PIX(KS_, N_, H_, W_) { \
        const long M = (long)(N_) * (H_) * (W_); const double bytes = (double)M * 32 * (KS_); \
        if (bytes > cap) { printf("arena too small\n"); return 1; } \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_pix_read<KS_>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, (const half_t*)buf, M, sink); \
        report("calib_pix_read<" #KS_ ">", bytes, "read"); \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_pix_read3x3<KS_, false>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, (const half_t*)buf, N_, H_, W_, sink); \
        report("calib_pix_read3x3<" #KS_ ", false>", bytes, "read"); \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_pix_read3x3<KS_, true>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, (const half_t*)buf, N_, H_, W_, sink); \
        report("calib_pix_read3x3<" #KS_ ", true>", bytes, "read"); }
    PIX(1, 128, 272, 480)      // 16 channels (V4_ch_det_fast unit 1 at 272 x 480; 2 x the batch: 0.53 GB)
    PIX(2, 64, 272, 480)       // 32 channels
    PIX(3, 192, 136, 240)      // 48 channels at 136 x 240 (3 x the batch: 0.6 GB)
    PIX(6, 128, 136, 240)      // 96 channels (0.8 GB)
#define DWROW(C8_, N_, H_, W_) { \
        const double bytes = (double)(N_) * (H_) * (W_) * 16 * (C8_); \
        if (bytes > cap) { printf("arena too small\n"); return 1; } \
        const long total = (long)(N_) * (H_) * (((W_) + 3) / 4) * (C8_); \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_dwrow_read<C8_, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (const half_t*)buf, N_, H_, W_, sink); \
        report("calib_dwrow_read<" #C8_ ", false>", bytes, "read"); \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_dwrow_read<C8_, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (const half_t*)buf, N_, H_, W_, sink); \
        report("calib_dwrow_read<" #C8_ ", true>", bytes, "read"); }
    DWROW(2, 128, 272, 480)
    DWROW(6, 192, 136, 240)
    DWROW(24, 256, 34, 60)     // 192 channels at 34 x 60 (the deep blocks; 4 x the batch: 0.2 GB — inside the Infinity Cache on purpose)
    DWROW(32, 64, 136, 240)    // 256 channels at 136 x 240 (the server nets stage transitions: 1.07 GB)
    // ---- writes ----
    {
        const long nvec = (long)(cap / 16);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_stream_write, dim3(256 * 16), dim3(256), 0, 0, (half8*)buf, nvec);
        report("calib_stream_write", (double)cap, "write");
    }
#define PIXW(C8_, N_, H_, W_) { \
        const long M = (long)(N_) * (H_) * (W_); const double bytes = (double)M * 16 * (C8_); \
        if (bytes > cap) { printf("arena too small\n"); return 1; } \
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((calib_pix_write<C8_>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, (half_t*)buf, M); \
        report("calib_pix_write<" #C8_ ">", bytes, "write"); }
    PIXW(4, 64, 272, 480)      // 32 couts
    PIXW(6, 192, 136, 240)     // 48 couts
    PIXW(12, 128, 136, 240)    // 96 couts
    PIXW(16, 64, 136, 240)     // 128 couts (the server detector's 3 x 3 layers)
    CK(hipDeviceSynchronize());
    printf(" \"_end\": {}\n}}\n");
    return 0;
}
