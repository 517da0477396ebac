// OpenCV INTER_LINEAR for 8-bit images restated in integers: 11-bit fixed-point coefficients, horizontal pass in
// int32, vertical pass  ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Shared by det_preprocess_kernel (prepost.hip) and the
// stem kernel's fused pre-processing (conv_stem.hip, F_U8SRC): both produce the same bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct LinCoef { int s0; short a0, a1; };
__host__ __device__ inline LinCoef lin_coef(int d, int dst, int src) {
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    LinCoef c;
    c.s0 = s;
    // cv::saturate_cast<short>(float) = round half to even
    c.a0 = (short)rintf((1.f - f) * 2048.f);
    c.a1 = (short)rintf(f * 2048.f);
    return c;
}
__device__ __forceinline__ int cv_bilinear_u8(int p00, int p01, int p10, int p11, LinCoef cx, LinCoef cy) {
    const int r0 = p00 * cx.a0 + p01 * cx.a1;
    const int r1 = p10 * cx.a0 + p11 * cx.a1;
    return ((((int)cy.a0 * (r0 >> 4)) >> 16) + (((int)cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
}
