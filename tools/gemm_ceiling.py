#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt through torch.matmul) reaches on GEMMs of the detector's conv shapes: a practical
ceiling to read the conv kernels' TFLOP/s against (bench scaffolding, not on the product path)."""
import torch

SHAPES = [  # (name, M, N, K)
    ("3x3 128->128 @136x240 x64", 64 * 136 * 240, 128, 1152),
    ("3x3 64->128 @272x480 x64", 64 * 272 * 480, 128, 576),
    ("1x1 896->256 @136x240 x64", 64 * 136 * 240, 256, 896),
    ("9x9 256->64 @136x240 x64 (K/4)", 64 * 136 * 240, 64, 20736 // 4),
    ("3x3 160->160 @68x120 x64", 64 * 68 * 120, 160, 1440),
    ("square 8192", 8192, 8192, 8192),
]


def main():
    for name, m, n, k in SHAPES:
        a = (torch.randn((m, k), device="cuda") * 0.1).half()
        b = (torch.randn((k, n), device="cuda") * 0.1).half()
        for _ in range(3):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:36s} M={m} N={n} K={k}: {ms:.3f} ms  {2 * m * n * k / ms / 1e9:.0f} TFLOP/s  "
              f"(A+C bytes at {(m * k + m * n) * 2 / ms / 1e9:.2f} TB/s)")
        del a, b, c


if __name__ == "__main__":
    main()
