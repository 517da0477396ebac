#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box: torch copy / add over fp16 tensors of the sizes the mobile detector's ops move
(yardstick for the HBM-bound rows of roofline.kernels; bytes = read + written)."""
import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for mb in (64, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.float16, device="cuda").normal_()
    y = torch.empty_like(x)
    t = timed(lambda: y.copy_(x))
    t2 = timed(lambda: torch.add(x, 1.0, out=y))
    t3 = timed(lambda: x.sum())
    t4 = timed(lambda: y.fill_(1.0))
    print(f"{mb:5d} MB tensor: copy {2 * mb / 1024 / t * 1e3 / 1e3:.2f} TB/s ({t * 1e3:.0f} us), add {2 * mb / 1024 / t2:.2f} TB/s, "
          f"read-only sum {mb / 1024 / t3:.2f} TB/s, write-only fill {mb / 1024 / t4:.2f} TB/s")
