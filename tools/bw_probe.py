#!/usr/bin/env python3
"""HBM bandwidth calibration with plain torch kernels (write-only, read-only-ish, copy)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (174, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    s = t(lambda: a.fill_(1.0)); print("fill  %5d MB: %6.1f us  %.2f TB/s (write only)" % (mb, s * 1e6, mb * 1.048576e6 / s / 1e12))
    s = t(lambda: b.copy_(a));   print("copy  %5d MB: %6.1f us  %.2f TB/s (read+write)" % (mb, s * 1e6, 2 * mb * 1.048576e6 / s / 1e12))
    s = t(lambda: a.sum());      print("sum   %5d MB: %6.1f us  %.2f TB/s (read only)" % (mb, s * 1e6, mb * 1.048576e6 / s / 1e12))
