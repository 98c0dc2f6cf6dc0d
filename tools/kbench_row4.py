#!/usr/bin/env python3
"""Kernel times of the SURVEY 8f row-4 layers (Frame, Energy, Delta, LogmelToMFCC) on cfg5-sized
inputs, with the algorithmic HBM traffic of each (development aid / numbers for BASELINE.md)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre


def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    rng = np.random.default_rng(0)
    b, t = 256, 160000                                   # cfg5 per-GPU batch: 10 s @ 16 kHz
    x = torch.from_numpy(rng.uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
    frames = 1 + (t - 400) // 160
    lm = torch.from_numpy((rng.uniform(-1, 1, (b, frames, 80, 1)) * 30 - 40).astype(np.float32)).cuda()
    rows = []
    fr = kapre.Frame(400, 160)
    us = timeit(lambda: fr(x)); rows.append(("Frame 400/160", us, 4 * b * t + 4 * b * frames * 400))
    en = kapre.Energy(16000, 0.05, 400, 160)
    us = timeit(lambda: en(x)); rows.append(("Energy 400/160", us, 4 * b * t + 4 * b * frames))
    dl = kapre.Delta(9)
    us = timeit(lambda: dl(lm)); rows.append(("Delta win 9 (80 bands)", us, 2 * 4 * lm.numel()))
    mf = kapre.LogmelToMFCC(13)
    us = timeit(lambda: mf(lm)); rows.append(("LogmelToMFCC 80 -> 13", us, 4 * lm.numel() + 4 * b * frames * 13))
    for name, us, byts in rows:
        print("%-26s %8.1f us   %7.1f Mframes/s   %6.0f GB/s algorithmic (%.2f of 8 TB/s)" %
              (name, us, b * frames / us, byts / us / 1e3, byts / us / 1e3 / 8000))


if __name__ == "__main__":
    main()
