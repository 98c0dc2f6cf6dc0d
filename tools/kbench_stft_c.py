#!/usr/bin/env python3
"""Kernel time of the stand-alone STFT, complex output only (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre

def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for (b, t, n_fft, hop) in ((256, 44100, 2048, 512), (128, 110250, 1024, 256), (256, 16000, 512, 256)):
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
    st = kapre.STFT(n_fft=n_fft, hop_length=hop, pad_begin=False, pad_end=False)
    us = timeit(lambda: st(x))
    frames = b * (1 + (t - n_fft) // hop)
    byts = frames * (4 * hop + 8 * (n_fft // 2 + 1))
    print("STFT complex b%d x %d n_fft %d hop %d: %7.1f us  %7.1f Mframes/s  %5.0f GB/s" % (b, t, n_fft, hop, us, frames / us, byts / us / 1e3))
