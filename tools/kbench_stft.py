#!/usr/bin/env python3
"""Kernel-time table for the stand-alone STFT / ISTFT kernels (HBM-bound stage of the path)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import kapre_amd as kapre

def time_graph(fn, launches=50):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(launches): fn()
    torch.cuda.synchronize(); graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches

cases = [
    ("cfg4 fwd  b128 x 110250, n_fft 1024 hop 256 (pad_begin+pad_end), complex out", dict(b=128, t=110250, n_fft=1024, hop=256, pads=True)),
    ("cfg1-like b256 x 16000,  n_fft 512 hop 256, complex out", dict(b=256, t=16000, n_fft=512, hop=256, pads=False)),
    ("b256 x 44100, n_fft 2048 hop 512, complex out", dict(b=256, t=44100, n_fft=2048, hop=512, pads=False)),
]
for name, c in cases:
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (c["b"], c["t"], 1)).astype(np.float32)).cuda()
    st = kapre.STFT(n_fft=c["n_fft"], hop_length=c["hop"], pad_begin=c["pads"], pad_end=c["pads"], window_name="hann_window")
    s = st(x)
    frames = s.shape[0] * s.shape[1]
    k = c["n_fft"] // 2 + 1
    us = time_graph(lambda: st(x))
    byt = x.numel() * 4 + s.numel() * 8
    print("STFT  %-78s %8.1f us  %7.1f Mframes/s  %6.0f GB/s (%.2f of 8 TB/s)" % (name, us, frames / us, byt / us / 1e3, byt / us / 1e3 / 8000))
    mag = kapre.Sequential([st, kapre.Magnitude()])
    us = time_graph(lambda: mag(x))
    byt = x.numel() * 4 + s.numel() * 4
    print("STFT+|.| %-75s %8.1f us  %7.1f Mframes/s  %6.0f GB/s (%.2f)" % ("", us, frames / us, byt / us / 1e3, byt / us / 1e3 / 8000))
    ist = kapre.InverseSTFT(n_fft=c["n_fft"], hop_length=c["hop"], forward_window_name="hann_window")
    y = ist(s)
    us = time_graph(lambda: ist(s))
    byt = s.numel() * 8 + y.numel() * 4
    print("ISTFT %-78s %8.1f us  %7.1f Mframes/s  %6.0f GB/s algorithmic (%.2f)" % ("", us, frames / us, byt / us / 1e3, byt / us / 1e3 / 8000))
