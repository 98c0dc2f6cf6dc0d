#!/usr/bin/env python3
"""STFT / InverseSTFT time over transform sizes on one workload (64 x 44100 x 1, hop = n_fft / 4): shows which n_fft are
off the FFT curve (development aid).   python tools/nfft_sweep.py [n_fft ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kapre_amd import STFT, InverseSTFT, _ffi


def time_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


if __name__ == "__main__":
    for a in [a for a in sys.argv[1:] if "=" in a]:          # name=value -> kpr_set_option
        _ffi.set_option(a.split("=")[0], int(a.split("=")[1]))
        print("option", a)
    sys.argv = [a for a in sys.argv if "=" not in a]
    sizes = [int(a) for a in sys.argv[1:]] or [512, 1000, 1001, 1024, 1200, 1280, 1536, 2000, 2048, 2049, 3000, 4096]
    x = torch.randn(64, 44100, 1, device="cuda")
    for n_fft in sizes:
        hop = max(1, n_fft // 4)
        st = STFT(n_fft=n_fft, hop_length=hop)
        y = st(x)
        ist = InverseSTFT(n_fft=n_fft, hop_length=hop)
        frames = y.shape[1] * 64
        us_f = time_us(lambda: st(x))
        us_i = time_us(lambda: ist(y))
        print("n_fft %5d  plan %d  frames %6d  stft %9.1f us (%6.2f ns/sample)  istft %9.1f us" % (
            n_fft, _ffi.lib().kpr_fft_plan(n_fft, n_fft), frames, us_f, us_f * 1e3 / (64 * 44100), us_i))
