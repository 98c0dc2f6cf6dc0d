#!/usr/bin/env python3
"""Kernel times for transform sizes that are not powers of two (Bluestein STFT) next to their
power-of-two neighbours, cfg5-sized input (development aid / numbers for BASELINE.md)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
from tools.kbench_row4 import timeit

rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-1, 1, (256, 160000, 1)).astype(np.float32)).cuda()
for n_fft, hop in ((400, 160), (512, 160), (1000, 250), (1024, 250), (480, 120), (960, 240), (300, 75)):
    mel = kapre.composed.get_melspectrogram_layer(n_fft=n_fft, hop_length=hop, sample_rate=16000, n_mels=80)
    st = kapre.STFT(n_fft=n_fft, hop_length=hop)
    frames = mel(x).shape[1] * 256
    t_mel, t_st = timeit(lambda: mel(x), 20), timeit(lambda: st(x), 20)
    print("n_fft %4d hop %3d: mel %8.1f us (%7.1f Mframes/s)   stft %8.1f us (%7.1f Mframes/s)" %
          (n_fft, hop, t_mel, frames / t_mel, t_st, frames / t_st))

# inverse STFT of the same sizes (perfect-reconstruction pair, hann, 75 % overlap)
xs = x[:64]
for n_fft in (400, 512, 1000, 1024):
    stft, istft = kapre.composed.get_perfectly_reconstructing_stft_istft(n_fft, n_fft // 4, "channels_last", "channels_last")
    s = stft(xs)
    frames = s.shape[1] * xs.shape[0]
    t_i = timeit(lambda: istft(s), 20)
    print("n_fft %4d hop %3d: istft %8.1f us (%7.1f Mframes/s)" % (n_fft, n_fft // 4, t_i, frames / t_i))
