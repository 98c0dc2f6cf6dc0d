#!/usr/bin/env python3
"""VERDICT r02 item 3 figures: short-window fetch (win_length 2018 vs 2048 on the target shape) and the stand-alone
ApplyFilterbank with six channels in both layouts (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
import kapre_amd as kapre
from tools.kbench_row4 import timeit  # noqa

name = bench.DEFAULT
for win in (None, 2018, 1024):
    w = dict(bench.WORKLOADS[name]); w["win"] = win
    model = kapre.composed.get_melspectrogram_layer(input_shape=(w["t"], 1), n_fft=2048, win_length=win, hop_length=512,
                                                    sample_rate=44100, n_mels=128)
    x = bench.make_input(w, 0, torch.device("cuda", 0), w["batch"])
    us, how = bench.kernel_time_us(model, x, launches=100)
    print("target shape, win_length %s: %.2f us" % (win, us))
rng = np.random.default_rng(0)
b, f, k, m, c = 256, 44, 1025, 128, 6
for fmt in ("channels_first", "channels_last"):
    shape = (b, c, f, k) if fmt == "channels_first" else (b, f, k, c)
    mag = torch.from_numpy(rng.uniform(0, 1, shape).astype(np.float32)).cuda()
    fb = kapre.ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=m), data_format=fmt)
    us = timeit(lambda: fb(mag))
    print("ApplyFilterbank 256x6x44 frames 1025 -> 128, %s: %.1f us (%.2f of 8 TB/s)" % (fmt, us, (4 * mag.numel() * (1 + m / k)) / us / 1e3 / 8000))
