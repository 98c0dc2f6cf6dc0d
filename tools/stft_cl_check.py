#!/usr/bin/env python3
"""STFT with several channels: kernel time by input / output layout (development aid).  python tools/stft_cl_check.py [n_fft hop ch batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kapre_amd import STFT, Magnitude, Sequential, _ffi

n_fft, hop, ch, batch = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (1024, 256, 4, 32)))
t = 110250
for mag in (False, True):
    for fi in ("channels_first", "channels_last"):
        for fo in ("channels_first", "channels_last"):
            shape = (batch, t, ch) if fi == "channels_last" else (batch, ch, t)
            x = torch.rand(shape, device="cuda") * 2 - 1
            st = STFT(n_fft=n_fft, hop_length=hop, input_data_format=fi, output_data_format=fo)
            model = Sequential([st, Magnitude()]) if mag else st
            us, _ = bench.kernel_time_us(model, x, launches=50, settle_s=0.5)
            print("%-9s in %-14s out %-14s %8.2f us  [%s]" % ("magnitude" if mag else "complex", fi, fo, us, _ffi.last_launches()))
