#!/usr/bin/env python3
"""Stand-alone ApplyFilterbank (+ Magnitude, MagnitudeToDecibel) by layout and channel count: kernel time (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kapre_amd import ApplyFilterbank, Magnitude, MagnitudeToDecibel, _ffi

k, n_mels, frames = 1025, 128, 83
for ch, batch in ((1, 256), (2, 128), (4, 64)):
    for fmt in ("channels_first", "channels_last"):
        shape = (batch, frames, k, ch) if fmt == "channels_last" else (batch, ch, frames, k)
        x = torch.rand(shape, device="cuda")
        fb = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=n_mels), data_format=fmt)
        us, _ = bench.kernel_time_us(fb, x, launches=50, settle_s=0.5)
        lab = _ffi.last_launches()
        db = MagnitudeToDecibel()
        us2, _ = bench.kernel_time_us(db, x, launches=50, settle_s=0.3)
        print("C=%d %-14s filterbank %8.2f us [%s]   decibel %8.2f us [%s]" % (ch, fmt, us, lab, us2, _ffi.last_launches()))
