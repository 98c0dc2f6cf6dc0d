#!/usr/bin/env python3
"""InverseSTFT with several channels: kernel time by layout (development aid).  python tools/istft_cl_check.py [n_fft hop ch batch frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kapre_amd import InverseSTFT, _ffi

n_fft, hop, ch, batch, frames = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (1024, 256, 2, 64, 434)))
k = n_fft // 2 + 1
for fi in ("channels_first", "channels_last"):
    for fo in ("channels_first", "channels_last"):
        shape = (batch, frames, k, ch) if fi == "channels_last" else (batch, ch, frames, k)
        s = torch.randn(shape, dtype=torch.complex64, device="cuda")
        m = InverseSTFT(n_fft=n_fft, hop_length=hop, input_data_format=fi, output_data_format=fo)
        us, _ = bench.kernel_time_us(m, s, launches=50, settle_s=0.5)
        print("in %-14s out %-14s %8.2f us  [%s]" % (fi, fo, us, _ffi.last_launches()))
