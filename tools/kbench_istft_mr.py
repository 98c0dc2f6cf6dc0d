#!/usr/bin/env python3
"""InverseSTFT kernel time at the mixed-radix sizes (ring kernel k_istft_ws_mr), 64 x 10 s @ 16 kHz, hop n_fft / 4 (development aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
import bench
from kapre_amd import _ffi

rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-1, 1, (64, 160000, 1)).astype(np.float32)).cuda()
for n_fft in (160, 200, 320, 400, 640, 800, 1000, 480, 960, 1024):
    stft, istft = kapre.composed.get_perfectly_reconstructing_stft_istft(n_fft, n_fft // 4, "channels_last", "channels_last")
    s = stft(x)
    us, _ = bench.kernel_time_us(istft, s, launches=50)
    print("n_fft %4d hop %3d: istft %8.1f us  [%s]" % (n_fft, n_fft // 4, us, _ffi.last_launches()), flush=True)
