#!/usr/bin/env python3
"""One STFT configuration, a few launches (for rocprofv3 --pmc passes).  usage: tools/pmc_stft.py [n_fft hop batch T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kapre_amd as kapre
n_fft, hop, b, t = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (2048, 512, 256, 44100)))
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
st = kapre.STFT(n_fft=n_fft, hop_length=hop, pad_begin=False, pad_end=False, window_name="hann_window")
for _ in range(5):
    s = st(x)
torch.cuda.synchronize()
print(tuple(s.shape))
