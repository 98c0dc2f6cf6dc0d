#!/usr/bin/env python3
"""Workload for the rocprofv3 passes that document the two stages north_star asks evidence for:
the FFT stage (stand-alone STFT, BASELINE cfg4 forward / inverse and the target shape) and the filterbank stage
(stand-alone ApplyFilterbank on the target shape's magnitudes).  A few launches of each."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre

rng = np.random.default_rng(0)
x4 = torch.from_numpy(rng.uniform(-1, 1, (128, 110250, 1)).astype(np.float32)).cuda()
st4, ist4 = kapre.composed.get_perfectly_reconstructing_stft_istft(1024, 256, "channels_last", "channels_last")
xt = torch.from_numpy(rng.uniform(-1, 1, (256, 44100, 1)).astype(np.float32)).cuda()
stt = kapre.STFT(n_fft=2048, hop_length=512)
mag = kapre.Sequential([kapre.STFT(n_fft=2048, hop_length=512), kapre.Magnitude()])(xt)
fb = kapre.ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=1025, n_mels=128))
for _ in range(5):
    s4 = st4(x4); s2 = stt(xt); m = fb(mag); y4 = ist4(s4)
torch.cuda.synchronize()
print("stft cfg4 frames", s4.shape[0] * s4.shape[1], "bytes", x4.numel() * 4 + s4.numel() * 8,
      "| stft target frames", s2.shape[0] * s2.shape[1], "bytes", xt.numel() * 4 + s2.numel() * 8,
      "| filterbank rows", mag.shape[0] * mag.shape[1], "bytes", mag.numel() * 4 + m.numel() * 4,
      "| istft cfg4 frames", s4.shape[0] * s4.shape[1], "bytes", s4.numel() * 8 + y4.numel() * 4)
