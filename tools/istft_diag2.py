#!/usr/bin/env python3
"""k_istft_pw vs the barrier kernel for ONE non-zero frame (development aid): python tools/istft_diag2.py FRAMES N_FFT HOP FRAME"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kapre_amd import InverseSTFT, _ffi
frames, n_fft, hop, f = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(1)
k = n_fft // 2 + 1
s = np.zeros((1, 1, frames, k), np.complex64)
s[0, 0, f] = (rng.standard_normal(k) + 1j * rng.standard_normal(k)).astype(np.complex64)
kw = dict(n_fft=n_fft, win_length=n_fft, hop_length=hop, forward_window_name="hann_window",
          input_data_format="channels_first", output_data_format="channels_first")
_ffi.set_option("istft_path", 4)
got = InverseSTFT(**kw)(s).cpu().numpy()[0, 0]
_ffi.set_option("istft_path", 1)
ref = InverseSTFT(**kw)(s).cpu().numpy()[0, 0]
np.set_printoptions(precision=5, linewidth=220, suppress=True)
t0 = f * hop
L2 = n_fft // 16          # samples per register slot
print("nonzero outside the frame (got):", np.abs(got[:t0]).max() if t0 else 0.0, np.abs(got[t0 + n_fft:]).max())
for m in range(16):
    a, b = got[t0 + L2 * m: t0 + L2 * m + 6], ref[t0 + L2 * m: t0 + L2 * m + 6]
    print("slot %2d got %s ref %s" % (m, a, b))
bad = np.nonzero(np.abs(got - ref) > 1e-6 * np.abs(ref).max())[0]
print("wrong samples:", len(bad), "relative to frame start:", (bad - t0)[:64])
