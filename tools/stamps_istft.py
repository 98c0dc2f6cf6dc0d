#!/usr/bin/env python3
"""Print in-kernel cycle stamps of workgroup 0 of k_istft_ws (development aid): producer waves stamp
each finished ticket, the consumer wave each emitted pass."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import kapre_amd as kapre
from kapre_amd import _ffi
b, t, n_fft, hop = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 110250, 1024, 256))]
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
st = kapre.STFT(n_fft=n_fft, hop_length=hop, window_name="hann_window")
ist = kapre.InverseSTFT(n_fft=n_fft, hop_length=hop, forward_window_name="hann_window")
s = st(x); ist(s); torch.cuda.synchronize()
NW = 8
buf = torch.zeros(NW * 32 + 1, dtype=torch.int64, device="cuda")
L = _ffi.lib()
L.kpr_debug_stamps.argtypes = [ctypes.c_void_p]
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
ist(s); torch.cuda.synchronize()
L.kpr_debug_stamps(ctypes.c_void_p(0))
d = buf.cpu().numpy()[:NW * 32].reshape(NW, 32)
t0 = d[:, 0][d[:, 0] != 0].min()
for wv in range(NW):
    row = d[wv]; n = int((row != 0).sum())
    if n == 0: continue
    print("wave", wv, " ".join("%7d" % (v - t0) for v in row[:n]))
    print("  delta", " ".join("%7d" % v for v in (row[1:n] - row[:n - 1])))
