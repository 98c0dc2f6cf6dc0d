#!/usr/bin/env python3
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (256, 44100, 1)).astype(np.float32)).cuda()
st = kapre.STFT(n_fft=2048, hop_length=512)
st(x); torch.cuda.synchronize()
buf = torch.zeros(4 * 32, dtype=torch.int64, device="cuda")
L = _ffi.lib(); L.kpr_debug_stamps.argtypes = [ctypes.c_void_p]
L.kpr_debug_stamps(ctypes.c_void_p(buf.data_ptr())); st(x); torch.cuda.synchronize(); L.kpr_debug_stamps(ctypes.c_void_p(0))
b = buf.cpu().numpy().reshape(4, 32); t0 = b[:, 0].min()
for wv in range(2):
    row = b[wv]; n = int((row != 0).sum())
    print("wave", wv, " ".join("%7d" % (v - t0) for v in row[:n]))
    print("  delta", " ".join("%7d" % d for d in (row[1:n] - row[:n-1])))
