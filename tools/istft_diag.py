#!/usr/bin/env python3
"""Where k_istft_pw differs from the barrier kernel: per hop block, for one shape (development aid).
    python tools/istft_diag.py FRAMES N_FFT HOP [WIN [BATCH]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kapre_amd import InverseSTFT, _ffi

frames, n_fft, hop = (int(a) for a in sys.argv[1:4])
win = int(sys.argv[4]) if len(sys.argv) > 4 else n_fft
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 2
rng = np.random.default_rng(1)
k = n_fft // 2 + 1
s = (rng.standard_normal((batch, 1, frames, k)) + 1j * rng.standard_normal((batch, 1, frames, k))).astype(np.complex64)
kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name="hann_window",
          input_data_format="channels_first", output_data_format="channels_first")
_ffi.set_option("verbose", 1)
_ffi.set_option("istft_path", 4)
got = InverseSTFT(**kw)(s).cpu().numpy()
print("launches:", _ffi.last_launches())
_ffi.set_option("istft_path", 1)
ref = InverseSTFT(**kw)(s).cpu().numpy()
d = np.abs(got - ref)[:, 0]
scale = np.abs(ref).max()
t_out = d.shape[1]
nb = (t_out + hop - 1) // hop
pad = np.zeros((batch, nb * hop), np.float32)
pad[:, :t_out] = d
blk = pad.reshape(batch, nb, hop).max(axis=2) / scale
for b in range(batch):
    bad = np.nonzero(blk[b] > 1e-5)[0]
    print("signal %d: max rel err %.3g, %d bad hop blocks of %d: %s" % (b, blk[b].max(), len(bad), nb, bad[:60]))
if len(sys.argv) > 6:
    np.set_printoptions(precision=5, linewidth=200)
    for t0 in (0, 64, 128, 200, 1024, 5000):
        print("t =", t0, "got", got[0, 0, t0:t0 + 6], "ref", ref[0, 0, t0:t0 + 6], "ratio", got[0, 0, t0:t0 + 6] / ref[0, 0, t0:t0 + 6])
    bad = np.nonzero(np.abs(got[0, 0] - ref[0, 0]) > 1e-6 * scale)[0]
    print("wrong samples:", len(bad), "positions mod hop:", sorted(set((bad % hop).tolist()))[:80])
    print("first wrong:", bad[:40])
    blocks = bad // hop
    print("wrong samples per block (first 40 blocks):", np.bincount(blocks, minlength=40)[:40])
    import torch
    t0 = 8 * hop
    dd = (got - ref)[0, 0, t0:t0 + 4]
    print("diff at t0..t0+3:", dd)
    for f in range(0, 12):
        s1 = np.zeros_like(s)
        s1[:, :, f] = s[:, :, f]
        of = InverseSTFT(**kw)(s1).cpu().numpy()[0, 0, t0:t0 + 4]
        print("  frame %2d contributes" % f, of)
