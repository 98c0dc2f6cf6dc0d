import sys, os
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi, STFT, Magnitude, ApplyFilterbank, Sequential
def run(n_fft, hop, T):
    K = n_fft // 2 + 1
    M = (K - 1) // 8
    x = np.random.default_rng(1).uniform(-1, 1, (1, T, 1)).astype(np.float32)
    st = STFT(n_fft=n_fft, hop_length=hop)
    mag = Sequential([st, Magnitude()])(x).cpu().numpy().astype(np.float64)[0, :, :, 0]     # (F, K)
    badbins = {}
    for o in range(9):
        fb = np.zeros((K, M), np.float32)
        for m in range(M):
            k = 8 * m + o
            if k < K: fb[k, m] = 1.0
        layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=K, n_mels=M))
        layer.filterbank = fb
        got = Sequential([st, Magnitude(), layer])(x).cpu().numpy()[0, :, :, 0]       # (F, M)
        for m in range(M):
            k = 8 * m + o
            if k >= K: continue
            err = np.abs(got[:, m] - mag[:, k]) / np.maximum(mag[:, k], 1e-20)
            if err.max() > 1e-3:
                badbins[k] = (float(err.max()), int(np.argmax(err)), float(got[np.argmax(err), m]), float(mag[np.argmax(err), k]))
    print(n_fft, "bad bins:", sorted(badbins))
    for k in sorted(badbins)[:12]:
        print("   bin", k, "max rel %.3g frame %d got %.5g want %.5g" % badbins[k])
run(1024, 256, 6000)
run(2048, 512, 12000)
