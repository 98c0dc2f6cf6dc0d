import sys, os
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi, STFT, Magnitude, ApplyFilterbank, Sequential
def run(B, T=44100, n_fft=2048, hop=512, M=128, sr=44100):
    K = n_fft // 2 + 1
    x = np.random.default_rng(1).uniform(-1, 1, (B, T, 1)).astype(np.float32)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=K, n_mels=M))
    fb = np.array(layer.filterbank)
    st = STFT(n_fft=n_fft, hop_length=hop)
    mag = Sequential([st, Magnitude()])(x).cpu().numpy().astype(np.float64)
    want = np.einsum("bfkc,km->bfmc", mag, fb.astype(np.float64))
    model = Sequential([st, Magnitude(), layer])
    got = model(x).cpu().numpy()
    got2 = model(x).cpu().numpy()
    F = want.shape[1]
    err = (np.abs(got - want) / np.maximum(np.abs(want), 1e-30)).reshape(B * F, M)
    badf = np.nonzero((err > 1e-3).any(axis=1))[0]
    total = B * F
    grid = min((total + 7) // 8, 256)
    # position of each bad frame inside its workgroup's run
    pos = []
    for g in badf[:4000]:
        b = min(int(g * grid // total), grid - 1)
        while (total * b // grid) > g: b -= 1
        while (total * (b + 1) // grid) <= g: b += 1
        pos.append(int(g - total * b // grid))
    print("B", B, "frames", total, "per WG %.2f" % (total / grid), "max rel %.3g" % err.max(), "bad frames", len(badf),
          "positions in WG run:", sorted(set(pos))[:40], "deterministic", bool((got == got2).all()),
          "bad mels", sorted(set(np.nonzero((err > 1e-3).any(axis=0))[0].tolist()))[:12])
for B in (32, 64, 100, 100):
    run(B)
