import sys, os
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi, STFT, Magnitude, ApplyFilterbank, Sequential
def run(n_fft, hop, M, B, T, fbmode, sr=44100):
    K = n_fft // 2 + 1
    x = np.random.default_rng(1).uniform(-1, 1, (B, T, 1)).astype(np.float32)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=K, n_mels=M))
    fb = np.array(layer.filterbank)
    if fbmode == "ones":
        fb = (fb != 0).astype(np.float32)
    layer.filterbank = fb.copy()
    st = STFT(n_fft=n_fft, hop_length=hop)
    mag = Sequential([st, Magnitude()])(x).cpu().numpy().astype(np.float64)
    want = np.einsum("bfkc,km->bfmc", mag, fb.astype(np.float64))
    got = Sequential([st, Magnitude(), layer])(x).cpu().numpy()
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    bad = np.argwhere(err > 1e-3)
    kr = _ffi.filterbank_kranges(fb).reshape(-1, 2)
    print(n_fft, fbmode, "B", B, "T", T, "sr", sr, "max rel %.3g" % err.max(), "n bad", len(bad), "of", err.size,
          "bad mels", sorted(set(bad[:, 2].tolist()))[:40], "bad frames", sorted(set(bad[:, 1].tolist()))[:6], "bad items", sorted(set(bad[:, 0].tolist()))[:6])
    if len(bad): print("   kranges", kr.tolist())
run(2048, 512, 128, 1, 12000, "mel")
run(2048, 512, 128, 1, 12000, "ones")
run(2048, 512, 128, 1, 12000, "mel", 22050)
run(2048, 512, 128, 16, 44100, "mel")
run(2048, 512, 64, 1, 12000, "mel")
run(2048, 512, 96, 1, 12000, "mel")
