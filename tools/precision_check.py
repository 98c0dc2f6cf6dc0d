#!/usr/bin/env python3
"""Error of the fused mel kernel against the float64 oracle for both filterbank precisions (development aid)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, REPO)
import numpy as np
import kapre_oracle as o, kapre_amd as kapre
from kapre_amd import _ffi
for kw, shape in ((dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128), (16, 44100, 1)),
                  (dict(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=80), (8, 48000, 1))):
    x = np.random.default_rng(5).uniform(-1, 1, shape).astype(np.float32)
    want = o.kapre_melspectrogram(x, **kw)
    for prec in (0, 1):
        _ffi.set_option("mel_precision", prec)
        got = kapre.get_melspectrogram_layer(**kw)(x).cpu().numpy()
        rel_each = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        print(kw["n_fft"], "precision", prec, "max err / max %.3g" % (np.abs(got - want).max() / want.max()),
              "max elementwise rel %.3g" % rel_each.max(), "rms rel %.3g" % np.sqrt((rel_each ** 2).mean()))
_ffi.set_option("mel_precision", 1)
