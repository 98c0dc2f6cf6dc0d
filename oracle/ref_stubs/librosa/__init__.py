"""librosa stand-in exposing only what kapre/backend.py calls (filters.mel, fft_frequencies,
util.normalize); arithmetic lives in oracle/kapre_oracle.py.  Test infrastructure only."""
import os
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, "..", "..")))
import kapre_oracle as _o  # noqa: E402

filters = types.ModuleType("librosa.filters")


def _mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney",
         dtype=np.float32):
    fb = _o.filterbank_mel(sr, n_fft // 2 + 1, n_mels, fmin, fmax, htk, norm)  # (K, M)
    return np.ascontiguousarray(fb.T).astype(dtype)


filters.mel = _mel


def fft_frequencies(*, sr=22050, n_fft=2048):
    return np.arange(n_fft // 2 + 1, dtype=np.float64) * (float(sr) / n_fft)


util = types.ModuleType("librosa.util")


def _normalize(S, *, norm=np.inf, axis=0):
    mag = np.abs(S).astype(float)
    if norm == np.inf:
        length = np.max(mag, axis=axis, keepdims=True)
    else:
        length = np.sum(mag ** norm, axis=axis, keepdims=True) ** (1.0 / norm)
    length[length < np.finfo(mag.dtype).tiny] = 1.0
    return S / length


util.normalize = _normalize
sys.modules[__name__ + ".filters"] = filters
sys.modules[__name__ + ".util"] = util
