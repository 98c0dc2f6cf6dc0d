"""The CPU baseline graphs used by bench.py's cpu_baseline leg agree with the float64 oracle."""
import numpy as np
import pytest

import cpu_graph
import kapre_oracle as o
from conftest import rel_err


@pytest.mark.parametrize("variant", ["scipy", "torch", "pooled"])
@pytest.mark.parametrize("db", [None, (1.0, 1e-5, 80.0)])
def test_cpu_graph_matches_oracle(variant, db):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (2, 9000, 2)).astype(np.float32)
    n_fft, hop, sr, m = 1024, 256, 22050, 64
    window = o.hann_window(n_fft).astype(np.float32)
    fb = o.filterbank_mel(sr, n_fft // 2 + 1, m)
    fn = {"scipy": cpu_graph.melspectrogram_scipy, "torch": cpu_graph.melspectrogram_torch,
          "pooled": lambda *a: cpu_graph.melspectrogram_pooled(*a, workers=3)}[variant]
    got = fn(x, window, fb, n_fft, hop, db)
    want = o.kapre_melspectrogram(x, n_fft=n_fft, hop_length=hop, sample_rate=sr, n_mels=m,
                                  return_decibel=db is not None)
    assert got.shape == want.shape and got.dtype == np.float32
    if db is None:
        assert rel_err(got, want) < 1e-5
    else:
        assert np.abs(got - want).max() < 1e-2
