"""The lane-level numpy model of the HIP FFT kernels (oracle/proto_stockham.py) against numpy.fft."""
import numpy as np
import pytest

import proto_stockham as ps


@pytest.mark.parametrize("n_fft", [64, 128, 256, 512, 1024, 2048, 4096])
def test_rfft_lane_model(n_fft):
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft)
    np.testing.assert_allclose(ps.rfft_lanes(x), np.fft.rfft(x), atol=1e-11)


@pytest.mark.parametrize("n_fft", [256, 512, 1024, 2048])
def test_irfft_lane_model(n_fft):
    rng = np.random.default_rng(n_fft + 1)
    X = rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)
    np.testing.assert_allclose(ps.irfft_lanes(X), np.fft.irfft(X, n=n_fft), atol=1e-12)


def test_swizzle_is_a_bijection_and_linear():
    for nc in (128, 256, 512, 1024):
        e = np.arange(nc)
        assert sorted(ps.swz(e)) == list(range(nc))
    a, c = 37, 64 * 5          # bit-disjoint parts: swz(a + c) == swz(a) ^ swz(c)
    assert ps.swz(a + c) == ps.swz(a) ^ ps.swz(c)
