"""CPU checks of the size-generic FFT engine's formulas (oracle/proto_generic_fft.py mirrors
kapre_amd/csrc/kpr_generic_kernels.h): plan, pass indexing, twiddle indices, even-size packing, float-reciprocal division."""
import numpy as np
import pytest

import proto_generic_fft as pg


def test_plan_covers_exactly_the_sizes_with_small_prime_factors():
    for n in range(2, 3000):
        plan = pg.gen_plan(n)
        m, big = n, 1
        f = 2
        while f * f <= m:
            while m % f == 0:
                big, m = max(big, f), m // f
            f += 1
        big = max(big, m) if m > 1 else big
        assert (plan is not None) == (big <= 64), n
        if plan:
            assert int(np.prod(plan)) == n and plan.count(2) <= 1


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 12, 15, 20, 77, 100, 105, 225, 600, 768, 1000, 1001, 1155, 1500, 2048, 3000])
def test_complex_fft_matches_numpy_in_both_directions(n):
    rng = np.random.default_rng(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    tw = np.exp(-2j * np.pi * np.arange(n) / n)
    plan = pg.gen_plan(n)
    np.testing.assert_allclose(pg.gen_fft(z, plan, tw, 1, -1), np.fft.fft(z), rtol=0, atol=1e-9 * n)
    np.testing.assert_allclose(pg.gen_fft(z, plan, tw, 1, +1), np.fft.ifft(z) * n, rtol=0, atol=1e-9 * n)
    # half-length transform on the full-length table (stride 2), as the packed even sizes use it
    if n % 2 == 0 and pg.gen_plan(n // 2):
        h = z[: n // 2]
        np.testing.assert_allclose(pg.gen_fft(h, pg.gen_plan(n // 2), tw, 2, -1), np.fft.fft(h), rtol=0, atol=1e-9 * n)


@pytest.mark.parametrize("n_fft", [4, 6, 12, 15, 30, 77, 1000, 1001, 1200, 1280, 1536, 2000, 3000, 6000, 2049])
def test_real_transforms_match_numpy(n_fft):
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft)
    want = np.fft.rfft(x)
    np.testing.assert_allclose(pg.rfft_generic(x, n_fft), want, rtol=0, atol=1e-9 * n_fft)
    s = rng.standard_normal(n_fft // 2 + 1) + 1j * rng.standard_normal(n_fft // 2 + 1)     # NOT a real signal's spectrum
    np.testing.assert_allclose(pg.irfft_generic(s, n_fft), np.fft.irfft(s, n=n_fft), rtol=0, atol=1e-10)
    np.testing.assert_allclose(pg.irfft_generic(want, n_fft), x, rtol=0, atol=1e-10)


def test_float_reciprocal_division_is_exact_for_every_index_the_kernels_form():
    """o / d via (int)((o + 0.5f) * (1.0f / d)): every o < 20480 (the largest FFT length that fits in LDS) and every
    divisor a pass can have."""
    o = np.arange(20480)
    for d in list(range(1, 4097)) + [5000, 6000, 8192, 10240, 20480]:
        assert (pg.float_div(o, d) == o // d).all(), d
