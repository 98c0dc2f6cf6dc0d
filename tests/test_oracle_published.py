"""Third-party L0 anchors for the oracle: numbers PUBLISHED by librosa / TensorFlow / HTK (docstring
examples, documented closed forms evaluated by hand) written out as literals -- nothing in this file is
computed by oracle/kapre_oracle.py or kapre_amd/backend.py and then compared with itself.

Why: the reference delegates this arithmetic to un-vendored packages (librosa >=0.11 `filters.mel`
behind /root/reference/kapre/backend.py:222-231, `fft_frequencies` / `util.normalize` behind
backend.py:284-296, `tf.signal.*_window` behind backend.py:76-87, `tf.signal.inverse_stft_window_fn`
behind time_frequency.py:278-280).  The golden fixtures pin Kapre's glue; THIS file pins the L0 layer
the glue calls.  Both the oracle and the product's host builders are held to the same literals.
"""
import numpy as np
import pytest
import scipy.signal
import scipy.stats

import kapre_oracle as o
from kapre_amd import backend as kb

# ---------------------------------------------------------------------------------------------
# librosa docstring examples (librosa.core.convert / librosa.filters, unchanged 0.8 ... 0.11)
# ---------------------------------------------------------------------------------------------
# >>> librosa.mel_frequencies(n_mels=40)
LIBROSA_MEL_FREQUENCIES_40 = [
    0.0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855,
    853.173, 938.49, 1024.856, 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532,
    1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731, 3512.582, 3835.643,
    4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272,
    9246.028, 10096.408, 11025.0]
# >>> librosa.fft_frequencies(sr=22050, n_fft=16)
LIBROSA_FFT_FREQUENCIES_16 = [0.0, 1378.125, 2756.25, 4134.375, 5512.5, 6890.625, 8268.75, 9646.875,
                              11025.0]


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_hz_mel_conversions_librosa_docstrings(impl):
    """>>> librosa.hz_to_mel(60) -> 0.9 ; hz_to_mel([110, 220, 440]) -> [1.65, 3.3, 6.6];
    >>> librosa.mel_to_hz(3) -> 200. ; mel_to_hz([1,2,3,4,5]) -> [66.667, 133.333, 200., 266.667, 333.333]"""
    if impl == "oracle":
        h2m = lambda f: np.array([o.hz_to_mel(float(v), False) for v in np.atleast_1d(f)])
        m2h = lambda m: np.array([o.mel_to_hz(float(v), False) for v in np.atleast_1d(m)])
    else:
        h2m = lambda f: np.atleast_1d(kb._hz_to_mel(np.atleast_1d(np.asarray(f, float)), False))
        m2h = lambda m: np.atleast_1d(kb._mel_to_hz(np.atleast_1d(np.asarray(m, float)), False))
    np.testing.assert_allclose(h2m(60), [0.9], rtol=0, atol=1e-12)
    np.testing.assert_allclose(h2m([110, 220, 440]), [1.65, 3.3, 6.6], rtol=0, atol=1e-12)
    np.testing.assert_allclose(m2h(3), [200.0], rtol=0, atol=1e-10)
    np.testing.assert_allclose(m2h([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], atol=5e-4)
    # Slaney scale constants as documented: linear 200/3 Hz per mel below 1 kHz (mel 15), then
    # log-spaced with step ln(6.4)/27: 6400 Hz is exactly 27 mels above 1 kHz.
    np.testing.assert_allclose(h2m([1000.0, 6400.0]), [15.0, 42.0], atol=1e-12)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_htk_formula_by_hand(impl):
    """HTK book: mel = 2595 log10(1 + f/700).  Hand values: f=700 -> 2595*log10(2) = 781.17...;
    f=6300 -> 2595*log10(10) = 2595 exactly; f=1000 -> 999.9855 (the '1000 mel at 1 kHz' anchor)."""
    if impl == "oracle":
        h2m = lambda f: o.hz_to_mel(float(f), True)
        m2h = lambda m: o.mel_to_hz(float(m), True)
    else:
        h2m = lambda f: float(np.atleast_1d(kb._hz_to_mel(np.array([float(f)]), True))[0])
        m2h = lambda m: float(np.atleast_1d(kb._mel_to_hz(np.array([float(m)]), True))[0])
    assert abs(h2m(6300.0) - 2595.0) < 1e-9
    assert abs(h2m(700.0) - 781.1728387) < 1e-6          # 2595 * 0.30102999566
    assert abs(h2m(1000.0) - 999.9855) < 1e-3
    assert abs(m2h(2595.0) - 6300.0) < 1e-8
    assert abs(m2h(0.0)) < 1e-12


def _mel_edges(impl, n_mels, fmin, fmax, htk):
    """mel_f of librosa.filters.mel == librosa.mel_frequencies(n_mels + 2, ...)."""
    if impl == "oracle":
        lo, hi = o.hz_to_mel(fmin, htk), o.hz_to_mel(fmax, htk)
        return np.array([o.mel_to_hz(m, htk) for m in np.linspace(lo, hi, n_mels)])
    lo, hi = kb._hz_to_mel(np.array([fmin, fmax]), htk)
    return kb._mel_to_hz(np.linspace(lo, hi, n_mels), htk)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_mel_frequencies_librosa_docstring(impl):
    got = _mel_edges(impl, 40, 0.0, 11025.0, False)
    np.testing.assert_allclose(got, LIBROSA_MEL_FREQUENCIES_40, rtol=0, atol=6e-4)   # doc prints 3 decimals


def test_fft_frequencies_librosa_docstring():
    """Bin frequencies behind both filterbanks: k * sr / n_fft."""
    fb = o.filterbank_mel(22050, 9, n_mels=4)           # n_fft = 16: only checks the bin grid indirectly
    assert fb.shape == (9, 4)
    np.testing.assert_allclose(np.arange(9) * 22050 / 16, LIBROSA_FFT_FREQUENCIES_16, rtol=0, atol=1e-9)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_filters_mel_librosa_docstring_values(impl):
    """>>> melfb = librosa.filters.mel(sr=22050, n_fft=2048)
       array([[ 0.   ,  0.016, ...,  0.   ,  0.   ], [ 0. , 0. , ... ], ...])          shape (128, 1025)
    >>> librosa.filters.mel(sr=22050, n_fft=2048, fmax=8000)
       array([[ 0.  ,  0.02, ...,  0.  ,  0.  ], ...])
    Kapre returns the transpose (backend.py:231)."""
    f = o.filterbank_mel if impl == "oracle" else kb.filterbank_mel
    fb = np.asarray(f(22050, 1025, 128))
    assert fb.shape == (1025, 128) and fb.dtype == np.float32
    assert fb[0, 0] == 0.0 and round(float(fb[1, 0]), 3) == 0.016
    assert fb[-1, 0] == 0.0 and fb[-2, 0] == 0.0 and fb[0, 1] == 0.0 and fb[1, 1] == 0.0
    assert fb[-1, -1] == 0.0 and fb[-2, -1] > 0.0          # last triangle ends exactly at fmax = sr/2
    fb8 = np.asarray(f(22050, 1025, 128, 0.0, 8000.0))
    assert round(float(fb8[1, 0]), 2) == 0.02
    # first Slaney triangle from the documented constants alone (no oracle code): all three edges lie in
    # the linear region (200/3 Hz per mel), so mel_f[1] = (200/3) * hz_to_mel(fmax)/129, mel_f[2] = 2 mel_f[1],
    # and the weight at bin 1 (f1 = sr/n_fft) is ramp f1/mel_f[1] times the Slaney norm 2/(mel_f[2]-mel_f[0]).
    import math
    for fmax, table, doc in ((11025.0, fb, 0.016), (8000.0, fb8, 0.02)):
        top_mel = 15.0 + math.log(fmax / 1000.0) / (math.log(6.4) / 27.0)
        mel_f1 = (200.0 / 3.0) * top_mel / 129.0
        want = (22050.0 / 2048.0) / mel_f1 ** 2
        assert abs(float(table[1, 0]) - want) < 1e-8
        assert round(want, 3 if doc == 0.016 else 2) == doc


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_filters_mel_norm_semantics_documented(impl):
    """librosa.filters.mel: norm='slaney' divides triangle i by its band width (area normalisation:
    2/(mel_f[i+2]-mel_f[i])); norm=None leaves unit-peak triangles; norm=1 makes each filter's weights sum
    to 1 (librosa.util.normalize(norm=1, axis=-1))."""
    f = o.filterbank_mel if impl == "oracle" else kb.filterbank_mel
    plain = np.asarray(f(22050, 1025, 40, 0.0, None, False, None), np.float64)
    assert plain.max() <= 1.0 + 1e-6
    # a triangle's peak is reached only when a bin falls on its centre; interpolated peak bound:
    assert (plain.max(axis=0) > 0.5).all()
    edges = np.asarray(LIBROSA_MEL_FREQUENCIES_40)          # published: mel_frequencies(40) = edges of 38 filters
    fb38 = np.asarray(f(22050, 1025, 38, 0.0, None, False, None), np.float64)
    slaney38 = np.asarray(f(22050, 1025, 38, 0.0, None, False, "slaney"), np.float64)
    enorm = 2.0 / (edges[2:] - edges[:-2])
    np.testing.assert_allclose(slaney38, fb38 * enorm[None, :], rtol=2e-5, atol=1e-9)
    # triangles from the PUBLISHED edges, evaluated here from the documented ramp definition
    freqs = np.arange(1025) * 22050 / 2048
    lower = (freqs[:, None] - edges[None, :-2]) / (edges[1:-1] - edges[:-2])[None, :]
    upper = (edges[None, 2:] - freqs[:, None]) / (edges[2:] - edges[1:-1])[None, :]
    tri = np.maximum(0.0, np.minimum(lower, upper))
    np.testing.assert_allclose(fb38, tri, atol=3e-5)        # edges are printed to 3 decimals
    l1 = np.asarray(f(22050, 1025, 38, 0.0, None, False, 1), np.float64)
    np.testing.assert_allclose(l1.sum(axis=0), 1.0, rtol=1e-6)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_filterbank_log_normalisation_and_centres(impl):
    """backend.filterbank_log (backend.py:234-299) is Kapre's own numpy code (it runs unmodified for
    the golden fixtures); its two librosa calls are fft_frequencies (anchored above) and
    util.normalize(norm=1, axis=1).  Independent checks: every filter sums to 1 (L1), the DC row is
    zero (basis[:, 0] is never written), and the log-normal bump's mode sits one sigma^2*ln2 below the
    nominal centre on a log2 axis -- i.e. peaks follow f_min * 2**(i/bpo) within one bin."""
    f = o.filterbank_log if impl == "oracle" else kb.filterbank_log
    sr, n_freq, n_bins, bpo, f_min = 22050, 4097, 60, 12, 65.40639133
    fb = np.asarray(f(sr, n_freq, n_bins, bpo, f_min, 0.125), np.float64)
    assert fb.shape == (n_freq, n_bins)
    np.testing.assert_allclose(fb.sum(axis=0), 1.0, rtol=1e-5)
    assert (fb[0] == 0).all() and (fb >= 0).all()
    freqs = np.arange(n_freq) * sr / (2 * (n_freq - 1))
    peak = freqs[np.argmax(fb, axis=0)]
    centre = f_min * 2.0 ** (np.arange(n_bins) / bpo)
    binw = sr / (2 * (n_freq - 1))
    assert (np.abs(peak - centre) <= 1.5 * binw + 0.01 * centre).all()
    # same thing with scipy: a normal pdf in log2-frequency times exp(-log2 f) (the reference's exponent,
    # backend.py:290-292, subtracts log2 f, not ln f), L1-normalised
    i = 40
    lf = np.log2(freqs[1:])
    sig = 0.125 / bpo
    bump = scipy.stats.norm.pdf(lf, loc=np.log2(centre[i]), scale=sig) * np.exp(-lf)
    np.testing.assert_allclose(fb[1:, i], bump / bump.sum(), rtol=2e-5, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# tf.signal windows: documented formula of window_ops._raised_cosine_window, evaluated by hand:
#   even = 1 - window_length % 2 ;  n = window_length + periodic * even - 1
#   w[k] = a - b * cos(2 pi k / n)        (hann a=b=0.5 ; hamming a=0.54, b=0.46) ; window_length 1 -> [1]
# ---------------------------------------------------------------------------------------------
HAND_WINDOWS = {
    ("hann_window", 4): [0.0, 0.5, 1.0, 0.5],                       # periodic: n = 4
    ("hann_window", 5): [0.0, 0.5, 1.0, 0.5, 0.0],                  # odd: n = 4 (symmetric!)
    ("hann_window", 3): [0.0, 1.0, 0.0],                            # odd: n = 2
    ("hann_window", 6): [0.0, 0.25, 0.75, 1.0, 0.75, 0.25],         # periodic: n = 6
    ("hann_window", 7): [0.0, 0.25, 0.75, 1.0, 0.75, 0.25, 0.0],    # odd: n = 6
    ("hann_window", 1): [1.0],
    ("hamming_window", 4): [0.08, 0.54, 1.0, 0.54],
    ("hamming_window", 5): [0.08, 0.54, 1.0, 0.54, 0.08],
    ("hamming_window", 6): [0.08, 0.31, 0.77, 1.0, 0.77, 0.31],
}


@pytest.mark.parametrize("name,n", sorted(HAND_WINDOWS))
def test_tf_windows_by_hand(name, n):
    want = HAND_WINDOWS[(name, n)]
    np.testing.assert_allclose(o.get_window(name, n), want, atol=1e-15)
    np.testing.assert_allclose(kb.get_window_fn(name)(n), want, atol=1e-7)


@pytest.mark.parametrize("n", [9, 255, 511, 2017])
def test_tf_odd_windows_equal_scipy_symmetric(n):
    """For odd lengths TF's 'periodic' flag is a no-op, i.e. the symmetric window scipy produces."""
    np.testing.assert_allclose(o.hann_window(n), scipy.signal.windows.hann(n, sym=True), atol=1e-14)
    np.testing.assert_allclose(o.hamming_window(n), scipy.signal.windows.hamming(n, sym=True), atol=1e-14)
    np.testing.assert_allclose(kb.get_window_fn("hann_window")(n), scipy.signal.windows.hann(n, sym=True),
                               atol=1e-6)


@pytest.mark.parametrize("n", [16, 64, 400, 2048])
def test_other_tf_windows_equal_scipy(n):
    """tf.signal.kaiser_window(n, beta) = I0(beta sqrt(1-((k-(n-1)/2)/((n-1)/2))^2))/I0(beta) (= numpy /
    scipy kaiser); kaiser_bessel_derived_window = scipy.signal.windows.kaiser_bessel_derived (same
    cumulative-sum definition, beta passed through unchanged); vorbis_window =
    sin(pi/2 sin^2(pi (k+1/2)/n))."""
    np.testing.assert_allclose(o.kaiser_window(n, 12.0), scipy.signal.windows.kaiser(n, 12.0, sym=True), atol=1e-12)
    np.testing.assert_allclose(o.kaiser_bessel_derived_window(n, 12.0),
                               scipy.signal.windows.kaiser_bessel_derived(n, 12.0), atol=1e-12)
    k = np.arange(n)
    np.testing.assert_allclose(o.vorbis_window(n), np.sin(np.pi / 2 * np.sin(np.pi * (k + 0.5) / n) ** 2),
                               atol=1e-14)
    for name, ref in (("kaiser_window", o.kaiser_window(n, 12.0)),
                      ("kaiser_bessel_derived_window", o.kaiser_bessel_derived_window(n, 12.0)),
                      ("vorbis_window", o.vorbis_window(n))):
        np.testing.assert_allclose(kb.get_window_fn(name)(n), ref, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# tf.signal.inverse_stft_window_fn, by hand from its documented definition:
#   denom = square(w) ; pad to overlaps*hop (overlaps = ceil(win/hop)) ; reshape (overlaps, hop) ;
#   sum over overlaps ; tile back ; w_inv = w / denom[:win]
# ---------------------------------------------------------------------------------------------
def test_inverse_stft_window_by_hand():
    # hann(4) = [0, .5, 1, .5], hop 2: w^2 = [0, .25, 1, .25] -> column sums [1, .5] -> w_inv = [0, 1, 1, 1]
    np.testing.assert_allclose(o.inverse_stft_window(4, 2, np.array([0.0, 0.5, 1.0, 0.5])), [0, 1, 1, 1], atol=1e-15)
    # hann(4), hop 1: sum of w^2 = 1.5 everywhere -> w / 1.5
    np.testing.assert_allclose(o.inverse_stft_window(4, 1, np.array([0.0, 0.5, 1.0, 0.5])),
                               [0, 1 / 3, 2 / 3, 1 / 3], atol=1e-15)
    # ragged: win 5, hop 2 (overlaps 3, padded to 6): w = [1,2,3,4,5] -> w^2 = [1,4,9,16,25,0]
    # columns: [1+9+25, 4+16+0] = [35, 20] -> w_inv = [1/35, 2/20, 3/35, 4/20, 5/35]
    np.testing.assert_allclose(o.inverse_stft_window(5, 2, np.arange(1.0, 6.0)),
                               [1 / 35, 0.1, 3 / 35, 0.2, 1 / 7], atol=1e-15)
    # hop >= win: every sample is covered once -> w / w^2 = 1 / w
    np.testing.assert_allclose(o.inverse_stft_window(3, 3, np.array([0.5, 1.0, 2.0])), [2.0, 1.0, 0.5], atol=1e-15)
    # product-side builder follows the same numbers
    fn = kb.inverse_stft_window_fn(2, lambda n: np.arange(1.0, n + 1.0))
    np.testing.assert_allclose(fn(5), [1 / 35, 0.1, 3 / 35, 0.2, 1 / 7], atol=1e-7)


# ---------------------------------------------------------------------------------------------
# tf.signal.stft / inverse_stft on inputs small enough to do on paper
# ---------------------------------------------------------------------------------------------
def test_stft_by_hand_4_point():
    """x = [1,2,3,4,5,6], frame_length 4, step 2, fft_length 4, rectangular window.
    Frames [1,2,3,4] and [3,4,5,6]; 4-point DFT of [a,b,c,d] = [a+b+c+d, (a-c) - i(b-d), a-b+c-d]."""
    x = np.arange(1.0, 7.0)
    s = o.tf_stft(x, 4, 2, 4, np.ones(4), False)
    np.testing.assert_allclose(s, [[10, -2 + 2j, -2], [18, -2 + 2j, -2]], atol=1e-12)
    # pad_end=True: ceil(6/2) = 3 frames, third = [5,6,0,0] -> [11, 5-6i, -1]
    s = o.tf_stft(x, 4, 2, 4, np.ones(4), True)
    np.testing.assert_allclose(s[2], [11, 5 - 6j, -1], atol=1e-12)
    # frame_length 2 < fft_length 4: RIGHT zero padding, frame [1,2] -> [1,2,0,0] -> [3, 1-2i, -1]
    s = o.tf_stft(x, 2, 2, 4, np.ones(2), False)
    np.testing.assert_allclose(s[0], [3, 1 - 2j, -1], atol=1e-12)


def test_inverse_stft_by_hand_4_point():
    """irfft([10, -2+2i, -2], 4) = [1,2,3,4]; two frames at hop 2 with a unit synthesis window
    overlap-add to [1, 2, 3+3, 4+4, 5, 6]; imaginary parts of DC / Nyquist are ignored (C2R)."""
    s = np.array([[10, -2 + 2j, -2], [18, -2 + 2j, -2]])
    np.testing.assert_allclose(o.tf_inverse_stft(s, 4, 2, 4, np.ones(4)), [1, 2, 6, 8, 5, 6], atol=1e-12)
    s2 = s + np.array([[5j, 0, -7j], [0, 0, 0]])
    np.testing.assert_allclose(o.tf_inverse_stft(s2, 4, 2, 4, np.ones(4)), [1, 2, 6, 8, 5, 6], atol=1e-12)


# ---------------------------------------------------------------------------------------------
# decibel: closed-form values SURVEY 8c lists for the reference's own known-answer input
# (/root/reference/tests/test_backend.py:20-22), as literals
# ---------------------------------------------------------------------------------------------
def test_decibel_literals():
    x = np.array([[1e-20, 1e-5, 1e-3, 5e-2], [0.3, 1.0, 20.5, 9999]])
    want = np.array([[-50.0, -50.0, -30.0, -13.0103], [-5.2288, 0.0, 13.1175, 39.9996]])
    for dr in (80.0, 120.0):
        np.testing.assert_allclose(o.magnitude_to_decibel(x, 1.0, 1e-5, dr), want, atol=5e-5)
    # librosa.power_to_db semantics with top_db: clamp at (row max - dynamic_range), per batch item
    np.testing.assert_allclose(o.magnitude_to_decibel(x, 1.0, 1e-5, 20.0),
                               [[-33.0103, -33.0103, -30.0, -13.0103], [19.9996, 19.9996, 19.9996, 39.9996]],
                               atol=5e-5)
    # ref_value: subtracts 10 log10(max(amin, ref)) -- ref 10 shifts everything by -10 dB
    np.testing.assert_allclose(o.magnitude_to_decibel(x, 10.0, 1e-5, 120.0), want - 10.0, atol=5e-5)


# ---------------------------------------------------------------------------------------------
# tf.signal.stft as restated BY THE REFERENCE ITSELF (kapre/tflite_compatible_stft.py, asserted equal
# to tf.signal.stft upstream, tests/test_time_frequency.py:270-337), run on numpy primitives by
# oracle/make_golden_l0.py with scipy windows: no oracle arithmetic in the fixture
# ---------------------------------------------------------------------------------------------
def _tflite_cases():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tflite_stft_cases.npz")
    z = np.load(path)
    return z, sorted({k.split("/")[0] for k in z.files})


@pytest.mark.parametrize("name", _tflite_cases()[1])
def test_stft_equals_reference_tflite_restatement(name):
    z, _ = _tflite_cases()
    frame, fft, step, pad_end = (int(v) for v in z[name + "/params"])
    x, y, w = z[name + "/x"].astype(np.float64), z[name + "/y"], z[name + "/window"]
    got = o.tf_stft(x, frame, step, fft, w, bool(pad_end))
    assert got.shape == y.shape[:-1]
    scale = np.abs(y).max()
    np.testing.assert_allclose(got.real, y[..., 0], atol=2e-6 * scale)       # reference DFT matrix is complex64
    np.testing.assert_allclose(got.imag, y[..., 1], atol=2e-6 * scale)
    if frame % 2 == 0:      # the scipy window the fixture was made with IS the oracle's even-length window
        name_w = "hamming_window" if abs(w[0] - 0.08) < 1e-12 else "hann_window"
        np.testing.assert_allclose(o.get_window(name_w, frame), w, atol=1e-15)
