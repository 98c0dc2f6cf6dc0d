"""CPU tests of the oracle (oracle/kapre_oracle.py):
  (1) against tests/golden/kapre_ref_cases.npz -- outputs of the REAL reference layer code run on
      numpy stand-ins (oracle/make_golden.py): pins the Kapre-level glue;
  (2) against independent implementations of the L0 arithmetic (explicit DFT, scipy, torch CPU);
  (3) against the closed-form / known-answer vectors the reference's own tests hold.
"""
import numpy as np
import pytest
import scipy.fft
import scipy.signal

import kapre_oracle as o
from conftest import golden_names, speech

TIGHT = dict(rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------- (1) reference-run golden vectors
@pytest.mark.parametrize("name", golden_names("stft"))
def test_stft_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_stft(x, **kw), y, **TIGHT)


@pytest.mark.parametrize("name", golden_names("stft_magnitude"))
def test_stft_magnitude_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_stft_magnitude(x, **kw), y, **TIGHT)


@pytest.mark.parametrize("name", golden_names("melspectrogram"))
def test_melspectrogram_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_melspectrogram(x, **kw), y, **TIGHT)


@pytest.mark.parametrize("name", golden_names("roundtrip"))
def test_roundtrip_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    wf, sf = kw["waveform_data_format"], kw["stft_data_format"]
    s = o.kapre_stft(x, kw["n_fft"], kw["n_fft"], kw["hop_length"], "hann_window", True, True,
                     wf, sf)
    r = o.kapre_istft(s, kw["n_fft"], kw["n_fft"], kw["hop_length"], "hann_window", sf, wf)
    np.testing.assert_allclose(r, y, **TIGHT)
    # perfect reconstruction after trimming (tests/test_time_frequency.py:480-486)
    lp = kw["n_fft"] - kw["hop_length"]
    t_axis = 2 if wf == "channels_first" else 1
    n = x.shape[t_axis]
    rec = np.take(r, np.arange(lp, lp + n), axis=t_axis)
    np.testing.assert_allclose(rec, x, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("istft"))
def test_istft_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_istft(x, **kw), y, **TIGHT)


@pytest.mark.parametrize("name", golden_names("apply_filterbank"))
def test_apply_filterbank_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    fb = (o.filterbank_mel if kw["type"] == "mel" else o.filterbank_log)(**kw["filterbank_kwargs"])
    np.testing.assert_allclose(o.apply_filterbank(x, fb, kw["data_format"]), y, **TIGHT)


@pytest.mark.parametrize("name", golden_names("magnitude_to_decibel"))
def test_decibel_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.magnitude_to_decibel(x, **kw), y, **TIGHT)


# ---------------------------------------------------------------- (3) reference known answers
def test_decibel_known_answers():
    """Input of tests/test_backend.py:20-22; expected values are closed form (power_to_db with
    ref=1, amin=1e-5; no clamping at dynamic_range 80/120)."""
    x = np.array([[1e-20, 1e-5, 1e-3, 5e-2], [0.3, 1.0, 20.5, 9999]])
    want = np.array([[-50.0, -50.0, -30.0, 10 * np.log10(5e-2)],
                     [10 * np.log10(0.3), 0.0, 10 * np.log10(20.5), 10 * np.log10(9999.0)]])
    for dr in (80.0, 120.0):
        np.testing.assert_allclose(o.magnitude_to_decibel(x, 1.0, 1e-5, dr), want, atol=1e-12)
    # dynamic range 20 clamps row 0 at -13.01-20 and row 1 at 40-20
    got = o.magnitude_to_decibel(x, 1.0, 1e-5, 20.0)
    assert np.isclose(got[0, 0], 10 * np.log10(5e-2) - 20) and np.isclose(got[1, 0], 10 * np.log10(9999.0) - 20)


@pytest.mark.parametrize("t,win,hop", [(8000, 1000, 250), (8000, 512, 256), (16000, 512, 256),
                                       (44100, 2048, 512), (44100, 2048, 1024), (160000, 1024, 160)])
def test_frame_count_formulas(t, win, hop):
    # tests/test_time_frequency.py:32-39
    assert o.num_frames(t, win, hop, False) == (t - (win - hop)) // hop
    assert o.num_frames(t, win, hop, True) == int(np.ceil(float(t) / hop))


def test_frame_counts_of_baseline_configs():
    assert o.num_frames(16000, 512, 256, False) == 61
    assert o.num_frames(44100, 2048, 512, False) == 83
    assert o.num_frames(44100, 2048, 1024, False) == 42       # examples notebook: 42 frames x 1025
    assert o.num_frames(110250 + 768, 1024, 256, True) == 434
    assert o.num_frames(160000, 1024, 160, False) == 994
    assert o.num_frames(100, 512, 256, False) == 0


# ---------------------------------------------------------------- (2) L0 cross-checks
@pytest.mark.parametrize("n", [512, 1024, 2048, 400])
@pytest.mark.parametrize("name,sp", [("hann_window", "hann"), ("hamming_window", "hamming")])
def test_even_windows_equal_scipy_periodic(n, name, sp):
    np.testing.assert_allclose(o.get_window(name, n), scipy.signal.get_window(sp, n, fftbins=True),
                               atol=1e-15)


def test_odd_window_rule_is_tf_not_scipy():
    # tf.signal periodic windows only honour `periodic` for even lengths: odd n is symmetric
    w = o.hann_window(511)
    np.testing.assert_allclose(w, scipy.signal.get_window("hann", 511, fftbins=False), atol=1e-15)
    assert np.abs(w - scipy.signal.get_window("hann", 511, fftbins=True)).max() > 1e-3
    assert o.hann_window(1).tolist() == [1.0]


def test_kaiser_window_matches_numpy():
    np.testing.assert_allclose(o.kaiser_window(64, 12.0), np.kaiser(64, 12.0), atol=1e-14)


@pytest.mark.parametrize("n_fft,win,hop,pad_end", [(512, 512, 256, False), (1000, 1000, 250, False),
                                                   (1000, 512, 256, True), (256, 200, 80, True)])
def test_stft_fft_equals_explicit_dft_and_scipy(n_fft, win, hop, pad_end):
    x = speech(3000).astype(np.float64)
    w = o.hann_window(win)
    a = o.tf_stft(x, win, hop, n_fft, w, pad_end)
    b = o.tf_stft(x, win, hop, n_fft, w, pad_end, use_matrix=True)
    np.testing.assert_allclose(a, b, atol=1e-10)
    fr = o.frame(x, win, hop, pad_end) * w
    c = scipy.fft.rfft(np.pad(fr, [(0, 0), (0, max(0, n_fft - win))])[:, :n_fft], axis=-1)
    np.testing.assert_allclose(a, c, atol=1e-10)


def test_stft_equals_torch_stft_center_false():
    import torch

    x = speech(8000)
    n_fft, hop = 512, 256
    ref = torch.stft(torch.from_numpy(x).double(), n_fft, hop_length=hop, win_length=n_fft,
                     window=torch.from_numpy(o.hann_window(n_fft)), center=False,
                     return_complex=True).numpy().T
    got = o.tf_stft(x, n_fft, hop, n_fft, o.hann_window(n_fft), False)
    np.testing.assert_allclose(got, ref, atol=1e-10)


def test_inverse_window_and_cola_identity():
    for n_fft, hop in ((2048, 512), (1024, 256), (512, 128), (400, 100)):
        w = o.hann_window(n_fft)
        wi = o.inverse_stft_window(n_fft, hop, w)
        # sum over shifts of w * w_inv == 1 in the steady state (perfect reconstruction)
        acc = np.zeros(n_fft * 4)
        for s in range(0, n_fft * 3 + 1, hop):
            acc[s:s + n_fft] += (w * wi)[: len(acc[s:s + n_fft])]
        np.testing.assert_allclose(acc[n_fft:2 * n_fft], 1.0, atol=1e-12)


def test_istft_equals_torch_istft_on_a_cola_window():
    import torch

    n_fft, hop = 512, 128
    x = speech(4096).astype(np.float64)
    w = o.hamming_window(n_fft)        # non-zero edges: torch refuses a zero envelope
    s = o.tf_stft(x, n_fft, hop, n_fft, w, False)
    y = o.tf_inverse_stft(s, n_fft, hop, n_fft, o.inverse_stft_window(n_fft, hop, w))
    ref = torch.istft(torch.from_numpy(s.T), n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.from_numpy(w), center=False, length=None).numpy()
    # torch normalises by the window envelope; compare the fully-overlapped interior
    np.testing.assert_allclose(y[n_fft:len(ref) - n_fft], ref[n_fft:len(ref) - n_fft], atol=1e-9)


def test_analytic_sinusoid_and_impulse():
    n_fft = 512
    n = np.arange(n_fft)
    k0 = 37
    x = np.cos(2 * np.pi * k0 * n / n_fft)
    s = o.tf_stft(x, n_fft, n_fft, n_fft, np.ones(n_fft), False)[0]
    want = np.zeros(n_fft // 2 + 1)
    want[k0] = n_fft / 2
    np.testing.assert_allclose(np.abs(s), want, atol=1e-9)
    imp = np.zeros(n_fft)
    imp[100] = 1.0
    w = o.hamming_window(n_fft)
    s = o.tf_stft(imp, n_fft, n_fft, n_fft, w, False)[0]
    np.testing.assert_allclose(np.abs(s), w[100], atol=1e-12)          # impulse -> window sample


def test_parseval():
    x = speech(2048).astype(np.float64)
    s = o.tf_stft(x, 2048, 2048, 2048, np.ones(2048), False)[0]
    e = (np.abs(s[0]) ** 2 + np.abs(s[-1]) ** 2 + 2 * np.sum(np.abs(s[1:-1]) ** 2)) / 2048
    assert np.isclose(e, np.sum(x * x), rtol=1e-12)


def test_mel_filterbank_structure():
    fb = o.filterbank_mel(44100, 1025, 128)
    assert fb.shape == (1025, 128) and fb.dtype == np.float32
    assert (fb >= 0).all() and (fb.max(axis=0) > 0).all()              # no empty filters
    assert ((fb != 0).sum(axis=1) <= 2).all()                           # <= 2 filters per bin
    # slaney norm: each triangle integrates to ~1 over frequency (bin width sr/n_fft)
    area = fb.sum(axis=0) * (44100 / 2048)
    assert np.all(np.abs(area - 1.0) < 0.15) and np.all(np.abs(area[64:] - 1.0) < 0.01)


def test_mel_filterbank_htk_and_hz_mapping():
    assert np.isclose(o.hz_to_mel(1000.0, False), 15.0)
    assert np.isclose(o.mel_to_hz(o.hz_to_mel(4321.0, False), False), 4321.0)
    assert np.isclose(o.hz_to_mel(700.0, True), 2595.0 * np.log10(2.0))
    assert np.isclose(o.mel_to_hz(o.hz_to_mel(4321.0, True), True), 4321.0)


# ------------------------------------------------------------------ SURVEY 8f row 4 layers
@pytest.mark.parametrize("name", golden_names("frame"))
def test_frame_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    got = o.kapre_frame(x, **kw)
    assert got.shape == y.shape
    np.testing.assert_array_equal(got, y)


@pytest.mark.parametrize("name", golden_names("energy"))
def test_energy_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_energy(x, **kw), y, rtol=1e-12)


@pytest.mark.parametrize("name", golden_names("delta"))
def test_delta_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_delta(x, **kw), y, rtol=1e-12, atol=1e-12)


def test_delta_known_answer_of_the_reference_test(golden):
    """/root/reference/tests/test_time_frequency.py:375-387"""
    kw, x, y, _ = golden.get("delta_known")
    np.testing.assert_allclose(y.reshape(-1), [0.5, 1.0, 1.0, 0.5])
    np.testing.assert_allclose(o.kapre_delta(x, **kw).reshape(-1), [0.5, 1.0, 1.0, 0.5])


@pytest.mark.parametrize("name", golden_names("logmel_to_mfcc"))
def test_mfcc_matches_reference_run(golden, name):
    kw, x, y, _ = golden.get(name)
    np.testing.assert_allclose(o.kapre_logmel_to_mfcc(x, **kw), y, rtol=1e-12, atol=1e-10)


def test_frame_equals_strided_view_and_counts():
    """tf.signal.frame semantics against an independent construction (what the reference's own test
    does with librosa.util.frame, tests/test_signal.py:29-38)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1000)
    for L, hop in ((50, 25), (32, 16), (64, 64), (400, 160)):
        view = np.lib.stride_tricks.sliding_window_view(x, L)[::hop]
        np.testing.assert_array_equal(o.tf_frame(x, L, hop), view)
        padded = o.tf_frame(x, L, hop, pad_end=True, pad_value=7.0)
        assert padded.shape[0] == -(-1000 // hop)
        np.testing.assert_array_equal(padded[:view.shape[0]], view)
        tail = padded[-1]
        n_real = 1000 - (padded.shape[0] - 1) * hop
        assert (tail[n_real:] == 7.0).all()
    assert o.tf_frame(x[:10], 50, 25).shape == (0, 50)


def test_mfcc_definition_equals_scipy_dct_and_the_reference_relation():
    """tf.signal.mfccs_from_log_mel_spectrograms = (unnormalised DCT-II) * rsqrt(2N); against
    scipy.fft.dct, and the relation to the orthonormal DCT the reference asserts
    (tests/test_signal.py:103-106: bins >= 1 equal, bin 0 differs by sqrt(2))."""
    import scipy.fft as sfft

    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 7, 128)) * 20 - 30
    got = o.mfccs_from_log_mel_spectrograms(x)
    np.testing.assert_allclose(got, sfft.dct(x, type=2, norm=None, axis=-1) / np.sqrt(2 * 128), rtol=1e-11, atol=1e-9)
    ortho = sfft.dct(x, type=2, norm="ortho", axis=-1)
    np.testing.assert_allclose(got[..., 1:], ortho[..., 1:], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got[..., 0] / np.sqrt(2.0), ortho[..., 0], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(x @ o.mfcc_matrix(128, 20), got[..., :20], rtol=1e-11, atol=1e-9)
