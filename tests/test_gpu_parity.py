"""-m gpu: parity of the HIP path (through the Kapre-shaped layer API -> ctypes -> C ABI) against
  * the committed reference-run golden vectors (tests/golden/kapre_ref_cases.npz),
  * the float64 oracle on seeded inputs at BASELINE.json's configurations,
  * size-independent properties at full size,
and the edge cases the reference tests (ragged / empty inputs, odd windows, non power-of-two FFT).

Tolerance (north_star): <= 1e-4 relative to the scale of the reference output, float32.
The reference's own tolerances (tests/test_time_frequency.py:65-69,120,256,265-267,486,534) are
looser and are asserted too where they apply.
"""
import os

import numpy as np
import pytest

import kapre_oracle as o
from conftest import golden_names, rel_err, speech

import kapre_amd as kapre
from kapre_amd import (STFT, InverseSTFT, Magnitude, Phase, MagnitudeToDecibel, ApplyFilterbank,
                       Sequential, Input, backend, composed)

pytestmark = pytest.mark.gpu

REL = 1e-4          # north_star tolerance, relative to output scale (the CONTRACT)
REG = 4e-6          # regression bound asserted next to it: the kernels measure 1e-7 ... 8e-7 against the float64 oracle, and a
                    # kernel that loses three digits must not pass because the contract is loose
DB_ABS = 1e-3       # dB outputs: absolute decibel tolerance (upstream: rtol 3e-3 of ~20-80 dB; measured ~5e-6)


def to_np(t):
    return t.detach().cpu().numpy()


def assert_close(got, want, rel=REL, reg=REG):
    """`rel`: the contract; `reg`: the regression bound of the float32 kernels (pass reg=None where the comparison is not
    against float64 truth of the same arithmetic, e.g. a float32 reference-run fixture)"""
    got = np.asarray(got)
    assert got.shape == tuple(want.shape), (got.shape, want.shape)
    assert np.isfinite(got).all()
    e = rel_err(got, want)
    assert e <= rel, "relative error %.3g > %.1g" % (e, rel)
    if reg is not None and rel >= REL:
        assert e <= reg, "relative error %.3g: inside the contract (%.0e) but beyond the regression bound %.0e" % (e, rel, reg)


def assert_db_close(got, want):
    """Decibel outputs: 1e-3 dB on every value within 40 dB of the maximum (measured: ~5e-6 dB on mel outputs),
    and the north-star bound in the LINEAR domain everywhere -- a bin 60 dB down carries a 1e-4-of-max linear
    error as ~0.4 dB, so one flat dB tolerance cannot express the requirement; 0.02 dB caps those too
    (upstream asserts rtol 3e-3 of 20..80 dB)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == tuple(want.shape)
    err = np.abs(got - want)
    strong = want >= want.max() - 40.0
    assert err[strong].max() <= DB_ABS, "max dB error on strong values %.4g" % err[strong].max()
    assert err.max() <= 0.02, "max dB error %.4g" % err.max()
    lin_g, lin_w = 10.0 ** (got / 10.0), 10.0 ** (want / 10.0)
    assert np.abs(lin_g - lin_w).max() <= REL * lin_w.max()
    np.testing.assert_allclose(got, want, rtol=3e-3, atol=0.02)        # upstream tolerance


# ------------------------------------------------------------------ golden vectors (reference run)
@pytest.mark.parametrize("name", golden_names("stft"))
def test_stft_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    s = to_np(STFT(**kw)(x))
    assert s.dtype == np.complex64
    assert_close(s, y)
    # the reference's own check: allclose_complex_numbers (rtol 1e-5, atol 1e-3)
    np.testing.assert_allclose(np.abs(s), np.abs(y), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s.real, y.real, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s.imag, y.imag, rtol=1e-5, atol=1e-3)
    # fused magnitude / phase epilogues == separate layers
    mag = to_np(Sequential([STFT(**kw), Magnitude()])(x))
    np.testing.assert_allclose(mag, np.abs(y), atol=2e-4)
    assert_close(mag, np.abs(y))
    ph = to_np(Sequential([STFT(**kw), Phase()])(x))
    strong = np.abs(y) > 1e-2 * np.abs(y).max()
    np.testing.assert_allclose(np.sin(ph)[strong], np.sin(np.angle(y))[strong], atol=1e-3)
    np.testing.assert_allclose(np.cos(ph)[strong], np.cos(np.angle(y))[strong], atol=1e-3)
    ph2 = to_np(Phase()(STFT(**kw)(x)))
    np.testing.assert_allclose(np.cos(ph2)[strong], np.cos(ph)[strong], atol=1e-4)


@pytest.mark.parametrize("name", golden_names("stft_magnitude"))
def test_stft_magnitude_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    got = to_np(composed.get_stft_magnitude_layer(**kw)(x))
    if kw.get("return_decibel"):
        assert_db_close(got, y)
    else:
        assert_close(got, y)


@pytest.mark.parametrize("name", golden_names("melspectrogram"))
def test_melspectrogram_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    model = composed.get_melspectrogram_layer(**kw)
    got = to_np(model(x))
    if kw.get("return_decibel"):
        assert_db_close(got, y)
    else:
        assert_close(got, y)
        np.testing.assert_allclose(got, y, atol=1e-4)          # upstream tolerance (:256)
    # layer-by-layer execution (user re-adds .layers to an own model) gives the same numbers
    z = x
    for layer in model.layers:
        z = layer(z)
    z = to_np(z)
    if kw.get("return_decibel"):
        assert_db_close(z, y)
    else:
        assert_close(z, y)
    # numpy in / numpy out
    p = model.predict(x)
    assert isinstance(p, np.ndarray) and np.array_equal(p, got)


@pytest.mark.parametrize("name", golden_names("roundtrip"))
def test_roundtrip_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    stft, istft = composed.get_perfectly_reconstructing_stft_istft(**kw)
    rec = to_np(istft(stft(x)))
    assert_close(rec, y)
    lp = kw["n_fft"] - kw["hop_length"]
    t_axis = 2 if kw["waveform_data_format"] == "channels_first" else 1
    n = x.shape[t_axis]
    trimmed = np.take(rec, np.arange(lp, lp + n), axis=t_axis)
    np.testing.assert_allclose(trimmed, x, atol=1e-5)          # upstream tolerance (:486)


@pytest.mark.parametrize("name", golden_names("istft"))
def test_istft_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    assert_close(to_np(InverseSTFT(**kw)(x)), y)


@pytest.mark.parametrize("name", golden_names("apply_filterbank"))
def test_apply_filterbank_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    assert_close(to_np(ApplyFilterbank(**kw)(x)), y, rel=2e-6)


@pytest.mark.parametrize("name", golden_names("magnitude_to_decibel"))
def test_decibel_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    got = to_np(backend.magnitude_to_decibel(x, **kw))
    np.testing.assert_allclose(got, y, atol=1e-4)
    got = to_np(MagnitudeToDecibel(**kw)(x))
    np.testing.assert_allclose(got, y, atol=1e-4)


def test_decibel_known_answers_reference_tolerance():
    # tests/test_backend.py:15-40 (atol 1e-5 there, float32 input)
    x = np.array([[1e-20, 1e-5, 1e-3, 5e-2], [0.3, 1.0, 20.5, 9999]], dtype=np.float32)
    for dr in (80.0, 120.0):
        want = o.magnitude_to_decibel(x.astype(np.float64), 1.0, 1e-5, dr)
        np.testing.assert_allclose(to_np(backend.magnitude_to_decibel(x, 1.0, 1e-5, dr)), want, atol=1e-5)


# ------------------------------------------------------------------ BASELINE.json configurations
def synth(shape, seed):
    return np.random.default_rng(seed).uniform(-1, 1, shape).astype(np.float32)


def test_cfg1_stft_magnitude_full():
    """configs[0]: STFT+Magnitude, batch=4, 1ch, 16000 @16k, n_fft=512 hop=256."""
    x = synth((4, 16000, 1), 1234)
    got = to_np(composed.get_stft_magnitude_layer(n_fft=512, hop_length=256)(x))
    want = o.kapre_stft_magnitude(x, n_fft=512, hop_length=256)
    assert got.shape == (4, 61, 257, 1)
    assert_close(got, want)
    # and on the reference's speech fixture
    xs = speech(8000)[None, :, None]
    assert_close(to_np(composed.get_stft_magnitude_layer(n_fft=512, hop_length=256)(xs)),
                 o.kapre_stft_magnitude(xs, n_fft=512, hop_length=256))


def test_cfg2_melspectrogram_full():
    """configs[1] (the bench workload): batch=64, 1ch, 44100 @44.1k, n_fft=2048 hop=512 n_mels=128."""
    x = synth((64, 44100, 1), 1235)
    kw = dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    want = o.kapre_melspectrogram(x, **kw)
    assert got.shape == (64, 83, 128, 1)
    assert_close(got, want)
    # scaled-down and silent-tail variants exercise small magnitudes / the amin clamp
    x2 = x[:4] * np.float32(1e-3)
    x2[:, 30000:, :] = 0
    kwd = dict(kw, return_decibel=True)
    assert_close(to_np(composed.get_melspectrogram_layer(**kw)(x2)), o.kapre_melspectrogram(x2, **kw))
    assert_db_close(to_np(composed.get_melspectrogram_layer(**kwd)(x2)), o.kapre_melspectrogram(x2, **kwd))


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_cfg3_logmel_db_six_channels(fmt):
    """configs[2] at reduced batch: 6ch, 44100, n_fft=2048 hop=1024 n_mels=128, dB."""
    shape = (8, 44100, 6) if fmt == "channels_last" else (8, 6, 44100)
    x = synth(shape, 1236)
    x *= np.linspace(0.05, 1.0, 8, dtype=np.float32).reshape(8, 1, 1)
    kw = dict(n_fft=2048, hop_length=1024, sample_rate=44100, n_mels=128, return_decibel=True,
              input_data_format=fmt, output_data_format=fmt)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    want = o.kapre_melspectrogram(x, **kw)
    assert got.shape == ((8, 42, 128, 6) if fmt == "channels_last" else (8, 6, 42, 128))
    assert_db_close(got, want)


def test_cfg4_roundtrip():
    """configs[3] at reduced batch: 5s @22.05k, n_fft=1024 hop=256, pad_begin+pad_end."""
    x = synth((4, 110250, 1), 1237)
    stft, istft = composed.get_perfectly_reconstructing_stft_istft(1024, 256, "channels_last",
                                                                   "channels_last")
    s = stft(x)
    assert tuple(s.shape) == (4, 434, 513, 1)
    assert_close(to_np(s), o.kapre_stft(x, 1024, 1024, 256, "hann_window", True, True))
    rec = to_np(istft(s))
    assert rec.shape == (4, 433 * 256 + 1024, 1)
    np.testing.assert_allclose(rec[:, 768:768 + 110250], x, atol=1e-5)


def test_cfg5_mel_16k():
    """configs[4] at reduced batch: 10s @16k, n_fft=1024 hop=160 n_mels=80."""
    x = synth((4, 160000, 1), 1238)
    kw = dict(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=80)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    assert got.shape == (4, 994, 80, 1)
    assert_close(got, o.kapre_melspectrogram(x, **kw))


# ------------------------------------------------------------------ properties at full size
def test_full_size_properties_target_workload():
    """north_star target batch=256 x 1ch x 44.1k: linearity in amplitude, batch independence,
    determinism, dB dynamic range."""
    import torch

    x = torch.from_numpy(synth((256, 44100, 1), 99)).cuda()
    kw = dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128)
    mel = composed.get_melspectrogram_layer(**kw)
    a = mel(x)
    assert tuple(a.shape) == (256, 83, 128, 1) and a.is_cuda
    b = mel(x)
    assert torch.equal(a, b)                                   # bitwise deterministic
    c = mel(x * 0.5)
    torch.testing.assert_close(c, a * 0.5, rtol=1e-5, atol=1e-6)   # |X| is 1-homogeneous
    d = mel(x[17:18])
    assert torch.equal(d[0], a[17])                            # batch items are independent
    sub = o.kapre_melspectrogram(to_np(x[250:256]), **kw)
    assert_close(to_np(a[250:256]), sub)
    db = composed.get_melspectrogram_layer(**dict(kw, return_decibel=True, db_dynamic_range=25.0))(x)
    flat = db.reshape(256, -1)
    spread = flat.max(dim=1).values - flat.min(dim=1).values
    assert float(spread.max()) <= 25.0 + 1e-4


def test_stft_linearity_and_shift():
    import torch

    x = torch.from_numpy(synth((3, 2, 30000), 5)).cuda()
    y = torch.from_numpy(synth((3, 2, 30000), 6)).cuda()
    st = STFT(n_fft=1024, hop_length=256, input_data_format="channels_first",
              output_data_format="channels_first")
    lhs = st(2.0 * x - 0.5 * y)
    rhs = 2.0 * st(x) - 0.5 * st(y)
    assert float((lhs - rhs).abs().max()) <= 1e-4 * float(rhs.abs().max())
    # shifting the signal by one hop shifts the frames by one
    s0 = st(x)
    s1 = st(x[:, :, 256:])
    assert float((s0[:, :, 1:] - s1[:, :, : s0.shape[2] - 1]).abs().max()) <= 1e-5 * float(s0.abs().max())


def test_parseval_rect_window():
    x = synth((2, 1, 8192), 8)
    st = STFT(n_fft=2048, hop_length=2048, window_name="vorbis_window",
              input_data_format="channels_first", output_data_format="channels_first")
    s = to_np(st(x)).astype(np.complex128)
    w = o.vorbis_window(2048)
    for b in range(2):
        for f in range(4):
            fr = x[b, 0, f * 2048:(f + 1) * 2048].astype(np.float64) * w
            e = (np.abs(s[b, 0, f, 0]) ** 2 + np.abs(s[b, 0, f, -1]) ** 2
                 + 2 * np.sum(np.abs(s[b, 0, f, 1:-1]) ** 2)) / 2048
            assert np.isclose(e, np.sum(fr * fr), rtol=1e-5)


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("n_fft,win,hop,pad_begin,pad_end,t", [
    (512, 512, 256, False, False, 100),      # shorter than one window: zero frames
    (512, 512, 256, False, True, 100),       # ... but pad_end yields ceil(T/hop) frames
    (512, 512, 600, False, False, 5000),     # hop > win
    (512, 511, 100, True, False, 3000),      # odd window (tf periodic-window quirk)
    (2048, 2018, 1024, False, True, 9000),   # README example win_length=2018
    (1000, 1000, 250, False, False, 8000),   # reference test size: DFT-as-GEMM path
    (1000, 512, 256, True, True, 8000),
    (256, 256, 64, False, False, 3000),
    (1024, 400, 160, False, True, 16000),    # speech-style 25 ms / 10 ms
    (384, 384, 96, False, False, 4000),      # 3 * 2^7: GEMM path
    (4096, 4096, 1024, False, False, 20000), # above the Stockham sizes: GEMM path
])
@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_stft_edge_shapes(n_fft, win, hop, pad_begin, pad_end, t, fmt):
    shape = (2, t, 2) if fmt == "channels_last" else (2, 2, t)
    x = synth(shape, n_fft + t)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=pad_begin, pad_end=pad_end,
              input_data_format=fmt, output_data_format=fmt)
    want = o.kapre_stft(x, **kw)
    got = to_np(STFT(**kw)(x))
    assert got.shape == want.shape
    if want.size:
        assert_close(got, want)
        assert_close(to_np(Sequential([STFT(**kw), Magnitude()])(x)), np.abs(want))


@pytest.mark.parametrize("n_fft, hop, ch", [(1024, 256, 4), (1024, 256, 2), (1024, 160, 3), (1024, 256, 6), (2048, 512, 2),
                                             (512, 128, 4)])
@pytest.mark.parametrize("fmt_in, fmt_out", [("channels_last", "channels_first"), ("channels_first", "channels_last"),
                                             ("channels_last", "channels_last")])
def test_stft_mixed_layouts_ticket_kernel(n_fft, hop, ch, fmt_in, fmt_out):
    """k_stft3 with several channels (regression, round 4: its one-run store of a wave's G rows was also taken for
    channel-fastest frame numbering -- channels_last in, channels_first out, n_fft 1024 -- where those rows are not
    neighbours; no test had that combination at a size that reaches the kernel), and its channels_last store."""
    from kapre_amd import _ffi
    batch, frames = 3, 37
    t = n_fft + (frames - 1) * hop - 55
    x = synth((batch, t, ch) if fmt_in == "channels_last" else (batch, ch, t), 99 + n_fft + ch)
    kw = dict(n_fft=n_fft, hop_length=hop, pad_begin=True, pad_end=True, input_data_format=fmt_in, output_data_format=fmt_out)
    want = o.kapre_stft(x, **kw)
    for variant in (3, 1, 0):
        old = _ffi.set_option("stft_variant", variant)
        try:
            got = to_np(STFT(**kw)(x))
            mag = to_np(Sequential([STFT(**kw), Magnitude()])(x))
        finally:
            _ffi.set_option("stft_variant", old)
        assert_close(got, want)
        assert_close(mag, np.abs(want))


@pytest.mark.parametrize("fmt_in, fmt_out", [("channels_last", "channels_first"), ("channels_first", "channels_last"),
                                             ("channels_last", "channels_last"), ("channels_first", "channels_first")])
@pytest.mark.parametrize("n_fft, hop, ch, frames", [(1024, 256, 2, 700), (512, 128, 3, 450), (2048, 512, 2, 260)])
def test_istft_and_phase_layout_pairs_at_launch_sizes_that_reach_the_big_kernels(n_fft, hop, ch, frames, fmt_in, fmt_out):
    """Every input / output layout pair at sizes where the automatic choice is one of the large-launch kernels (the STFT bug of
    round 4 sat in exactly such a corner: right at test sizes, wrong beyond 16 frame groups per CU)."""
    rng = np.random.default_rng(n_fft + ch)
    batch, k = 4, n_fft // 2 + 1
    shape = (batch, frames, k, ch) if fmt_in == "channels_last" else (batch, ch, frames, k)
    s = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    kw = dict(n_fft=n_fft, hop_length=hop, input_data_format=fmt_in, output_data_format=fmt_out)
    assert_close(to_np(InverseSTFT(**kw)(s)), o.kapre_istft(s, **kw))
    t = n_fft + (frames - 1) * hop
    x = synth((batch, t, ch) if fmt_in == "channels_last" else (batch, ch, t), 5 + n_fft)
    want = o.kapre_stft(x, **kw)
    ph = to_np(Sequential([STFT(**kw), Phase()])(x))
    big = np.abs(want) > 1e-3 * np.abs(want).max()                # the phase of a near-zero bin is noise
    d = np.angle(np.exp(1j * (ph - np.angle(want))))
    assert np.abs(d[big]).max() < 2e-3


@pytest.mark.parametrize("fmt_in, fmt_out", [("channels_last", "channels_first"), ("channels_first", "channels_last"),
                                             ("channels_last", "channels_last"), ("channels_first", "channels_first")])
@pytest.mark.parametrize("n_fft, hop, ch", [(2048, 512, 1), (2048, 512, 2), (2048, 700, 3), (2048, 1024, 4), (1024, 160, 2),
                                             (1024, 256, 4), (512, 128, 2), (512, 200, 3)])
def test_mel_layout_pairs_at_launch_sizes_that_reach_the_big_kernels(n_fft, hop, ch, fmt_in, fmt_out):
    """The fused log-mel chain over every layout pair and channel count at ~8 k frames per channel: the automatic choices
    there (sixteen-wave k_mel_pw, its PAIR form, the stereo pair fetch) are not the ones small test shapes get."""
    batch, frames = 64, 130
    t = n_fft + (frames - 1) * hop - 31
    x = synth((batch, t, ch) if fmt_in == "channels_last" else (batch, ch, t), 1234 + n_fft + ch)
    x *= np.logspace(-2, 0, batch, dtype=np.float32).reshape(batch, 1, 1)
    kw = dict(n_fft=n_fft, hop_length=hop, sample_rate=22050, n_mels=64, pad_end=True, return_decibel=True,
              input_data_format=fmt_in, output_data_format=fmt_out)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    assert_db_close(got, o.kapre_melspectrogram(x, **kw))


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
@pytest.mark.parametrize("k, n_mels, ch", [(1025, 128, 1), (1025, 128, 2), (513, 80, 3), (257, 40, 2)])
def test_apply_filterbank_at_large_sizes(k, n_mels, ch, fmt):
    """Stand-alone ApplyFilterbank on ~10 k rows per channel (the loader-wave MFMA kernel / the generic GEMM), both layouts."""
    batch, frames = 24, 420
    x = np.abs(synth((batch, frames, k, ch) if fmt == "channels_last" else (batch, ch, frames, k), 7 + k + ch))
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=k, n_mels=n_mels), data_format=fmt)
    assert_close(to_np(layer(x)), o.apply_filterbank(x, o.filterbank_mel(22050, k, n_mels), fmt), rel=2e-6)


def test_empty_batch_and_zero_frames():
    import torch

    out = composed.get_melspectrogram_layer(n_fft=512)(np.zeros((0, 4000, 1), np.float32))
    assert tuple(out.shape) == (0, 28, 128, 1)
    out = composed.get_melspectrogram_layer(n_fft=512, return_decibel=True)(np.zeros((2, 100, 1), np.float32))
    assert tuple(out.shape) == (2, 0, 128, 1)
    out = InverseSTFT(n_fft=512)(torch.zeros((2, 0, 257, 1), dtype=torch.complex64))
    assert tuple(out.shape) == (2, 0, 1)


def test_silence_and_amin_floor():
    x = np.zeros((2, 8000, 1), np.float32)
    x[1, :4000, 0] = synth((4000,), 3)
    kw = dict(n_fft=512, hop_length=128, sample_rate=16000, n_mels=40, return_decibel=True)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    want = o.kapre_melspectrogram(x, **kw)
    assert np.allclose(got[0], -50.0, atol=1e-4)         # all-silent item: 10 log10(amin)
    assert_db_close(got, want)


def test_amin_below_the_smallest_normal_float_is_raised_to_it():
    """Documented deviation: the float32 decibel kernels feed max(x, amin) to v_log_f32, which has no
    denormal support, so an amin below FLT_MIN (a floor under -379 dB; backend.py:186 takes any positive amin) acts as
    FLT_MIN.  Values at or above FLT_MIN are unaffected; the float64 layer keeps the caller's amin."""
    flt_min = np.float32(1.17549435e-38)
    x = np.array([0.0, 1e-42, float(flt_min), 1e-30, 1e-5, 1.0], np.float32).reshape(1, 1, 6, 1)
    lay = MagnitudeToDecibel(ref_value=1.0, amin=1e-42, dynamic_range=1000.0)
    got = to_np(lay(x)).reshape(-1)
    want = 10.0 * np.log10(np.maximum(x.reshape(-1).astype(np.float64), float(flt_min)))
    np.testing.assert_allclose(got, want, atol=2e-4)
    assert got[0] == got[1] == got[2] and abs(got[0] + 379.2978) < 1e-3
    lay64 = MagnitudeToDecibel(ref_value=1.0, amin=1e-42, dynamic_range=1000.0, dtype='float64')
    got64 = to_np(lay64(x.astype(np.float64))).reshape(-1)
    np.testing.assert_allclose(got64, 10.0 * np.log10(np.maximum(x.reshape(-1).astype(np.float64), 1e-42)), atol=1e-9)


@pytest.mark.parametrize("n_fft,hop,win", [(1000, 250, 1000), (512, 100, 400), (2048, 512, 2048),
                                           (1024, 256, 1024), (300, 75, 300), (256, 64, 256),
                                           # mixed-radix inverse (every plan of kpr_fft_mr.h)
                                           (400, 100, 400), (160, 40, 160), (200, 50, 150), (320, 80, 320),
                                           (640, 160, 400), (800, 200, 800),
                                           # two-pass plans (sizes with a factor 3)
                                           (96, 24, 96), (120, 30, 100), (192, 48, 192), (240, 60, 240),
                                           (360, 90, 300), (384, 96, 384), (480, 120, 480), (600, 150, 600),
                                           (720, 180, 512), (768, 192, 768), (960, 240, 960),
                                           # 4096 / 8192: sub-FFT inverse kernel
                                           (4096, 1024, 4096), (4096, 1000, 3000), (8192, 2048, 8192),
                                           # size-generic inverse FFT kernel (no tuned plan)
                                           (1001, 250, 1001), (1200, 300, 1200), (1536, 384, 1024), (2000, 500, 2000),
                                           (3000, 750, 2048), (77, 19, 77), (15, 4, 15), (1280, 320, 1280),
                                           (2049, 512, 2049)])
@pytest.mark.parametrize("fmt_in,fmt_out", [("channels_last", "channels_first"),
                                            ("channels_first", "channels_last")])
def test_istft_vs_oracle(n_fft, hop, win, fmt_in, fmt_out):
    rng = np.random.default_rng(n_fft + hop)
    k = n_fft // 2 + 1
    shape = (2, 11, k, 3) if fmt_in == "channels_last" else (2, 3, 11, k)
    s = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name="hamming_window",
              input_data_format=fmt_in, output_data_format=fmt_out)
    assert_close(to_np(InverseSTFT(**kw)(s)), o.kapre_istft(s, **kw))


# k_istft_ws (ring of frames, producer / consumer waves): every branch of its schedule.
#   (frames, n_fft, hop, win, batch, channels)
_WS_CASES = [
    (431, 1024, 256, 1024, 3, 1),     # R = 4, two or more segments per signal (few signals, many frames)
    (40, 2048, 512, 2048, 2, 2),      # ring of 16 rows, one frame per wave
    (61, 512, 256, 512, 5, 1),        # R = 2 (two rows per sample group)
    (33, 512, 64, 512, 2, 1),         # R = 8
    (57, 512, 160, 400, 2, 1),        # win < n_fft, hop not a power of two: partial consumer passes
    (25, 256, 256, 256, 3, 1),        # hop == win: R = 1, no overlap at all
    (9, 256, 64, 256, 300, 1),        # more segments than compute units: several per workgroup
    (3000, 512, 128, 512, 1, 1),      # one long signal: cut into ~190 segments to fill the GPU
    (1, 1024, 256, 1024, 2, 1),       # a single frame
    (3, 1024, 512, 1000, 2, 1),       # fewer frames than producer waves, win % hop != 0
    (200, 256, 32, 128, 1, 2),        # long run of tiny frames, eight per wave
    (19, 1024, 256, 1536, 2, 1),      # win_length > n_fft (zero-extended frames)
    # mixed-radix producers (k_istft_ws_mr), every plan
    (100, 400, 100, 400, 3, 1),       # n_fft 400: 10 lanes per frame, six frames per ticket
    (37, 1000, 250, 1000, 2, 2),      # three passes (20, 5, 5)
    (50, 800, 400, 800, 2, 1),        # R = 2
    (64, 160, 40, 160, 2, 1),         # sixteen frames per ticket
    (40, 200, 52, 200, 2, 1),         # win % hop != 0
    (30, 320, 80, 320, 2, 1),
    (30, 640, 160, 400, 2, 1),        # win < n_fft, R = 3
    (5, 400, 200, 400, 300, 1),       # fewer frames than one ticket, many segments per workgroup
    # two-pass producers (sizes with a factor 3)
    (60, 480, 120, 480, 2, 1), (30, 960, 240, 960, 2, 1), (50, 96, 24, 96, 2, 1), (40, 120, 60, 120, 3, 1),
    (20, 720, 180, 720, 2, 1), (25, 600, 300, 600, 2, 1), (45, 240, 60, 200, 2, 1), (30, 360, 92, 360, 2, 1),
    (40, 384, 96, 384, 2, 2), (40, 192, 48, 192, 2, 1), (33, 768, 192, 512, 2, 1),
    # hop even but not a multiple of 4 (the default hop n_fft / 4 of these sizes): pairs of samples per lane
    (37, 1000, 250, 1000, 2, 1), (30, 600, 150, 600, 2, 1), (30, 360, 90, 360, 2, 1), (40, 400, 50, 200, 3, 1),
    (20, 200, 50, 100, 2, 1),         # ... and two rows only: no such instance, two-kernel path
]


@pytest.mark.parametrize("frames,n_fft,hop,win,batch,ch", _WS_CASES)
def test_istft_ring_kernel(frames, n_fft, hop, win, batch, ch):
    rng = np.random.default_rng(frames * 7 + n_fft + hop)
    k = n_fft // 2 + 1
    s = (rng.standard_normal((batch, ch, frames, k)) + 1j * rng.standard_normal((batch, ch, frames, k))).astype(np.complex64)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name="hamming_window",
              input_data_format="channels_first", output_data_format="channels_first")
    from kapre_amd import _ffi
    try:
        _ffi.set_option("istft_path", 3)                          # the ring kernel whatever the launch size
        got = to_np(InverseSTFT(**kw)(s))
    finally:
        _ffi.set_option("istft_path", 0)
    want = o.kapre_istft(s, **kw)
    assert_close(got, want)
    np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)     # the automatic choice (small launches: barrier kernel)
    # and bit for bit what the barrier kernel and the two-kernel path (irFFT, then gather) produce:
    # the same frames summed in the same (ascending) order
    try:
        _ffi.set_option("istft_path", 1)                          # no wave-specialised ring kernel
        np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)
        _ffi.set_option("istft_path", 2)                          # irFFT, then overlap-add
        np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)
    finally:
        _ffi.set_option("istft_path", 0)


@pytest.mark.parametrize("frames,n_fft,hop,win,batch,ch", [
    (431, 1024, 256, 1024, 3, 1),     # cfg4's signal length: four segments per signal (halo frames), runs of 3 / 4 frames
    (100, 1024, 256, 1024, 2, 2),     # one segment, runs of exactly R - 1 = 3 frames and a few of 4
    (97, 1024, 256, 800, 2, 1),       # win < n_fft
    (120, 1024, 256, 1023, 2, 1),     # odd window: odd signal length, rows 4-byte aligned only
    (333, 1024, 256, 1024, 40, 1),    # more items than one round of workgroups may hold on a small part
    (50, 2048, 512, 2048, 5, 1),      # one frame per wave
    (205, 2048, 512, 2018, 2, 1),
    (200, 512, 128, 512, 3, 1),       # four frames per wave
    (777, 512, 128, 400, 2, 1),
    (64, 1024, 512, 1024, 4, 1),      # hop = n_fft / 2: two frames per sample
    (130, 2048, 1024, 2048, 2, 1),
    (230, 1024, 128, 1024, 2, 1),     # hop = n_fft / 8: eight frames per sample, runs of >= 7
    (2000, 1024, 128, 1000, 1, 1),    # ... long runs, several segments
    (120, 2048, 256, 2048, 2, 2),
    (470, 512, 64, 512, 1, 1),
    (70, 512, 256, 512, 3, 1),        # n_fft 512, hop = n_fft / 2
])
def test_istft_per_wave_kernel(frames, n_fft, hop, win, batch, ch):
    """k_istft_pw (overlap-add in registers, static frame runs per lane group): against the oracle, against the ring / barrier
    kernels (bit-identical away from run boundaries, a rounding or two at them), and bit-identical from call to call."""
    from kapre_amd import _ffi
    rng = np.random.default_rng(frames * 11 + n_fft + hop)
    k = n_fft // 2 + 1
    s = (rng.standard_normal((batch, ch, frames, k)) + 1j * rng.standard_normal((batch, ch, frames, k))).astype(np.complex64)
    s *= np.logspace(-2, 0, batch, dtype=np.float32).reshape(batch, 1, 1, 1)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name="hann_window",
              input_data_format="channels_first", output_data_format="channels_first")
    try:
        _ffi.set_option("istft_path", 4)
        got = to_np(InverseSTFT(**kw)(s))
        assert "k_istft_pw" in _ffi.last_launches(), _ffi.last_launches()
        for _ in range(3):
            np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)
        _ffi.set_option("istft_path", 1)
        ref = to_np(InverseSTFT(**kw)(s))
    finally:
        _ffi.set_option("istft_path", 0)
    assert got.shape == ref.shape
    want = o.kapre_istft(s, **kw)
    assert_close(got, want)
    scale = np.abs(ref).reshape(batch, -1).max(axis=1).reshape(batch, 1, 1)
    assert (np.abs(got - ref) <= 4e-7 * scale).all(), float((np.abs(got - ref) / scale).max())
    assert (got == ref).mean() > 0.3                              # interior samples: the same sums in the same order


@pytest.mark.parametrize("fmt_in, fmt_out", [("channels_last", "channels_last"), ("channels_last", "channels_first"),
                                             ("channels_first", "channels_last")])
@pytest.mark.parametrize("frames,n_fft,hop,win,batch,ch", [
    (434, 1024, 256, 1024, 3, 2),     # stereo: the two lane groups of a wave = the two channels of one frame run
    (97, 1024, 256, 800, 2, 4),       # four channels: eight frame runs per workgroup, win < n_fft
    (60, 1024, 512, 1023, 2, 2),      # hop = n_fft / 2, odd window
    (150, 2048, 512, 2048, 2, 2),     # one frame per wave: neighbouring waves = channels
    (64, 2048, 1024, 2048, 1, 8),
    (300, 512, 128, 512, 2, 2),       # four lane groups per wave = two runs x two channels
    (210, 512, 256, 400, 3, 4),
])
def test_istft_per_wave_kernel_interleaved(frames, n_fft, hop, win, batch, ch, fmt_in, fmt_out):
    """k_istft_pw's interleaved instances (channels_last with a power-of-two channel count on either side): streams =
    (frame run, channel), element strides C; against the oracle and the barrier kernel, identical from call to call."""
    from kapre_amd import _ffi
    rng = np.random.default_rng(frames * 13 + n_fft + ch)
    k = n_fft // 2 + 1
    shape = (batch, frames, k, ch) if fmt_in == "channels_last" else (batch, ch, frames, k)
    s = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name="hann_window",
              input_data_format=fmt_in, output_data_format=fmt_out)
    try:
        _ffi.set_option("istft_path", 4)
        got = to_np(InverseSTFT(**kw)(s))
        assert "k_istft_pw_il" in _ffi.last_launches(), _ffi.last_launches()
        for _ in range(2):
            np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)
        _ffi.set_option("istft_path", 1)
        ref = to_np(InverseSTFT(**kw)(s))
    finally:
        _ffi.set_option("istft_path", 0)
    assert_close(got, o.kapre_istft(s, **kw))
    assert (np.abs(got - ref) <= 4e-7 * np.abs(ref).max()).all()
    assert (got == ref).mean() > 0.3


@pytest.mark.parametrize("frames,n_fft,hop,batch,ch,fmt", [
    (20, 2048, 1024, 300, 1, "channels_first"),      # 300 items on 256 workgroups (S = 8: one partial block per run)
    (200, 2048, 1024, 24, 6, "channels_first"),      # the configuration tools/fuzz_parity.py failed on: 432 items
    (50, 2048, 512, 280, 1, "channels_first"),
    (100, 1024, 256, 150, 2, "channels_first"),      # 300 signals, one segment each
    (100, 1024, 256, 300, 2, "channels_last"),       # interleaved instances, 300 items
    (200, 512, 128, 290, 1, "channels_last"),
])
def test_istft_per_wave_kernel_more_items_than_workgroups(frames, n_fft, hop, batch, ch, fmt):
    """k_istft_pw when a workgroup walks several items (regression, round 4: the LDS stashes of the partial head blocks were
    reused by a stream that was ahead before its predecessor had read the previous item's)."""
    from kapre_amd import _ffi
    rng = np.random.default_rng(frames + n_fft + batch)
    k = n_fft // 2 + 1
    shape = (batch, frames, k, ch) if fmt == "channels_last" else (batch, ch, frames, k)
    s = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    kw = dict(n_fft=n_fft, hop_length=hop, input_data_format=fmt, output_data_format=fmt)
    try:
        _ffi.set_option("istft_path", 4)
        got = to_np(InverseSTFT(**kw)(s))
        assert "k_istft_pw" in _ffi.last_launches(), _ffi.last_launches()
        for _ in range(3):
            np.testing.assert_array_equal(to_np(InverseSTFT(**kw)(s)), got)
        _ffi.set_option("istft_path", 1)
        ref = to_np(InverseSTFT(**kw)(s))
    finally:
        _ffi.set_option("istft_path", 0)
    assert (np.abs(got - ref) <= 4e-7 * np.abs(ref).max()).all(), float(np.abs(got - ref).max() / np.abs(ref).max())
    assert_close(got[:8], o.kapre_istft(s[:8], **kw))


def test_log_frequency_spectrogram_vs_oracle():
    x = speech(8000)[None, :, None].repeat(2, axis=0) * np.array([1.0, 0.3], np.float32).reshape(2, 1, 1)
    kw = dict(n_fft=2048, hop_length=512, sample_rate=22050, return_decibel=True)
    got = to_np(composed.get_log_frequency_spectrogram_layer(**kw)(x))
    s = o.magnitude(o.kapre_stft(x, 2048, None, 512))
    fb = o.filterbank_log(22050, 1025)
    want = o.magnitude_to_decibel(o.apply_filterbank(s, fb, "channels_last"))
    assert got.shape == want.shape == (2, 12, 84, 1)
    assert_db_close(got, want)


@pytest.mark.parametrize("n_fft, hop, sr", [(400, 160, 16000), (480, 120, 16000), (256, 64, 22050), (1000, 250, 22050)])
def test_log_frequency_spectrogram_on_the_fused_kernels_of_round_3(n_fft, hop, sr):
    """Log-frequency banks are wider than mel banks (more chunks per filter tile): through k_mel_mr / k_mel_ts<128>, and a
    DENSE matrix of the same shape as well (every tile spans all rows: if the schedule does not fit, the call must fall
    back to the two-launch path, never fail)."""
    x = synth((3, n_fft + 57 * hop, 2), 5 + n_fft)
    kw = dict(n_fft=n_fft, hop_length=hop, sample_rate=sr, return_decibel=True)
    got = to_np(composed.get_log_frequency_spectrogram_layer(**kw)(x))
    s = o.magnitude(o.kapre_stft(x, n_fft, None, hop))
    fb = o.filterbank_log(sr, n_fft // 2 + 1)
    want = o.magnitude_to_decibel(o.apply_filterbank(s, fb, "channels_last"))
    assert got.shape == want.shape
    assert_db_close(got, want)
    rng = np.random.default_rng(n_fft)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=n_fft // 2 + 1, n_mels=8))
    layer.filterbank = rng.standard_normal((n_fft // 2 + 1, 96)).astype(np.float32)
    dense = to_np(Sequential([STFT(n_fft=n_fft, hop_length=hop), Magnitude(), layer])(x))
    assert_close(dense, o.apply_filterbank(s, layer.filterbank, "channels_last"))


def test_stft_mag_phase_layer():
    x = synth((2, 6000, 2), 21)
    got = to_np(composed.get_stft_mag_phase((6000, 2), n_fft=512, hop_length=256)(x))
    s = o.kapre_stft(x, 512, None, 256)
    assert got.shape == (2, 22, 257, 4)
    assert_close(got[..., :2], np.abs(s))
    strong = np.abs(s) > 1e-2 * np.abs(s).max()
    np.testing.assert_allclose(np.cos(got[..., 2:])[strong], np.cos(np.angle(s))[strong], atol=1e-3)


def test_dense_filterbank_without_structure():
    """ApplyFilterbank must be correct for ANY matrix, not only banded mel ones."""
    rng = np.random.default_rng(0)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=257, n_mels=37))
    layer.filterbank = rng.standard_normal((257, 37)).astype(np.float32)
    layer._kranges = None
    x = rng.uniform(0, 2, (3, 9, 257, 2)).astype(np.float32)
    want = o.apply_filterbank(x, layer.filterbank, "channels_last")
    assert_close(to_np(layer(x)), want, rel=2e-6)
    # and inside the fused kernel
    st = STFT(n_fft=512, hop_length=128)
    wav = synth((2, 4000, 2), 4)
    fused = to_np(Sequential([st, Magnitude(), layer])(wav))
    assert_close(fused, o.apply_filterbank(np.abs(o.kapre_stft(wav, 512, None, 128)), layer.filterbank, "channels_last"))


def test_torch_inputs_streams_and_dtypes():
    import torch

    x = torch.from_numpy(synth((2, 9000, 1), 77))
    mel = composed.get_melspectrogram_layer(n_fft=1024, hop_length=256, n_mels=64)
    a = mel(x)                       # CPU tensor in -> GPU tensor out
    assert a.is_cuda and a.dtype == torch.float32
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        b = mel(x.cuda())
    side.synchronize()
    assert torch.equal(a, b)
    c = mel(x.double())              # float64 input is computed in float32 (floatx)
    assert torch.equal(a, c)


# ------------------------------------------------------------------ scheduling edge cases of k_mel_ws
# n_fft = 2048 runs on the wave-specialised kernel: frames are handed out by tickets, tiles may be
# short, workgroups get runs cut at 8-frame granularity.  These sizes hit: fewer frames than
# producer waves, exactly one round, one frame into a second tile, short last tiles, several
# workgroups with unequal runs, multi-channel frame numbering, filter counts off the 16-grid,
# short analysis windows and zero-padded edge frames.
@pytest.mark.parametrize("batch,frames,ch,fmt,n_mels,win,pad_end,db", [
    (1, 1, 1, "channels_last", 128, None, False, False),
    (1, 7, 1, "channels_last", 128, None, False, True),
    (1, 8, 1, "channels_first", 40, None, False, False),
    (1, 9, 2, "channels_last", 128, None, False, True),
    (2, 17, 1, "channels_last", 130, 1500, False, False),
    (3, 33, 3, "channels_first", 64, None, True, True),
    (5, 100, 1, "channels_last", 128, 2048, True, False),
    (37, 21, 2, "channels_last", 96, None, False, True),
])
def test_mel_ws_schedules(batch, frames, ch, fmt, n_mels, win, pad_end, db):
    import os
    import torch

    n_fft, hop = 2048, 512
    t = n_fft + (frames - 1) * hop - (137 if pad_end else 0)      # pad_end: last frame partly padding
    shape = (batch, t, ch) if fmt == "channels_last" else (batch, ch, t)
    x = synth(shape, 4242 + frames)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, sample_rate=44100, n_mels=n_mels, pad_end=pad_end,
              return_decibel=db, input_data_format=fmt, output_data_format=fmt)
    layer = composed.get_melspectrogram_layer(**kw)
    got = layer(x)
    want = o.kapre_melspectrogram(x, **kw)
    (assert_db_close if db else assert_close)(to_np(got), want)
    for _ in range(3):                                            # ticket order varies, results must not
        assert torch.equal(layer(x), got)
    from kapre_amd import _ffi
    _ffi.set_option("mel_variant", 4)                             # the tile-synchronous MFMA kernel: same values to round-off
    try:
        ts = composed.get_melspectrogram_layer(**kw)(x)
    finally:
        _ffi.set_option("mel_variant", 0)
    (assert_db_close if db else assert_close)(to_np(ts), want)


@pytest.mark.parametrize("n_fft, hop, ch, win", [
    (2048, 512, 6, None),      # three channel pairs per frame index
    (2048, 700, 3, 2018),      # odd channel count (no PAIR form), short window
    (2048, 512, 2, None),      # stereo at 64 lanes per frame
    (1024, 256, 3, None),      # 32 lanes per frame, two frames (of different channels) per wave
    (1024, 160, 2, 800),       # 32 lanes, stereo: the stereo pair fetch of the plain kernel / the PAIR form (variant 8)
    (512, 128, 5, None),       # 16 lanes per frame
    (2048, 512, 4, None),      # PAIR form of k_mel_pw (variant 8): channel pairs (0, 1), (2, 3)
    (1024, 256, 6, None),      # PAIR form, two pairs per wave
    (1024, 160, 4, 800),       # PAIR form, short window, hop not a multiple of anything
])
def test_interleaved_waveforms(n_fft, hop, ch, win):
    """channels_last waveforms with C > 1 through the per-wave mel kernel (plain, stereo pair fetch, PAIR form) and the STFT,
    with padded edge frames on both sides; channels_first of the same data must give the same numbers bit for bit (same
    arithmetic, different loads)."""
    import torch
    from kapre_amd import _ffi

    batch, frames = 5, 23
    t = n_fft + (frames - 1) * hop - 91
    x = synth((batch, t, ch), 4242 + n_fft + ch)
    x *= np.logspace(-1, 0, batch, dtype=np.float32).reshape(batch, 1, 1)
    xt = np.ascontiguousarray(x.transpose(0, 2, 1))
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, pad_begin=True, pad_end=True)
    want = o.kapre_stft(x, input_data_format="channels_last", output_data_format="channels_last", **kw)
    old = _ffi.set_option("stft_variant", 3)
    try:
        got = STFT(input_data_format="channels_last", output_data_format="channels_last", **kw)(x)
        same = STFT(input_data_format="channels_first", output_data_format="channels_last", **kw)(xt)
    finally:
        _ffi.set_option("stft_variant", old)
    assert_close(to_np(got), want)
    assert torch.equal(got, same)
    mkw = dict(sample_rate=44100, n_mels=96, return_decibel=True, **kw)
    wantm = o.kapre_melspectrogram(x, input_data_format="channels_last", output_data_format="channels_last", **mkw)
    for variant in (0, 5, 7, 8):                          # (8: the PAIR form of k_mel_pw where it applies -- even C, n_fft >= 1024)
        old = _ffi.set_option("mel_variant", variant)
        try:
            gm = composed.get_melspectrogram_layer(input_data_format="channels_last", output_data_format="channels_last", **mkw)(x)
            sm = composed.get_melspectrogram_layer(input_data_format="channels_first", output_data_format="channels_last", **mkw)(xt)
        finally:
            _ffi.set_option("mel_variant", old)
        assert_db_close(to_np(gm), wantm)
        assert torch.equal(gm, sm)
        old = _ffi.set_option("mel_variant", variant)
        try:                                                      # interleaved waveform in, channels_first spectrogram out
            tm = composed.get_melspectrogram_layer(input_data_format="channels_last", output_data_format="channels_first", **mkw)(x)
        finally:
            _ffi.set_option("mel_variant", old)
        assert torch.equal(tm, gm.permute(0, 3, 1, 2))


@pytest.mark.parametrize("n_fft, hop, batch, frames, ch, fmt, n_mels, win, pad_end, db", [
    (2048, 512, 256, 83, 1, "channels_last", 128, None, False, False),     # the north-star shape: 41.5 frames per workgroup
    (2048, 512, 48, 87, 1, "channels_last", 128, None, True, True),        # 13..19 tickets per workgroup (SKEW tickets)
    (2048, 1024, 7, 45, 6, "channels_first", 128, None, False, True),      # items change inside tiles: running dB statistics
    (2048, 512, 3, 5, 2, "channels_last", 40, 2018, True, False),          # fewer frames than a round, short window, odd tail
    (1024, 160, 33, 97, 1, "channels_last", 80, None, False, False),       # two frames per wave (G = 2)
    (1024, 256, 5, 33, 2, "channels_first", 96, 800, True, True),
    (512, 128, 64, 169, 2, "channels_last", 40, None, False, True),        # the reference's own test shape (four frames per wave)
    (512, 256, 9, 20, 1, "channels_first", 128, 400, True, False),
    (256, 64, 21, 130, 1, "channels_last", 40, None, False, True),         # n_fft 256: eight frames per wave, 64-frame rounds
    (256, 128, 4, 33, 3, "channels_first", 64, 200, True, False),
])
@pytest.mark.parametrize("variant", [0, 3, 4, 5, 6, 7])
def test_mel_kernel_variants(variant, n_fft, hop, batch, frames, ch, fmt, n_mels, win, pad_end, db):
    """Every fused mel kernel (0 = the default choice, 3 = k_mel_ws / ring, 4 = k_mel_ts, 5 / 6 / 7 = the per-wave kernel
    k_mel_pw with 8 / 4 / 16 waves per workgroup) against the oracle on shapes that exercise its scheduling edge cases;
    repeated calls must be bit-identical (deterministic partial-sum order)."""
    import torch
    from kapre_amd import _ffi

    t = n_fft + (frames - 1) * hop - (137 if pad_end else 0)
    shape = (batch, t, ch) if fmt == "channels_last" else (batch, ch, t)
    x = synth(shape, 777 + frames + n_fft)
    x *= np.logspace(-2, 0, batch, dtype=np.float32).reshape(batch, 1, 1)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, sample_rate=44100, n_mels=n_mels, pad_end=pad_end,
              return_decibel=db, input_data_format=fmt, output_data_format=fmt)
    old = _ffi.set_option("mel_variant", variant)
    try:
        layer = composed.get_melspectrogram_layer(**kw)
        got = layer(x)
        for _ in range(2):
            assert torch.equal(layer(x), got)
    finally:
        _ffi.set_option("mel_variant", old)
    want = o.kapre_melspectrogram(x, **kw)
    g = to_np(got)
    if db:
        assert_db_close(g, want)
    else:                                   # every batch item against its own scale
        n = g.shape[0]
        err = np.abs(g - want).reshape(n, -1).max(axis=1)
        scale = np.abs(want).reshape(n, -1).max(axis=1)
        assert (err <= 1e-4 * scale + 1e-30).all(), float((err / np.maximum(scale, 1e-30)).max())


@pytest.mark.parametrize("n_fft, hop, sr, batch, frames, ch, fmt, n_mels, win, pad_end, db", [
    (400, 160, 16000, 64, 101, 1, "channels_last", 80, None, False, False),     # the speech front end: six frames per wave, 24-frame rounds
    (400, 160, 16000, 5, 77, 2, "channels_last", 80, None, True, True),         # items change inside tiles, ragged tail
    (400, 100, 16000, 3, 50, 2, "channels_first", 40, 320, True, True),         # short window
    (400, 160, 16000, 2, 3, 1, "channels_first", 128, None, False, False),      # fewer frames than a wave's group
    (320, 160, 16000, 9, 130, 1, "channels_last", 64, None, False, True),       # eight frames per wave
    (640, 320, 16000, 7, 61, 1, "channels_first", 80, 512, False, False),       # four frames per wave
    (1000, 250, 22050, 4, 40, 2, "channels_last", 96, None, True, True),        # two frames per wave, 25-lane frames
    (200, 80, 8000, 17, 200, 1, "channels_last", 40, None, False, False),       # twelve frames per wave
    (160, 80, 8000, 6, 333, 3, "channels_first", 40, None, True, True),         # sixteen frames per wave
    (800, 200, 16000, 5, 55, 1, "channels_last", 64, None, False, True),        # three frames per wave: 12-frame rounds, one partial tile
    (400, 160, 16000, 1, 1, 1, "channels_last", 80, None, False, False),        # a single frame
    # sizes with a factor 3 (TwoPassFft<N1, N2>: the lane count changes between the two passes)
    (480, 160, 16000, 12, 90, 1, "channels_last", 80, None, False, False),      # <16, 15>: four frames per wave
    (960, 480, 48000, 5, 41, 2, "channels_first", 128, None, True, True),       # <20, 24>: two frames per wave (48 lanes)
    (96, 48, 8000, 7, 300, 1, "channels_last", 23, None, False, True),          # <8, 6>: eight frames per wave, exchange row > N
    (192, 64, 8000, 3, 111, 2, "channels_last", 40, 160, True, False),          # <8, 12>: five frames per wave, 20-frame rounds
    (600, 150, 16000, 4, 37, 1, "channels_first", 64, None, False, True),       # <20, 15>: three frames per wave
    (720, 240, 22050, 2, 29, 3, "channels_last", 80, 512, True, False),         # <15, 24>
    (120, 40, 8000, 9, 64, 1, "channels_first", 20, None, False, False),        # <4, 15>: fewer points per input lane than lanes
    (384, 128, 16000, 6, 50, 1, "channels_last", 64, None, True, True),         # <16, 12>
])
@pytest.mark.parametrize("variant", [0, 3])
def test_mixed_radix_mel_kernel(variant, n_fft, hop, sr, batch, frames, ch, fmt, n_mels, win, pad_end, db):
    """k_mel_mr (variant 0: one launch for every n_fft with a mixed-radix or two-pass plan) and the two-launch path it
    replaces (variant 3) against the oracle; repeated calls bit-identical."""
    import torch
    from kapre_amd import _ffi

    t = n_fft + (frames - 1) * hop - (37 if pad_end else 0)
    shape = (batch, t, ch) if fmt == "channels_last" else (batch, ch, t)
    x = synth(shape, 991 + frames + n_fft)
    x *= np.logspace(-2, 0, batch, dtype=np.float32).reshape(batch, 1, 1)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, sample_rate=sr, n_mels=n_mels, pad_end=pad_end,
              return_decibel=db, input_data_format=fmt, output_data_format=fmt)
    old = _ffi.set_option("mel_variant", variant)
    try:
        layer = composed.get_melspectrogram_layer(**kw)
        got = layer(x)
        for _ in range(2):
            assert torch.equal(layer(x), got)
    finally:
        _ffi.set_option("mel_variant", old)
    want = o.kapre_melspectrogram(x, **kw)
    g = to_np(got)
    assert g.shape == want.shape
    if db:
        assert_db_close(g, want)
    else:
        n = g.shape[0]
        err = np.abs(g - want).reshape(n, -1).max(axis=1)
        scale = np.abs(want).reshape(n, -1).max(axis=1)
        assert (err <= 1e-4 * scale + 1e-30).all(), float((err / np.maximum(scale, 1e-30)).max())


# ------------------------------------------------------------------ Frame / Energy / Delta / MFCC
from kapre_amd import Frame, Energy, LogmelToMFCC, Delta  # noqa: E402


@pytest.mark.parametrize("name", golden_names("frame"))
def test_frame_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    got = to_np(Frame(**kw)(x))
    assert got.shape == y.shape
    np.testing.assert_array_equal(got, y.astype(np.float32))      # a copy: bit exact (reference: assert_equal)


@pytest.mark.parametrize("name", golden_names("energy"))
def test_energy_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    assert_close(to_np(Energy(**kw)(x)), y, rel=2e-6)
    np.testing.assert_allclose(to_np(Energy(**kw)(x)), y, atol=1e-5 * max(1.0, float(np.abs(y).max())))


@pytest.mark.parametrize("name", golden_names("delta"))
def test_delta_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    got = to_np(Delta(**kw)(x))
    assert got.shape == y.shape
    np.testing.assert_allclose(got, y, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", golden_names("logmel_to_mfcc"))
def test_mfcc_golden(golden, name):
    kw, x, y, _ = golden.get(name)
    got = to_np(LogmelToMFCC(**kw)(x))
    assert got.shape == y.shape
    assert_close(got, y, rel=2e-6)
    np.testing.assert_allclose(got, y, atol=1e-4 * max(1.0, float(np.abs(y).max()) / 100))   # upstream atol 1e-4


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_frame_energy_delta_mfcc_vs_oracle_larger(fmt):
    """BASELINE-sized inputs: cfg5's waveform (10 s @ 16 kHz) framed 400/160, its energy, and the
    delta / MFCC of an 80-band log-mel of the same length."""
    b, t, c = 8, 160000, 2
    x = synth((b, t, c) if fmt == "channels_last" else (b, c, t), 77)
    fr = to_np(Frame(400, 160, pad_end=True, pad_value=-1.5, data_format=fmt)(x))
    np.testing.assert_array_equal(fr, o.kapre_frame(x, 400, 160, True, -1.5, fmt).astype(np.float32))
    en = to_np(Energy(16000, 0.05, 400, 160, data_format=fmt)(x))
    assert_close(en, o.kapre_energy(x, 16000, 0.05, 400, 160, data_format=fmt), rel=2e-6)
    shp = (b, 998, 80, c) if fmt == "channels_last" else (b, c, 998, 80)
    lm = (synth(shp, 78) * 30 - 40).astype(np.float32)
    for win, mode in ((5, "symmetric"), (9, "reflect"), (3, "constant")):
        got = to_np(Delta(win, mode, data_format=fmt)(lm))
        np.testing.assert_allclose(got, o.kapre_delta(lm, win, mode, fmt), rtol=1e-5, atol=2e-5)
    got = to_np(LogmelToMFCC(13, data_format=fmt)(lm))
    assert_close(got, o.kapre_logmel_to_mfcc(lm, 13, fmt), rel=2e-6)


def test_mel_db_mfcc_delta_chain():
    """the usual speech front end assembled from the layers: log-mel -> MFCC -> delta"""
    x = synth((3, 16000, 1), 5)
    kw = dict(n_fft=512, hop_length=160, sample_rate=16000, n_mels=40, return_decibel=True)
    model = Sequential([composed.get_melspectrogram_layer(**kw), LogmelToMFCC(13), Delta(9)])
    got = to_np(model(x))
    ref = o.kapre_delta(o.kapre_logmel_to_mfcc(o.kapre_melspectrogram(x, **kw), 13), 9)
    assert got.shape == ref.shape == (3, 97, 13, 1)
    np.testing.assert_allclose(got, ref, atol=2e-3)      # dB in, DCT gain ~ sqrt(2*40): 1e-4 relative of ~100


def test_frame_edge_cases():
    import torch

    assert tuple(Frame(64, 32)(synth((2, 10, 1), 1)).shape) == (2, 0, 64, 1)       # shorter than a frame
    assert tuple(Frame(64, 32, pad_end=True)(synth((2, 10, 1), 1)).shape) == (2, 1, 64, 1)
    assert tuple(Energy(frame_length=64, hop_length=32)(torch.zeros(0, 100, 1)).shape) == (0, 2, 1)
    one = to_np(Delta(3)(np.ones((1, 1, 4, 1), np.float32)))                        # single frame: all pads mirror it
    assert one.shape == (1, 1, 4, 1) and not one.any()


@pytest.mark.parametrize("fmt,ch", [("channels_last", 1), ("channels_first", 2), ("channels_last", 3)])
@pytest.mark.parametrize("n_freq,n_mels,sr", [(1025, 128, 44100), (513, 80, 16000), (257, 96, 22050)])
def test_apply_filterbank_standalone_wide(fmt, ch, n_freq, n_mels, sr):
    """Stand-alone ApplyFilterbank with a wide mel bank: contiguous rows run on the fused kernel's
    MFMA consumers fed by loader waves (kpr_apply_filterbank_packed_f32), channels_last with several
    channels on the generic GEMM -- same numbers either way."""
    rng = np.random.default_rng(n_freq + ch)
    frames = 37
    shp = (3, frames, n_freq, ch) if fmt == "channels_last" else (3, ch, frames, n_freq)
    x = (rng.uniform(0, 4, shp) ** 3).astype(np.float32)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=n_freq, n_mels=n_mels),
                            data_format=fmt)
    got = to_np(layer(x))
    want = o.apply_filterbank(x, o.filterbank_mel(sr, n_freq, n_mels), fmt)
    assert_close(got, want, rel=2e-6)
    import torch
    assert torch.equal(layer(x), layer(x))


# ------------------------------------------------------------------ n_fft 4096 / 8192: sub-FFT kernel k_stft_big
@pytest.mark.parametrize("n_fft,win,hop", [(4096, 4096, 1024), (4096, 3000, 1000), (8192, 8192, 2048), (8192, 4097, 4096)])
@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_stft_big_transform_sizes(n_fft, win, hop, fmt):
    """two / four sub-FFTs of 1024 points per frame; complex, magnitude and phase, both layouts, padding"""
    t = 3 * n_fft + 123
    shape = (2, t, 2) if fmt == "channels_last" else (2, 2, t)
    x = synth(shape, n_fft)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=True, pad_end=True,
              input_data_format=fmt, output_data_format=fmt)
    want = o.kapre_stft(x, n_fft, win, hop, None, True, True, fmt, fmt)
    assert_close(to_np(STFT(**kw)(x)), want, rel=2e-5)
    assert_close(to_np(Sequential([STFT(**kw), Magnitude()])(x)), np.abs(want), rel=2e-5)
    ph = to_np(Sequential([STFT(**kw), Phase()])(x))
    big = np.abs(want) > 1e-2 * np.abs(want).max()
    assert np.abs(np.angle(np.exp(1j * (ph - np.angle(want))))[big]).max() < 2e-3


@pytest.mark.parametrize("n_fft,ch,fmt,db", [(4096, 1, "channels_last", True), (4096, 2, "channels_first", False),
                                              (8192, 3, "channels_last", True), (4096, 2, "channels_last", False)])
def test_mel_big_transform_size(n_fft, ch, fmt, db):
    """n_fft 4096 / 8192: |X| rows from the sub-FFT kernel, then the banded filterbank kernel (K = 2049 / 4097
    bins are beyond the MFMA consumers' tile); items of different batch entries share a 4-row step"""
    shape = (3, 5 * n_fft + 77, ch) if fmt == "channels_last" else (3, ch, 5 * n_fft + 77)
    x = synth(shape, n_fft)
    kw = dict(n_fft=n_fft, hop_length=n_fft // 4, sample_rate=44100, n_mels=128, return_decibel=db,
              input_data_format=fmt, output_data_format=fmt)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    want = o.kapre_melspectrogram(x, **kw)
    (assert_db_close if db else assert_close)(got, want)


# ------------------------------------------------------------------ even non-power-of-two n_fft: Bluestein STFT
@pytest.mark.parametrize("n_fft,win,hop", [(400, 400, 160), (1000, 1000, 250), (1000, 512, 256), (300, 300, 75),
                                            (480, 400, 120), (12, 12, 4), (100, 64, 10), (1022, 1022, 511),
                                            # 2^a 5^b: the mixed-radix kernel (every plan of kpr_fft_mr.h)
                                            (160, 160, 40), (200, 150, 50), (320, 320, 80), (640, 400, 160),
                                            (800, 800, 200),
                                            # sizes with a factor 3: the two-pass plans (TwoPassFft<N1, N2>)
                                            (96, 96, 24), (120, 100, 30), (192, 192, 48), (240, 240, 60),
                                            (360, 300, 90), (384, 384, 96), (600, 600, 150), (720, 512, 180),
                                            (768, 768, 192), (960, 960, 240),
                                            # no tuned plan: the size-generic run-time mixed-radix kernel (odd sizes,
                                            # sizes above 1024) -- 1001 = 7 11 13, 1200, 1536, 2000, 3000, 77, 15, 1155
                                            (1001, 1001, 250), (1200, 1200, 300), (1536, 1024, 384), (2000, 2000, 500),
                                            (3000, 2048, 750), (77, 77, 19), (15, 15, 4), (1155, 1000, 289),
                                            (1280, 1280, 320), (6000, 4410, 1500),
                                            # ... at sizes whose twiddle table does not fit in LDS next to the frame buffers
                                            (12000, 12000, 3000), (16384, 16384, 4096),
                                            # a prime factor above 64 (2049 = 3 x 683): still the DFT-as-GEMM path
                                            (2049, 2049, 512)])
@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_stft_non_power_of_two(n_fft, win, hop, fmt):
    """n_fft = 2^a 3^b 5^c ... (the reference tests use 1000): mixed-radix FFT for 2^a 5^b sizes,
    chirp-z on the power-of-two FFT for the rest.  Complex, magnitude and phase outputs, both
    layouts, padding on both sides."""
    t = 6 * n_fft + 37
    shape = (2, t, 2) if fmt == "channels_last" else (2, 2, t)
    x = synth(shape, n_fft)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=True, pad_end=True,
              input_data_format=fmt, output_data_format=fmt)
    want = o.kapre_stft(x, n_fft, win, hop, None, True, True, fmt, fmt)
    got = to_np(STFT(**kw)(x))
    assert_close(got, want, rel=2e-5)
    mag = to_np(Sequential([STFT(**kw), Magnitude()])(x))
    assert_close(mag, np.abs(want), rel=2e-5)
    ph = to_np(Sequential([STFT(**kw), Phase()])(x))
    big = np.abs(want) > 1e-2 * np.abs(want).max()                 # phase is ill-conditioned near zero
    dphi = np.angle(np.exp(1j * (ph - np.angle(want))))
    assert np.abs(dphi[big]).max() < 2e-3


@pytest.mark.parametrize("n_fft,hop,ch,fmt,db", [(1200, 300, 1, "channels_last", True), (1001, 250, 2, "channels_first", False),
                                                  (1536, 512, 3, "channels_last", True), (3000, 1000, 1, "channels_last", False)])
def test_mel_generic_transform_sizes(n_fft, hop, ch, fmt, db):
    """melspectrogram at sizes served by the size-generic FFT kernel (two launches: STFT into the workspace, then
    the |.| x filterbank product with the decibel epilogue)."""
    t = 5 * n_fft + 123
    shape = (3, t, ch) if fmt == "channels_last" else (3, ch, t)
    x = synth(shape, n_fft + 1)
    kw = dict(n_fft=n_fft, hop_length=hop, sample_rate=22050, n_mels=64, return_decibel=db,
              input_data_format=fmt, output_data_format=fmt)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    want = o.kapre_melspectrogram(x, **kw)
    (assert_db_close if db else assert_close)(got, want)


@pytest.mark.parametrize("n_mels", [23, 40, 64, 80])
def test_mel_non_power_of_two_speech_front_end(n_mels):
    """the 25 ms / 10 ms speech front end: n_fft = 400, hop = 160, dB; 80 bands and the narrow banks
    (23 / 40 / 64 filters on 201 bins: also through the MFMA consumers, not the generic GEMM)"""
    x = synth((4, 16000, 1), 400)
    kw = dict(n_fft=400, hop_length=160, sample_rate=16000, n_mels=n_mels, return_decibel=True)
    got = to_np(composed.get_melspectrogram_layer(**kw)(x))
    assert got.shape == (4, 98, n_mels, 1)
    assert_db_close(got, o.kapre_melspectrogram(x, **kw))


def test_apply_filterbank_standalone_shapes_sweep():
    """the stand-alone filterbank instance of k_mel_ws (4 loader waves + two consumer groups on alternate
    tiles): row counts around the tile / workgroup boundaries, short and long rows, few and many filters"""
    rng = np.random.default_rng(7)
    for n_freq, n_mels in ((65, 16), (201, 17), (257, 80), (513, 1), (1025, 200), (129, 128)):
        fb = o.filterbank_mel(16000, n_freq, n_mels)
        layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=16000, n_freq=n_freq, n_mels=n_mels),
                                data_format="channels_first")
        for rows in (1, 15, 16, 17, 31, 33, 255, 257, 4099):
            x = rng.uniform(0, 3, (1, 1, rows, n_freq)).astype(np.float32)
            assert_close(to_np(layer(x)), o.apply_filterbank(x, fb, "channels_first"))


@pytest.mark.parametrize("n_freq,n_mels", [(201, 40), (257, 23), (513, 64), (1025, 10), (2049, 128), (4097, 40)])
def test_apply_filterbank_standalone_narrow(n_freq, n_mels):
    """narrow banks on rows of an odd number of bins (K = n_fft/2 + 1) are not thin-GEMM material"""
    rng = np.random.default_rng(n_freq + n_mels)
    x = rng.uniform(0, 3, (3, 2, 37, n_freq)).astype(np.float32)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=16000, n_freq=n_freq, n_mels=n_mels),
                            data_format="channels_first")
    fb = o.filterbank_mel(16000, n_freq, n_mels)
    assert_close(to_np(layer(x)), o.apply_filterbank(x, fb, "channels_first"))
    if n_freq <= 1025:
        # channels_last with several channels must equal the channels_first result of the SAME kernel bit for bit: by default
        # both layouts take k_fb_pw when the bank has a band plan (two interleaved channels: its ST instances, the same sums in the
        # same order); under "fb_variant" 1 the MFMA kernel, whose loader waves read rows strided by C (every row-length class
        # of ws_loader)
        xl = np.ascontiguousarray(x.transpose(0, 2, 3, 1))
        ll = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=16000, n_freq=n_freq, n_mels=n_mels),
                             data_format="channels_last")
        from kapre_amd import _ffi
        for variant in (0, 1):
            prev = _ffi.set_option("fb_variant", variant)
            try:
                got_cl = to_np(ll(xl)).transpose(0, 3, 1, 2)
                assert_close(got_cl, o.apply_filterbank(x, fb, "channels_first"))
                assert np.array_equal(got_cl, to_np(layer(x))), variant
            finally:
                _ffi.set_option("fb_variant", prev)


# ------------------------------------------------------------------ randomised configurations
def _random_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        n_fft = int(rng.choice([256, 512, 1024, 2048, 400, 1000, 300, 96, 250, 511, 480, 960, 640, 4096,
                                1200, 1001, 77, 1536, 3000, 160, 200, 320, 800, 400]))
        win = int(rng.choice([n_fft, n_fft, max(2, n_fft // 2), max(3, n_fft - 7)]))
        hop = int(rng.choice([max(1, win // 4), max(1, win // 2), max(1, win // 3 + 1), win]))
        fmt_in = str(rng.choice(["channels_last", "channels_first"]))
        fmt_out = str(rng.choice(["channels_last", "channels_first"]))
        out.append(dict(n_fft=n_fft, win=win, hop=hop, pad_begin=bool(rng.integers(2)), pad_end=bool(rng.integers(2)),
                        ch=int(rng.integers(1, 4)), batch=int(rng.integers(1, 4)), frames=int(rng.integers(1, 40)),
                        fmt_in=fmt_in, fmt_out=fmt_out, n_mels=int(rng.choice([13, 40, 64, 80, 128])),
                        db=bool(rng.integers(2)), window=str(rng.choice(["hann_window", "hamming_window", "vorbis_window"])),
                        seed=1000 + i))
    return out


# (KPR_RANDOM_CONFIGS="count,seed" widens the sweep for a one-off soak run)
_RC = [int(v) for v in os.environ.get("KPR_RANDOM_CONFIGS", "64,2024").split(",")]


@pytest.mark.parametrize("cfg", _random_configs(_RC[0], _RC[1]), ids=lambda c: "nfft%d_w%d_h%d_c%d_%s" % (
    c["n_fft"], c["win"], c["hop"], c["ch"], c["fmt_in"][9:] + c["fmt_out"][9:]))
def test_random_configurations_stft_mel_istft(cfg):
    """64 seeded random configurations across every dispatch path (Stockham / sub-FFT / mixed-radix /
    Bluestein / DFT-GEMM STFT, wave-specialised / ring / two-kernel mel, ring / barrier / two-kernel ISTFT): STFT, mel (+dB) and the
    inverse STFT of the computed spectrum against the float64 oracle."""
    n_fft, win, hop = cfg["n_fft"], cfg["win"], cfg["hop"]
    if cfg["pad_begin"] and n_fft < hop:
        pytest.skip("pad_begin needs n_fft >= hop")
    t = win + (cfg["frames"] - 1) * hop + 5
    shape = (cfg["batch"], t, cfg["ch"]) if cfg["fmt_in"] == "channels_last" else (cfg["batch"], cfg["ch"], t)
    x = synth(shape, cfg["seed"])
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, window_name=cfg["window"], pad_begin=cfg["pad_begin"],
              pad_end=cfg["pad_end"], input_data_format=cfg["fmt_in"], output_data_format=cfg["fmt_out"])
    want = o.kapre_stft(x, n_fft, win, hop, cfg["window"], cfg["pad_begin"], cfg["pad_end"], cfg["fmt_in"], cfg["fmt_out"])
    st = STFT(**kw)
    spec = st(x)
    assert_close(to_np(spec), want, rel=5e-5)
    mkw = dict(kw, sample_rate=16000, n_mels=cfg["n_mels"], return_decibel=cfg["db"])
    got = to_np(composed.get_melspectrogram_layer(**mkw)(x))
    ref = o.kapre_melspectrogram(x, **{k: v for k, v in mkw.items()})
    (assert_db_close if cfg["db"] else assert_close)(got, ref)
    if hop <= win:
        ikw = dict(n_fft=n_fft, win_length=win, hop_length=hop, forward_window_name=cfg["window"],
                   input_data_format=cfg["fmt_out"], output_data_format=cfg["fmt_in"])
        rec = to_np(InverseSTFT(**ikw)(spec))
        ref_rec = o.kapre_istft(want, n_fft, win, hop, cfg["window"], cfg["fmt_out"], cfg["fmt_in"])
        assert rec.shape == ref_rec.shape
        # the synthesis window w / sum_shifts(w^2) has a large gain wherever the shifted windows nearly
        # vanish (hop == win with a window that goes to zero at its ends): fp32 round-off of the
        # spectrum is amplified by exactly that gain, so the bound scales with it
        # (where every shifted window is exactly zero the window is 0/0 = NaN, as in TensorFlow: same
        # positions in both)
        sw = np.abs(o.inverse_stft_window(win, hop, o.get_window(cfg["window"], win)))
        gain = float(np.nanmax(sw)) if np.isfinite(sw).any() else 1.0
        fin = np.isfinite(ref_rec)
        assert np.array_equal(np.isfinite(rec), fin)
        assert np.abs(rec[fin] - ref_rec[fin]).max() <= 1e-4 * max(1.0, gain) * max(1.0, float(np.abs(ref_rec[fin]).max()))


# ------------------------------------------------------------------ boundary behaviour (round 2)
def test_decibel_more_than_2_pow_20_items():
    """k_stats_init must initialise every item's max / min slot, also beyond one capped grid
    (4096 blocks x 256 threads = 2^20 items)."""
    import torch

    n = (1 << 20) + 4097
    x = torch.rand((n, 3), device="cuda") * 4.0 + 1e-3
    x[-1] = torch.tensor([1e-9, 1.0, 100.0], device="cuda")
    got = backend.magnitude_to_decibel(x, dynamic_range=15.0)
    y = 10.0 * torch.log10(torch.clamp(x, min=1e-5))
    want = torch.maximum(y, y.max(dim=1, keepdim=True).values - 15.0)
    torch.testing.assert_close(got, want, rtol=0, atol=1e-4)
    assert got[-1].tolist() == pytest.approx([5.0, 5.0, 20.0], abs=1e-4)


def test_filterbank_reassignment_invalidates_fused_plans():
    """A model that was CALLED (fused-call plan cached, device / packed copies made) must use a filterbank
    assigned afterwards -- what dist.broadcast_constants does on the non-source ranks."""
    x = synth((3, 9000, 1), 77)
    kw = dict(n_fft=1024, hop_length=256, sample_rate=16000, n_mels=64)
    model = composed.get_melspectrogram_layer(**kw)
    before = to_np(model(x))
    assert_close(before, o.kapre_melspectrogram(x, **kw))
    fb_layer = model.layers[2]
    new_fb = (fb_layer.filterbank[:, ::-1] * np.float32(0.5)).copy()        # another banded matrix
    fb_layer.filterbank = new_fb
    after = to_np(model(x))
    want = o.apply_filterbank(np.abs(o.kapre_stft(x, 1024, None, 256)), new_fb, "channels_last")
    assert_close(after, want)
    assert not np.allclose(after, before)
    assert_close(to_np(fb_layer(Sequential([model.layers[0], Magnitude()])(x))), want)   # unfused path agrees


def test_packed_filterbank_of_another_matrix_is_refused():
    """kpr_mel_f32 / kpr_apply_filterbank_packed_f32 must not trust that fb_packed and fb_kranges_host describe the
    same matrix: the blob's header is verified on first use -> KPR_E_BADARG."""
    import ctypes
    import torch
    from kapre_amd import _ffi

    L = _ffi.lib()
    # the header check is cached per device address; torch's allocator hands freed addresses out again, so start from a
    # clean slate (what a caller that frees blobs does with kpr_filterbank_forget(ptr))
    assert L.kpr_filterbank_forget(None) == 0
    fb_a = np.asarray(backend.filterbank_mel(44100, 1025, 128), np.float32)
    fb_b = np.asarray(backend.filterbank_mel(22050, 1025, 128, 300.0, 8000.0), np.float32)
    kr_a, kr_b = _ffi.filterbank_kranges(fb_a), _ffi.filterbank_kranges(fb_b)
    assert not np.array_equal(kr_a, kr_b)
    packed_b = torch.from_numpy(_ffi.filterbank_pack(fb_b, kr_b)).cuda()
    fb_dev = torch.from_numpy(fb_a).cuda()
    x = torch.rand((2 * 5 * 1025,), device="cuda")
    out = torch.empty((2 * 5 * 128,), device="cuda")
    rc = L.kpr_apply_filterbank_packed_f32(_ffi.ptr(x), 2, 1, 5, 1025, 0, _ffi.ptr(fb_dev), _ffi.ptr(packed_b), 128,
                                           kr_a.ctypes.data_as(ctypes.c_void_p), _ffi.ptr(out),
                                           _ffi.current_stream_ptr())
    assert rc == -1 and b"another filterbank" in L.kpr_last_error()
    junk = torch.zeros(4096, device="cuda")
    rc = L.kpr_apply_filterbank_packed_f32(_ffi.ptr(x), 2, 1, 5, 1025, 0, _ffi.ptr(fb_dev), _ffi.ptr(junk), 128,
                                           kr_a.ctypes.data_as(ctypes.c_void_p), _ffi.ptr(out),
                                           _ffi.current_stream_ptr())
    assert rc == -1 and b"header" in L.kpr_last_error()
    # the matching blob is accepted and gives the product
    packed_a = torch.from_numpy(_ffi.filterbank_pack(fb_a, kr_a)).cuda()
    rc = L.kpr_apply_filterbank_packed_f32(_ffi.ptr(x), 2, 1, 5, 1025, 0, _ffi.ptr(fb_dev), _ffi.ptr(packed_a), 128,
                                           kr_a.ctypes.data_as(ctypes.c_void_p), _ffi.ptr(out),
                                           _ffi.current_stream_ptr())
    assert rc == 0
    assert_close(to_np(out).reshape(10, 128), to_np(x).reshape(10, 1025).astype(np.float64) @ fb_a.astype(np.float64))
    # the check is cached per device address (documented as best effort): the same buffer with a destroyed header is only
    # noticed after the caller has said that the address was released
    packed_a[:8] = 0.0
    torch.cuda.synchronize()
    call = lambda: L.kpr_apply_filterbank_packed_f32(_ffi.ptr(x), 2, 1, 5, 1025, 0, _ffi.ptr(fb_dev), _ffi.ptr(packed_a), 128,
                                                     kr_a.ctypes.data_as(ctypes.c_void_p), _ffi.ptr(out),
                                                     _ffi.current_stream_ptr())
    # (a) the MFMA kernels ("fb_variant" 1: rounds 2-5) read fragments whose layout the cached key fixes: the call simply runs
    prev = _ffi.set_option("fb_variant", 1)
    try:
        assert call() == 0
    finally:
        _ffi.set_option("fb_variant", prev)
    assert _ffi.device_status(raise_on_error=False) == 0
    # (b) k_fb_pw (round 6, the default for this bank) compares the header words with the plan it was launched for ON THE DEVICE,
    # as k_mel_pw does: the launch is enqueued (rc 0), computes nothing and raises the stale-plan bit -- the next call would fail
    # with KPR_E_DEVICE; reading the status clears it
    assert call() == 0
    assert _ffi.device_status(raise_on_error=False) == 1 << 4
    assert _ffi.device_status(raise_on_error=False) == 0
    assert L.kpr_filterbank_forget(_ffi.ptr(packed_a)) == 0
    assert call() == -1 and b"header" in L.kpr_last_error()


def test_more_than_1024_filters_fall_back_to_the_dense_product():
    """The packed schedule holds 64 tiles; the reference has no such limit (tensordot)."""
    import ctypes

    from kapre_amd import _ffi
    rng = np.random.default_rng(11)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=129, n_mels=8))
    layer.filterbank = rng.standard_normal((129, 1040)).astype(np.float32)
    x = rng.uniform(0, 1, (2, 6, 129, 1)).astype(np.float32)
    assert_close(to_np(layer(x)), o.apply_filterbank(x, layer.filterbank, "channels_last"), rel=2e-6)
    wav = synth((2, 3000, 1), 12)
    fused = to_np(Sequential([STFT(n_fft=256, hop_length=64), Magnitude(), layer])(wav))
    assert_close(fused, o.apply_filterbank(np.abs(o.kapre_stft(wav, 256, None, 64)), layer.filterbank, "channels_last"))
    # same at an n_fft that HAS a fused kernel: the workspace must then be sized for the two-kernel path
    layer2 = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=1025, n_mels=8))
    layer2.filterbank = (rng.standard_normal((1025, 1040)) * (rng.random((1025, 1040)) < 0.02)).astype(np.float32)
    wav2 = synth((2, 6000, 1), 13)
    fused2 = to_np(Sequential([STFT(n_fft=2048, hop_length=512), Magnitude(), layer2])(wav2))
    assert_close(fused2, o.apply_filterbank(np.abs(o.kapre_stft(wav2, 2048, None, 512)), layer2.filterbank, "channels_last"))
    g = _ffi.StftGeom(batch=2, channels=1, time=6000, n_fft=2048, win_length=2048, hop_length=512, pad_begin=0,
                      pad_end=0, in_layout=1, out_layout=1)
    L = _ffi.lib()
    assert L.kpr_mel_workspace_bytes(ctypes.byref(g), 1040, None) == L.kpr_mel_workspace_bytes_unpacked(ctypes.byref(g), 1040)
    assert L.kpr_mel_workspace_bytes(ctypes.byref(g), 128, None) < L.kpr_mel_workspace_bytes_unpacked(ctypes.byref(g), 128)


def test_malformed_kranges_are_reported_not_swallowed():
    """kpr_mel_f32 falls back to the dense product only for KPR_E_UNSUPPORTED schedules; bad k-ranges are an error."""
    import ctypes

    import torch

    from kapre_amd import _ffi
    L = _ffi.lib()
    g = _ffi.StftGeom(batch=1, channels=1, time=4096, n_fft=2048, win_length=2048, hop_length=512, pad_begin=0,
                      pad_end=0, in_layout=1, out_layout=1)
    fb = torch.zeros((1025, 32), device="cuda")
    x = torch.zeros(4096, device="cuda")
    win = torch.ones(2048, device="cuda")
    n_frames = int(L.kpr_num_frames(ctypes.byref(g)))
    out = torch.empty(n_frames * 32, device="cuda")
    ws = torch.empty(int(L.kpr_mel_workspace_bytes_unpacked(ctypes.byref(g), 32)), dtype=torch.uint8, device="cuda")
    bad = np.array([0, 1028, 6, 1028], dtype=np.int32)           # lo of the second tile is not a multiple of four
    rc = L.kpr_mel_f32(_ffi.ptr(x), ctypes.byref(g), _ffi.ptr(win), _ffi.ptr(fb), None, 32,
                       bad.ctypes.data_as(ctypes.c_void_p), None, _ffi.ptr(out), _ffi.ptr(ws), ws.numel(),
                       _ffi.current_stream_ptr())
    assert rc == -1 and b"k-range" in L.kpr_last_error()


# ------------------------------------------------------------------ k_mel_pw PAIR form: the staged channels_last store (round 5)
@pytest.mark.parametrize("n_fft,hop,batch,frames,ch,n_mels,db,fmt_out", [
    (2048, 512, 5, 23, 6, 128, True, "channels_last"),        # cfg3's layout: three pair-waves per (item, frame) block
    (2048, 1024, 3, 41, 4, 128, False, "channels_last"),
    (2048, 512, 2, 7, 8, 40, False, "channels_last"),         # fewer blocks than slots
    (1024, 256, 4, 37, 6, 80, True, "channels_last"),         # n_fft 1024 (two pairs per ticket): the 8-byte stores, not staged
    (1024, 160, 3, 50, 4, 64, False, "channels_last"),
    (2048, 512, 40, 30, 6, 128, True, "channels_last"),       # several workgroups, runs cut at whole blocks
    (2048, 512, 4, 19, 2, 128, False, "channels_last"),       # C = 2: no staging (the 8-byte pairs are contiguous)
    (2048, 512, 4, 19, 6, 128, False, "channels_first"),      # channels_first output: no staging
    (2048, 512, 3, 11, 6, 125, False, "channels_last"),       # M C not a multiple of 4 ... 125 * 6 = 750: the 8-byte stores
])
def test_mel_pw_pair_staged_channels_last_store(n_fft, hop, batch, frames, ch, n_mels, db, fmt_out):
    """The PAIR form collects the n_mels x C block of every (item, frame) in an LDS slot and stores it as one contiguous run
    (VERDICT r04 item 2).  Same values, bit for bit, as the 8-byte pair stores ("mel_cl_stage" 0), both against the oracle;
    repeated calls identical (which wave arrives last at a block varies, the block does not)."""
    import torch
    from kapre_amd import _ffi

    t = n_fft + (frames - 1) * hop
    x = synth((batch, t, ch), 7000 + frames + ch)
    x *= np.logspace(-1, 0, batch, dtype=np.float32).reshape(batch, 1, 1)
    kw = dict(n_fft=n_fft, hop_length=hop, sample_rate=44100, n_mels=n_mels, return_decibel=db,
              input_data_format="channels_last", output_data_format=fmt_out)
    layer = composed.get_melspectrogram_layer(**kw)
    old = _ffi.set_option("mel_variant", 8)                     # the PAIR form wherever it applies (small launches too)
    try:
        got = layer(x)
        assert "k_mel_pw_pair" in _ffi.last_launches()
        for _ in range(3):
            assert torch.equal(layer(x), got)
        prev = _ffi.set_option("mel_cl_stage", 0)
        try:
            plain = layer(x)
        finally:
            _ffi.set_option("mel_cl_stage", prev)
        assert torch.equal(plain, got)
    finally:
        _ffi.set_option("mel_variant", old)
    want = o.kapre_melspectrogram(x, **kw)
    (assert_db_close if db else assert_close)(to_np(got), want)
    assert _ffi.device_status() == 0
