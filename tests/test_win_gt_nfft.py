"""Forward transforms with win_length > n_fft ("missing" 3).

kapre/time_frequency.py:174-182 hands frame_length = win_length and fft_length = n_fft straight to tf.signal.stft: frames of
win_length samples are cut (the frame COUNT follows win_length) and windowed, then rfft(fft_length) keeps their first n_fft
samples.  Through round 5 float32 sent these calls to the DFT-as-GEMM fallback (untested on the GPU) and float64 refused them;
since round 6 every FFT family takes them (kapre_hip.hip: forward_geom).  Oracle: oracle/kapre_oracle.py tf_stft (crop), numpy's
rfft(n=...) for float64."""
import numpy as np
import pytest

import kapre_oracle as o

pytestmark = pytest.mark.gpu

CL, CF = "channels_last", "channels_first"


def _err(got, want):
    return float(np.abs(np.asarray(got) - want).max() / max(np.abs(want).max(), 1e-30))


# (n_fft, win_length, hop, family label that must appear in kpr_last_launches)
CASES = [(1024, 1536, 256, "k_stft<512"), (2048, 2049, 512, "k_stft<1024"), (512, 1000, 128, "k_stft<256"), (256, 300, 64, "k_stft<128"),
         (400, 500, 160, "k_stft_mr"), (1000, 1024, 250, "k_stft_mr"), (480, 512, 120, "k_stft_mr"), (300, 301, 75, "k_stft_bs"),
         (4096, 5000, 1024, "k_stft_big"), (1200, 1201, 300, "k_stft_gen<float>"), (250, 400, 60, "k_stft_bs"), (15, 22, 4, "k_stft_gen<float>")]


@pytest.mark.parametrize("n_fft, win, hop, label", CASES)
@pytest.mark.parametrize("fmt, pads", [(CL, (False, False)), (CF, (True, True)), (CL, (False, True))])
def test_stft_crops_frames_longer_than_n_fft(n_fft, win, hop, label, fmt, pads):
    from kapre_amd import STFT, Magnitude, Phase, Sequential, _ffi
    rng = np.random.default_rng(n_fft + win)
    b, c, t = 3, 2, 4 * win + 7 * hop + 13
    x = rng.standard_normal((b, t, c) if fmt == CL else (b, c, t)).astype(np.float32)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=pads[0], pad_end=pads[1], input_data_format=fmt,
              output_data_format=fmt)
    want = o.kapre_stft(x, **kw)
    assert want.shape[1 if fmt == CL else 2] == o.num_frames(t + (n_fft - hop if pads[0] else 0), win, hop, pads[1])   # frames follow win_length
    got = STFT(**kw)(x).cpu().numpy()
    assert label in _ffi.last_launches(), _ffi.last_launches()
    assert got.shape == want.shape
    e = _err(got, want)
    assert e <= 1e-4 and e <= 4e-6, e
    mag = Sequential([STFT(**kw), Magnitude()])(x).cpu().numpy()
    assert _err(mag, np.abs(want)) <= 4e-6
    ph = Sequential([STFT(**kw), Phase()])(x).cpu().numpy()
    strong = np.abs(want) > 1e-2 * np.abs(want).max()
    assert float(np.abs(np.angle(np.exp(1j * (ph - np.angle(want))))[strong]).max()) < 2e-3


@pytest.mark.parametrize("n_fft, win, hop, n_mels, sr, label", [(2048, 2500, 512, 128, 44100, "k_mel_pw<1024"), (1024, 1100, 160, 80, 16000, "k_mel_pw<512"),
                                                                 (512, 700, 128, 40, 22050, "k_mel_pw<256"), (400, 480, 160, 80, 16000, "k_mel_mr<200>"),
                                                                 (300, 320, 100, 32, 16000, "")])
@pytest.mark.parametrize("db", [False, True])
def test_fused_mel_with_frames_longer_than_n_fft(n_fft, win, hop, n_mels, sr, label, db):
    """get_melspectrogram_layer(win_length > n_fft) stays on the single-launch kernels (through round 5: the two-launch fallback)"""
    from kapre_amd import composed, _ffi
    rng = np.random.default_rng(n_fft)
    x = rng.uniform(-1, 1, (5, 30 * hop + win, 2)).astype(np.float32)
    kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, sample_rate=sr, n_mels=n_mels, return_decibel=db, pad_end=True)
    want = o.kapre_melspectrogram(x, **kw)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    assert label in _ffi.last_launches(), _ffi.last_launches()
    if db:
        got, want = 10.0 ** (got.astype(np.float64) / 10.0), 10.0 ** (want / 10.0)
    assert _err(got, want) <= 4e-6


@pytest.mark.parametrize("n_fft, win, hop", [(512, 700, 160), (1000, 1300, 250), (15, 22, 4), (2048, 2049, 512)])
@pytest.mark.parametrize("fmt", [CL, CF])
def test_stft_float64_crops_frames_longer_than_n_fft(n_fft, win, hop, fmt):
    """kpr_stft_f64 used to return KPR_E_UNSUPPORTED for win_length > n_fft (kapre_hip.hip:2550 in round 5)"""
    from kapre_amd import STFT, backend
    rng = np.random.default_rng(win)
    b, c, t = 2, 2, 3 * win + 5 * hop + 1
    x = rng.standard_normal((b, c, t))
    window = backend.window_values(backend.get_window_fn(None), win, np.float64)
    idx = np.arange(win)[None, :] + hop * np.arange(1 + (t - win) // hop)[:, None]
    want = np.fft.rfft(x[..., idx] * window, n=n_fft, axis=-1)             # numpy crops to the first n_fft samples as TF does
    xin = x if fmt == CF else np.ascontiguousarray(x.transpose(0, 2, 1))
    got = STFT(n_fft=n_fft, win_length=win, hop_length=hop, input_data_format=fmt, output_data_format=fmt, dtype="float64")(xin)
    assert str(got.dtype) == "torch.complex128"
    got = got.cpu().numpy()
    if fmt == CL:
        got = got.transpose(0, 3, 1, 2)
    assert got.shape == want.shape and _err(got, want) <= 1e-11


def test_gradient_of_a_cropped_stft():
    """STFT^T with win_length > n_fft = InverseSTFT with zero-extended frames (the adjoint of the crop): checked against central
    differences of the forward pass in float64"""
    import torch
    from kapre_amd import STFT
    rng = np.random.default_rng(2)
    layer = STFT(n_fft=256, win_length=320, hop_length=64, dtype="float64")
    x = torch.from_numpy(rng.standard_normal((2, 2000, 1))).cuda()
    r = torch.from_numpy(rng.standard_normal(tuple(layer(x).shape)) + 1j * rng.standard_normal(tuple(layer(x).shape))).cuda()
    d = torch.from_numpy(rng.standard_normal(tuple(x.shape))).cuda()
    xg = x.clone().requires_grad_(True)
    (layer(xg) * r.conj()).real.sum().backward()
    lhs = float((xg.grad * d).sum())
    eps = 1e-6
    rhs = float(((layer(x + eps * d) - layer(x - eps * d)) * r.conj()).real.sum()) / (2 * eps)
    assert abs(lhs - rhs) <= 1e-6 * max(abs(rhs), 1e-30)


def test_wide_banks_without_a_band_plan_run(tmp_path):
    """banks of more than 256 filters without a band plan at n_fft 512 / 2048 -- beyond k_mel_ts, beyond
    k_mel_ws at n_fft 512 -- take the two-launch path, whose workspace kpr_mel_workspace_bytes now covers; a dense bank of fewer
    filters that no fused schedule holds makes the Python layer retry with kpr_mel_workspace_bytes_unpacked."""
    from kapre_amd import STFT, Magnitude, ApplyFilterbank, Sequential, _ffi
    rng = np.random.default_rng(4)
    for n_fft, hop, n_filt, dense in ((512, 128, 300, False), (2048, 512, 700, False), (512, 128, 200, True), (1024, 256, 256, True)):
        k = n_fft // 2 + 1
        if dense:
            fb = rng.uniform(0.0, 1.0, (k, n_filt)).astype(np.float32)
        else:                                                                # log-frequency-like bumps, three non-zeros per bin
            fb = np.zeros((k, n_filt), np.float32)
            for kk in range(k):
                m = min(n_filt - 3, kk * n_filt // k)
                fb[kk, m:m + 3] = rng.uniform(0.1, 1.0, 3)
        x = rng.uniform(-1, 1, (3, 20 * hop + n_fft, 1)).astype(np.float32)
        layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=k, n_mels=8))
        layer.filterbank = fb
        model = Sequential([STFT(n_fft=n_fft, hop_length=hop), Magnitude(), layer])
        got = model(x).cpu().numpy()
        want = o.apply_filterbank(np.abs(o.kapre_stft(x, n_fft=n_fft, hop_length=hop)), fb.astype(np.float64), CL)
        assert got.shape == want.shape and _err(got, want) <= 4e-6, (n_fft, n_filt, dense, _ffi.last_launches())
