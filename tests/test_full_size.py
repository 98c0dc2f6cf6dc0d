"""-m gpu: parity against the float64 oracle at the FULL size of every BASELINE.json configuration
(SURVEY.md section 8 table): cfg3 256 x 6ch in both layouts, cfg4 128 items forward / inverse / round trip,
cfg5 one GPU's share (256 items), and every item of the north-star batch.  The oracle walks the batch in
chunks (the batch axis is independent: the dB maximum is per item), each chunk compared as soon as it exists,
so host memory stays bounded.  Tolerances: north_star's 1e-4 relative to the output scale; decibel outputs to
1e-3 dB absolute."""
import numpy as np
import pytest

import kapre_oracle as o

from kapre_amd import composed

pytestmark = pytest.mark.gpu

REL = 1e-4            # north_star's contract
REG = 4e-6            # regression bound next to it (measured: 1e-7 ... 8e-7 of the item's scale)
DB_ABS = 1e-3


def synth(shape, seed):
    return np.random.default_rng(seed).uniform(-1, 1, shape).astype(np.float32)


def chunked_check(got, x, oracle_fn, chunk, db=False, scale=None):
    """Compare got[i:i+chunk] with oracle_fn(x[i:i+chunk]) for the whole batch, EVERY ITEM AGAINST ITS OWN SCALE (the
    largest |value| the oracle has for that item; `scale` overrides it with a fixed number, e.g. 1.0 for waveforms):
    a quiet item is held to 1e-4 of its own level, not of its loud neighbours'.  Returns the worst relative error."""
    worst = 0.0
    for i in range(0, x.shape[0], chunk):
        want = oracle_fn(x[i:i + chunk])
        g = got[i:i + chunk]
        assert g.shape == want.shape, (g.shape, want.shape)
        assert np.isfinite(g).all()
        n = g.shape[0]
        err = np.abs(g - want).reshape(n, -1).max(axis=1)
        if db:
            assert float(err.max()) <= DB_ABS, "items %d..: dB error %.3g" % (i, float(err.max()))
            worst = max(worst, float(err.max()))
            continue
        s = np.full(n, scale, np.float64) if scale is not None else np.abs(want).reshape(n, -1).max(axis=1)
        s = np.maximum(s, np.finfo(np.float32).tiny)
        bad = np.nonzero(err > REL * s)[0]
        assert bad.size == 0, "item %d: relative error %.3g (own scale %.3g)" % (i + bad[0], err[bad[0]] / s[bad[0]], s[bad[0]])
        worst = max(worst, float((err / s).max()))
    assert db or worst <= REG, "worst relative error %.3g: inside the contract, beyond the regression bound %.0e" % (worst, REG)
    return worst


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_cfg3_full_logmel_db_256x6(fmt):
    """configs[2]: LogMel + dB, batch=256, 6ch, 44100 @44.1 kHz, n_fft=2048 hop=1024 n_mels=128."""
    shape = (256, 44100, 6) if fmt == "channels_last" else (256, 6, 44100)
    x = synth(shape, 1236)
    # per-item gains over 60 dB; every fifth item is loud (+90 dB) with a silent tail, so that its amin floor
    # (-50 dB) lies more than 80 dB below its maximum: the per-item dynamic-range clamp must engage there
    x *= np.logspace(-3, 0, 256, dtype=np.float32).reshape(256, 1, 1)
    t_axis = 1 if fmt == "channels_last" else 2
    sl = [slice(None)] * 3
    sl[0] = slice(0, 256, 5)
    x[tuple(sl)] *= np.float32(3e4)
    sl[t_axis] = slice(30000, None)
    x[tuple(sl)] = 0
    kw = dict(n_fft=2048, hop_length=1024, sample_rate=44100, n_mels=128, return_decibel=True,
              input_data_format=fmt, output_data_format=fmt)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    assert got.shape == ((256, 42, 128, 6) if fmt == "channels_last" else (256, 6, 42, 128))
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kw), 16, db=True)
    flat = got.reshape(256, -1)
    assert float((flat.max(axis=1) - flat.min(axis=1)).max()) <= 80.0 + 1e-3
    spread = flat.max(axis=1) - flat.min(axis=1)
    on_clamp = np.abs(spread - 80.0) <= 1e-3
    assert on_clamp[::5].sum() >= 20 and not on_clamp[1::5].any()           # loud items with silent tails sit on the clamp


def test_cfg4_full_stft_istft_roundtrip_128():
    """configs[3]: batch=128, 1ch, 5 s @22.05 kHz, n_fft=1024 hop=256, pad_begin + pad_end."""
    x = synth((128, 110250, 1), 1237)
    stft, istft = composed.get_perfectly_reconstructing_stft_istft(1024, 256, "channels_last", "channels_last")
    s = stft(x)
    assert tuple(s.shape) == (128, 434, 513, 1)
    s_np = s.cpu().numpy()
    chunked_check(s_np, x, lambda xc: o.kapre_stft(xc, 1024, 1024, 256, "hann_window", True, True), 8)
    rec = istft(s).cpu().numpy()
    assert rec.shape == (128, 433 * 256 + 1024, 1)
    chunked_check(rec, s_np, lambda sc: o.kapre_istft(sc, 1024, 1024, 256, "hann_window"), 8, scale=1.0)
    np.testing.assert_allclose(rec[:, 768:768 + 110250], x, atol=1e-5)       # upstream's round-trip tolerance


def test_cfg5_full_share_mel_256():
    """configs[4], one GPU's share: 256 of the 2048 items, 10 s @16 kHz, n_fft=1024 hop=160 n_mels=80."""
    x = synth((256, 160000, 1), 1238)
    kw = dict(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=80)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    assert got.shape == (256, 994, 80, 1)
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kw), 16)


def test_cfg5_full_2048_items_one_launch():
    """configs[4] at its FULL batch in ONE launch (the shape bench.py times as cfg5_..._strong, 2048 x 10 s @16 kHz: 2 M frames,
    8 x more tickets than workgroup slots -- the region where round 4's stash-reuse bug lived).  The oracle
    would need 16 GB for it, so: items 0 .. 63 are random and checked against the oracle; item i >= 64 is item (i mod 64) scaled
    by a power of two, and scaling by a power of two is EXACT through window, FFT, magnitude and the mel sums -- every one of the
    2048 outputs must be bit-equal to the scaled output of its base item, wherever in the launch it was computed."""
    import torch
    base = synth((64, 160000, 1), 1242)
    e = (np.arange(2048) // 64) % 13 - 6                                     # 2^-6 ... 2^6
    gain = np.ldexp(np.float32(1.0), e).astype(np.float32)
    x = torch.from_numpy(base).cuda().repeat(32, 1, 1) * torch.from_numpy(gain).cuda().reshape(2048, 1, 1)
    kw = dict(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=80)
    got = composed.get_melspectrogram_layer(**kw)(x)
    assert tuple(got.shape) == (2048, 994, 80, 1)
    from kapre_amd import _ffi
    assert "k_mel_pw<512,w16>" in _ffi.last_launches()
    chunked_check(got[:64].cpu().numpy() * 64.0, base, lambda xc: o.kapre_melspectrogram(xc, **kw), 16)   # (gain of items 0 .. 63: 2^-6)
    want = got[:64].repeat(32, 1, 1, 1) * torch.from_numpy(gain * np.float32(64.0)).cuda().reshape(2048, 1, 1, 1)
    same = torch.equal(got, want)
    if not same:
        bad = torch.nonzero((got != want).reshape(2048, -1).any(dim=1)).flatten().cpu().numpy()
        raise AssertionError("items not bit-equal to their scaled base item: %s ..." % bad[:16])


def test_target_full_mel_256_every_item():
    """north_star target: batch=256 x 1ch x 44100, n_fft=2048 hop=512 n_mels=128 -- all 256 items, plain and dB."""
    x = synth((256, 44100, 1), 1239)
    x[7] *= np.float32(1e-3)
    x[100, 20000:] = 0
    kw = dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kw), 32)      # item 7 (x 1e-3) against its own scale
    kwd = dict(kw, return_decibel=True)
    got = composed.get_melspectrogram_layer(**kwd)(x).cpu().numpy()
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kwd), 32, db=True)


def test_speech_full_mel_256_nfft400():
    """The speech front-end bench.py carries in `also`: 256 x 10 s @16 kHz, n_fft=400 (25 ms) hop=160 (10 ms), 80 mels
    (mixed-radix STFT + MFMA filterbank consumers: two launches)."""
    x = synth((256, 160000, 1), 1240)
    x[3] *= np.float32(1e-3)
    kw = dict(n_fft=400, hop_length=160, sample_rate=16000, n_mels=80)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    assert got.shape == (256, 998, 80, 1)
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kw), 16)


@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
def test_reference_test_shape_full_logmel_256x2(fmt):
    """The shape of the reference's own mel tests (/root/reference/tests/test_time_frequency.py:188-267: n_fft 512,
    sr 22050, 2 channels, 40 mels up to 8 kHz, decibel) at the batch bench.py times it with."""
    shape = (256, 22050, 2) if fmt == "channels_last" else (256, 2, 22050)
    x = synth(shape, 1241)
    x *= np.logspace(-2, 0, 256, dtype=np.float32).reshape(256, 1, 1)
    kw = dict(n_fft=512, hop_length=128, sample_rate=22050, n_mels=40, mel_f_max=8000.0, return_decibel=True,
              input_data_format=fmt, output_data_format=fmt)
    got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    assert got.shape == ((256, 169, 40, 2) if fmt == "channels_last" else (256, 2, 169, 40))
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kw), 32, db=True)
    kwl = dict(kw, return_decibel=False)
    got = composed.get_melspectrogram_layer(**kwl)(x).cpu().numpy()
    chunked_check(got, x, lambda xc: o.kapre_melspectrogram(xc, **kwl), 32)
