"""Non-finite magnitudes through the filterbank kernels ("missing" 2).

The reference's product is DENSE (kapre/time_frequency.py:544: tf.tensordot over the frequency axis): one NaN / Inf bin in a row
makes EVERY filter of that row NaN (0 * Inf) or +-Inf.  Kernels that skip exact zeros of the matrix behave differently, and this
file pins what each one returns:

* k_fb_pw (stand-alone ApplyFilterbank, mel / triangular banks on contiguous rows -- the default since round 6): EXACTLY the dense
  result's pattern of NaN / +Inf / -Inf (a row whose bins do not sum to a finite number is recomputed as the dense dot product);
* k_thin_gemm (narrow matrices, LogmelToMFCC) and k_gemm without k-ranges: dense, the same pattern;
* k_mel_ws<1024, FROM_MAG> (three or more interleaved channels, log-frequency banks, "fb_variant" 1), k_band_mel (n_freq > 1025), k_gemm with
  k-ranges: the 16-filter TILES whose row range contains the bin are poisoned, filters of other tiles stay finite;
* the fused chain (k_mel_pw): a NaN SAMPLE makes every bin of its frames NaN -- the Nyquist bin too, which enters every filter's
  sum -- hence every filter of those frames: the dense product's result.
All of them: rows without a non-finite bin are untouched.  INTEGRATION.md section "Non-finite values" says the same in prose.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CL, CF = "channels_last", "channels_first"


def _classes(a):
    return np.where(np.isnan(a), 3, np.where(a == np.inf, 1, np.where(a == -np.inf, 2, 0)))


def _poisoned_input(k, rows=40, seed=0):
    rng = np.random.default_rng(seed)
    x = np.abs(rng.standard_normal((2, 1, rows, k))).astype(np.float32)
    x[0, 0, 3, k // 3] = np.nan
    x[0, 0, 5, k // 2] = np.inf
    x[1, 0, 7, 0] = np.inf
    x[1, 0, 9, k - 1] = np.nan
    x[1, 0, 11, 5] = np.inf
    x[1, 0, 11, k - 7] = np.nan
    return x, [(0, 3), (0, 5), (1, 7), (1, 9), (1, 11)]


def _dense(x, fb):
    with np.errstate(all="ignore"):
        return x.astype(np.float64) @ np.asarray(fb, np.float64)


@pytest.mark.parametrize("k, n_mels", [(1025, 128), (513, 80), (257, 40), (129, 20)])
@pytest.mark.parametrize("stereo", [False, True])
def test_fb_pw_returns_the_dense_products_pattern(k, n_mels, stereo):
    from kapre_amd import ApplyFilterbank, _ffi
    x, bad = _poisoned_input(k)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=k, n_mels=n_mels), data_format=CL if stereo else CF)
    if stereo:                       # two interleaved channels (the ST instance): channel 0 = the poisoned rows, channel 1 = clean ones
        clean = np.abs(np.random.default_rng(9).standard_normal(x.shape)).astype(np.float32)
        y2 = layer(np.ascontiguousarray(np.concatenate([x, clean], axis=1).transpose(0, 2, 3, 1))).cpu().numpy().transpose(0, 3, 1, 2)
        assert "k_fb_pw<%d,st>" % (k - 1) in _ffi.last_launches()
        assert np.isfinite(y2[:, 1]).all()                               # a channel's poison stays in its channel
        got = y2[:, :1]
    else:
        got = layer(x).cpu().numpy()
        assert "k_fb_pw<%d>" % (k - 1) in _ffi.last_launches()
    want = _dense(x, layer.filterbank)
    assert np.array_equal(_classes(got), _classes(want))
    for b, r in bad:
        assert not np.isfinite(got[b, 0, r]).any()                       # no finite value survives in such a row
    fin = np.isfinite(want)
    assert np.abs(got[fin] - want[fin]).max() <= 4e-6 * np.abs(want[fin]).max()


def test_thin_and_dense_gemm_are_dense():
    from kapre_amd import ApplyFilterbank, _ffi
    x, bad = _poisoned_input(256)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=256, n_mels=40), data_format=CF)
    got = layer(x).cpu().numpy()
    assert "k_thin_gemm" in _ffi.last_launches()
    assert np.array_equal(_classes(got), _classes(_dense(x, layer.filterbank)))
    rng = np.random.default_rng(1)
    x, bad = _poisoned_input(300)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=300, n_mels=24), data_format=CF)
    layer.filterbank = rng.uniform(0.1, 1, (300, 24)).astype(np.float32)       # no zeros: nothing to skip
    got = layer(x).cpu().numpy()
    assert np.array_equal(_classes(got), _classes(_dense(x, layer.filterbank))), _ffi.last_launches()


@pytest.mark.parametrize("case", ["mel_ws_cl", "mel_ws_forced", "log_bank", "band_mel"])
def test_zero_skipping_kernels_poison_whole_tiles_only(case):
    """what the kernels that skip exact zeros return: NaN / Inf in every filter of the 16-filter tiles whose row range [lo, hi)
    contains a non-finite bin (a superset of the <= 2 filters that overlap it, a subset of the dense result); nothing else changes"""
    from kapre_amd import ApplyFilterbank, _ffi
    k = 1300 if case == "band_mel" else 1025
    x, bad = _poisoned_input(k)
    kw = dict(sample_rate=22050, n_freq=k) if case == "log_bank" else dict(sample_rate=22050, n_freq=k, n_mels=64 if case == "band_mel" else 128)
    fmt = CL if case == "mel_ws_cl" else CF
    layer = ApplyFilterbank(type="log" if case == "log_bank" else "mel", filterbank_kwargs=kw, data_format=fmt)
    xin = x if fmt == CF else np.ascontiguousarray(np.concatenate([x, x, x], axis=1).transpose(0, 2, 3, 1))   # three channels, interleaved
    prev = _ffi.set_option("fb_variant", 1 if case == "mel_ws_forced" else 0)
    try:
        got = layer(xin).cpu().numpy()
        label = _ffi.last_launches()
    finally:
        _ffi.set_option("fb_variant", prev)
    assert ("k_band_mel" if case == "band_mel" else "k_mel_ws<1024>") in label, label
    if fmt == CL:
        got = got.transpose(0, 3, 1, 2)[:, :1]
    fb = np.asarray(layer.filterbank, np.float32)
    dense = _dense(x, fb)
    kr = _ffi.filterbank_kranges(fb).reshape(-1, 2)
    ntiles = kr.shape[0]
    some_tile_survives = False
    for b in range(2):
        for r in range(x.shape[2]):
            nf = np.flatnonzero(~np.isfinite(x[b, 0, r]))
            row = got[b, 0, r]
            if len(nf) == 0:
                assert np.isfinite(row).all(), (case, b, r)
                continue
            for t in range(ntiles):
                tile = row[16 * t:16 * t + 16]
                # whole tiles: all of a tile's filters are finite, or none is
                assert np.isfinite(tile).all() or not np.isfinite(tile).any(), (case, b, r, t)
                # a tile whose row range holds the bin is poisoned (the kernels round the ranges outwards to whole chunks of rows,
                # so a neighbouring tile may be poisoned as well)
                if any(kr[t, 0] <= kk < kr[t, 1] for kk in nf):
                    assert not np.isfinite(tile).any(), (case, b, r, t)
                else:
                    some_tile_survives = some_tile_survives or bool(np.isfinite(tile).all())
            if len(nf) == 1:        # the filters the bin really feeds carry the dense product's class (NaN, +Inf)
                feeds = fb[nf[0]] != 0
                assert np.array_equal(_classes(row)[feeds], _classes(dense[b, 0, r])[feeds]), (case, b, r)
    assert some_tile_survives           # ... which is where these kernels differ from the dense product (every filter non-finite)


def test_fused_mel_chain_with_a_nan_sample():
    """get_melspectrogram_layer (k_mel_pw): a NaN sample makes every bin of the frames that contain it NaN -- the Nyquist bin
    included, which enters EVERY filter's sum as fb[Nyquist][m] * |X[Nyquist]| (0 * NaN for all but the last filters) -- so every
    filter of those frames is NaN, empty filters (all-zero columns: 44.1 kHz, n_fft 512, 128 mels has eleven) included: the dense
    product's result.  Frames that do not contain the sample are untouched."""
    from kapre_amd import composed, _ffi
    rng = np.random.default_rng(5)
    for n_fft, hop, sr, n_mels in ((2048, 512, 44100, 128), (512, 128, 44100, 128)):
        x = rng.uniform(-1, 1, (3, 20 * hop + n_fft, 1)).astype(np.float32)
        t_bad = 7 * hop + 11
        x[1, t_bad, 0] = np.nan
        layer = composed.get_melspectrogram_layer(n_fft=n_fft, hop_length=hop, sample_rate=sr, n_mels=n_mels)
        got = layer(x).cpu().numpy()[..., 0]
        assert "k_mel_pw" in _ffi.last_launches()
        fb = np.asarray(layer.layers[2].filterbank, np.float32)
        empty = ~(fb != 0).any(axis=0)
        assert empty.any() == (n_fft == 512)
        frames = np.arange(got.shape[1])
        touched = (frames * hop <= t_bad) & (t_bad < frames * hop + n_fft)
        assert touched.sum() == n_fft // hop
        assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all() and np.isfinite(got[1][~touched]).all()
        assert np.isnan(got[1][touched]).all()
