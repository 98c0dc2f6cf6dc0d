"""k_fb_pw -- the stand-alone ApplyFilterbank (kapre/time_frequency.py:535-548: tf.tensordot + transpose) for banks with a band
plan, as banded row sums (kapre_amd/csrc/kpr_fb_pw_kernels.h, round 6).

* parity against the float64 oracle for every instance (n_freq 129 ... 1025) x layout x C in {1, 2, 3, 6}, from one row to launches
  that fill the chip, on both sides of the dispatch (band plan + contiguous rows -> k_fb_pw, two interleaved channels -> its ST
  instances; more interleaved channels, "fb_variant" 1 or a bank without a plan -> the MFMA kernels), each asserting the kernel
  that ran;
* bit-identical to pw_band_core -- the function it shares with the fused kernel k_mel_pw -- executed on the CPU in the kernel's
  order of operations;
* a row with a NaN / Inf bin returns what the reference's DENSE product returns (every filter NaN or +-Inf), see test_nonfinite.py.
"""
import numpy as np
import pytest

import kapre_oracle as o

pytestmark = pytest.mark.gpu

CL, CF = "channels_last", "channels_first"


def _layer(k, n_mels, fmt, sr=22050, **kw):
    from kapre_amd import ApplyFilterbank
    return ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=k, n_mels=n_mels, **kw), data_format=fmt)


def _item_err(got, want):
    b = want.shape[0]
    d = np.abs(np.asarray(got, np.float64) - want).reshape(b, -1).max(axis=1)
    sc = np.abs(want).reshape(b, -1).max(axis=1)
    return float((d / np.maximum(sc, 1e-30)).max())


@pytest.mark.parametrize("k, n_mels", [(129, 20), (257, 40), (513, 80), (1025, 128), (1025, 130), (513, 13)])
@pytest.mark.parametrize("fmt", [CF, CL])
@pytest.mark.parametrize("ch", [1, 2, 3, 6])
@pytest.mark.parametrize("rows, batch", [(1, 1), (7, 3), (83, 9)])
def test_fb_pw_matches_oracle(k, n_mels, fmt, ch, rows, batch):
    """both sides of the dispatch: contiguous rows (channels_first, or one channel) take k_fb_pw, two interleaved channels its ST
    instances, more interleaved channels the MFMA kernels"""
    from kapre_amd import _ffi
    rng = np.random.default_rng(k + 7 * n_mels + ch + rows)
    shape = (batch, rows, k, ch) if fmt == CL else (batch, ch, rows, k)
    x = (np.abs(rng.standard_normal(shape)) ** 3).astype(np.float32)
    x *= np.logspace(-3, 0, batch, dtype=np.float32).reshape((batch, 1, 1, 1))          # items of very different scale
    layer = _layer(k, n_mels, fmt)
    got = layer(x).cpu().numpy()
    label = _ffi.last_launches()
    want = o.apply_filterbank(x, o.filterbank_mel(22050, k, n_mels), fmt)
    assert got.shape == want.shape
    e = _item_err(got, want)
    assert e <= 1e-4 and e <= 4e-6, (e, label)                                           # contract / regression bound (measured ~2e-7)
    if fmt == CL and ch == 2:
        assert "k_fb_pw<%d,st>" % (k - 1) in label, label
    else:
        assert ("k_fb_pw<%d>" % (k - 1) in label) == (fmt == CF or ch == 1), label


@pytest.mark.parametrize("k, n_mels, rows, batch, ch", [(201, 80, 998, 8, 1), (201, 40, 1, 1, 2), (161, 64, 50, 3, 3), (81, 40, 200, 4, 1),
                                                       (1001, 128, 17, 5, 2), (481, 80, 333, 6, 1), (101, 24, 7, 2, 1), (5, 3, 9, 2, 1)])
def test_fb_pw_rows_of_any_multiple_of_four_bins(k, n_mels, rows, batch, ch):
    """n_freq - 1 a multiple of four (n_fft 400, 320, 160, 2000, 960, 200 ...): a plan laid out for the next 16 L bins; the quads a
    row does not have are never loaded (the last rows of the tensor end at the buffer's end) and count as zeros"""
    from kapre_amd import _ffi
    rng = np.random.default_rng(k + rows)
    x = np.abs(rng.standard_normal((batch, ch, rows, k), dtype=np.float32))
    got = _layer(k, n_mels, CF, sr=16000)(x).cpu().numpy()
    label = _ffi.last_launches()
    assert "k_fb_pw<" in label, label
    assert _item_err(got, o.apply_filterbank(x, o.filterbank_mel(16000, k, n_mels), CF)) <= 4e-6, label
    x[0, 0, rows // 2, k // 2] = np.inf                                  # ... and the dense recomputation reads rows of K floats
    x[batch - 1, ch - 1, rows - 1, k - 1] = np.nan                       # (the tensor's last element)
    got = _layer(k, n_mels, CF, sr=16000)(x).cpu().numpy()
    with np.errstate(all="ignore"):
        want = x.astype(np.float64) @ o.filterbank_mel(16000, k, n_mels).astype(np.float64)
    cls = lambda a: np.where(np.isnan(a), 3, np.where(a == np.inf, 1, np.where(a == -np.inf, 2, 0)))
    assert np.array_equal(cls(got), cls(want))


@pytest.mark.parametrize("k, n_mels, rows, batch, ch, fmt", [
    (1025, 128, 83, 256, 1, CL),        # the north-star shape (bench row k2_filterbank): 21 248 rows, 83 per CU
    (1025, 128, 83, 40, 6, CF),         # six channels
    (513, 80, 994, 24, 1, CF),          # two rows per wave
    (257, 40, 173, 64, 2, CF),
    (129, 20, 3000, 16, 3, CF),         # eight rows per wave
    (1025, 128, 83, 128, 2, CL),        # two interleaved channels: the ST instance, one (item, frame) block per wave
    (513, 80, 994, 12, 2, CL),
    (201, 80, 998, 16, 2, CL),          # ... on a padded plan
])
def test_fb_pw_large_launches_both_sides_of_the_dispatch(k, n_mels, rows, batch, ch, fmt):
    """launches that fill the chip; the MFMA kernels ("fb_variant" 1: what ran before round 6) within the same tolerance"""
    from kapre_amd import _ffi
    rng = np.random.default_rng(k + rows)
    shape = (batch, rows, k, ch) if fmt == CL else (batch, ch, rows, k)
    x = np.abs(rng.standard_normal(shape, dtype=np.float32))
    layer = _layer(k, n_mels, fmt, sr=44100)
    want = o.apply_filterbank(x, o.filterbank_mel(44100, k, n_mels), fmt)
    for variant in (0, 1):
        prev = _ffi.set_option("fb_variant", variant)
        try:
            got = layer(x).cpu().numpy()
            label = _ffi.last_launches()
        finally:
            _ffi.set_option("fb_variant", prev)
        e = _item_err(got, want)
        assert e <= 4e-6, (variant, e, label)
        assert ("k_fb_pw<" in label) == (variant == 0), (variant, label)


def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 values is exact in float64; the float64 sum is then rounded once more
    to float32 (a double rounding that differs from a true fma only when the float64 sum lands exactly on a float32 tie)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _run_plan_exact(plan, mag, n_filt):
    """stage 1 + stage 2 of pw_band_core (kapre_amd/csrc/kpr_mel_pw_kernels.h) for a batch of rows, in the kernel's order of
    operations and float32 roundings: mag (rows, K) -> (rows, n_filt)"""
    from test_band_plan import pw_zero_word
    L, NR, CMQ = plan["L"], plan["NR"], plan["CMQ"]
    nc = 16 * L
    n = mag.shape[0]
    nb = mag.shape[1] - 1                                              # bins below Nyquist (<= nc: a padded plan's other bins are zeros)
    zero_b = 4 * pw_zero_word(nc)
    rowb = np.zeros((n, zero_b + 16), np.uint8)                        # the partial-sum list, then the zero words
    mags = np.concatenate([mag[:, :nb], np.zeros((n, nc - nb), np.float32)], axis=1).reshape(n, L, 16)
    ptr = plan["p"].astype(np.int64).copy()
    acc = np.zeros((n, L, 2), np.float32)
    for i in range(16):
        j, e = i // 2, i % 2
        w = plan["t1"][j, :, 2 * e:2 * e + 2][None]                   # (1, L, 2): (w0, w1) of bin i per lane
        acc = _fma32(mags[:, :, i:i + 1], w, acc)
        mask = int(plan["em"][i])
        for fl in range(L):
            if (mask >> fl) & 1:
                rowb[:, ptr[fl]:ptr[fl] + 8] = np.ascontiguousarray(acc[:, fl]).view(np.uint8)
                acc[:, fl] = 0.0
                ptr[fl] += 8
    out = np.zeros((n, NR * L), np.float32)
    rd = lambda off: np.ascontiguousarray(rowb[:, off:off + 4]).view(np.float32)[:, 0]
    for r in range(NR):
        for fl in range(L):
            u = np.zeros(n, np.float32)
            d = np.zeros(n, np.float32)
            for q in range(CMQ):
                for e in range(4):
                    o_ = int(plan["t2"][r, q, fl, e])
                    u = (u + rd(o_ & 0xffff)).astype(np.float32)
                    d = (d + rd(o_ >> 16)).astype(np.float32)
            out[:, fl + L * r] = _fma32(np.full(n, plan["wn"][r, fl], np.float32), mag[:, nb], (u + d).astype(np.float32))
    return out[:, :n_filt]


@pytest.mark.parametrize("k, n_mels, sr", [(1025, 128, 44100), (513, 80, 16000), (257, 40, 22050), (129, 40, 8000),
                                           (201, 80, 16000), (81, 40, 16000), (1001, 128, 44100), (481, 96, 48000), (13, 8, 16000)])
def test_fb_pw_bit_identical_to_the_band_plan(k, n_mels, sr):
    """pw_band_core is ONE function shared by the fused kernel (k_mel_pw) and by k_fb_pw: executed here on the CPU in its exact order
    of operations (tests/test_band_plan.py parses the plan out of the packed blob), the rows k_fb_pw returns must be the SAME BITS."""
    from kapre_amd import _ffi, backend
    from test_band_plan import parse
    rng = np.random.default_rng(k)
    fb = np.asarray(backend.filterbank_mel(sr, k, n_mels), np.float32)
    plan = parse(_ffi.filterbank_pack(fb, _ffi.filterbank_kranges(fb)), k, n_mels)
    x = (np.abs(rng.standard_normal((3, 2, 37, k))) ** 2).astype(np.float32)
    want = _run_plan_exact(plan, x.reshape(-1, k), n_mels).reshape(3, 2, 37, n_mels)
    got = _layer(k, n_mels, CF, sr=sr)(x).cpu().numpy()
    assert "k_fb_pw<%d>" % (16 * plan["L"]) in _ffi.last_launches()
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    # two interleaved channels (ST instances): the same sums, the channel picked by the multiply-add's op_sel
    got_cl = _layer(k, n_mels, CL, sr=sr)(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cpu().numpy()
    assert "k_fb_pw<%d,st>" % (16 * plan["L"]) in _ffi.last_launches()
    assert np.array_equal(got_cl.transpose(0, 3, 1, 2), want)


def test_fb_pw_other_banks_keep_the_mfma_kernels():
    """no band plan (log-frequency bank, a dense matrix, a stray third non-zero): k_mel_ws<1024, FROM_MAG> / k_gemm as before"""
    from kapre_amd import ApplyFilterbank, _ffi
    rng = np.random.default_rng(3)
    x = np.abs(rng.standard_normal((4, 1, 50, 1025), dtype=np.float32))
    lfb = ApplyFilterbank(type="log", filterbank_kwargs=dict(sample_rate=22050, n_freq=1025), data_format=CF)
    got = lfb(x).cpu().numpy()
    assert "k_fb_pw" not in _ffi.last_launches()
    assert _item_err(got, o.apply_filterbank(x, o.filterbank_log(22050, 1025), CF)) <= 4e-6
    layer = _layer(1025, 128, CF)
    fb = np.array(layer.filterbank, np.float32)
    fb[500, 3] = 0.5                                                                     # a third non-zero in bin 500
    layer.filterbank = fb
    got = layer(x).cpu().numpy()
    assert "k_fb_pw" not in _ffi.last_launches(), _ffi.last_launches()
    assert _item_err(got, x.astype(np.float64) @ fb.astype(np.float64)) <= 4e-6
