"""CPU tests of the host side: layer API parity with the reference (constructor defaults,
get_config keys, error types), host-built constants, C-ABI surface, and product/oracle hygiene."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import kapre_oracle as o
from conftest import REPO, golden_names

import kapre_amd
from kapre_amd import (STFT, InverseSTFT, Magnitude, Phase, MagnitudeToDecibel, ApplyFilterbank,
                       Sequential, Input, backend, composed, _ffi)


# ------------------------------------------------------------------ construction / config parity
def test_stft_defaults_match_reference():
    s = STFT()
    assert (s.n_fft, s.win_length, s.hop_length) == (2048, 2048, 512)     # hop = win // 4
    assert s.window_name is None and not s.pad_begin and not s.pad_end
    assert s.input_data_format == s.output_data_format == "channels_last"  # keras default
    s = STFT(n_fft=1000, win_length=512)
    assert s.hop_length == 128


@pytest.mark.parametrize("name", golden_names("stft"))
def test_stft_get_config_equals_reference(golden, name):
    kw, _, _, extra = golden.get(name)
    ref_cfg = extra["config"]
    cfg = STFT(**kw, name=ref_cfg["name"]).get_config()
    assert set(cfg) == set(ref_cfg)
    for k in ref_cfg:
        assert cfg[k] == ref_cfg[k], k
    # config round trip
    again = STFT.from_config(cfg)
    assert again.get_config() == cfg


def test_istft_get_config_equals_reference(golden):
    name = golden_names("roundtrip")[0]
    _, _, _, extra = golden.get(name)
    ref_cfg = extra["istft_config"]
    kw = {k: ref_cfg[k] for k in ("n_fft", "win_length", "hop_length", "forward_window_name",
                                  "input_data_format", "output_data_format")}
    cfg = InverseSTFT(**kw, name=ref_cfg["name"]).get_config()
    assert cfg == ref_cfg


def test_other_layer_configs():
    assert MagnitudeToDecibel().get_config().items() >= {"ref_value": 1.0, "amin": 1e-5,
                                                         "dynamic_range": 80.0}.items()
    fbk = dict(sample_rate=22050, n_freq=257, n_mels=40, f_min=0.0, f_max=8000)
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=fbk, data_format="channels_first")
    cfg = layer.get_config()
    assert cfg["type"] == "mel" and cfg["filterbank_kwargs"] == fbk
    assert cfg["data_format"] == "channels_first" and layer.freq_axis == 3
    assert ApplyFilterbank(type="mel", filterbank_kwargs=fbk).freq_axis == 2
    assert layer.filterbank.shape == (257, 40) and layer.filterbank.dtype == np.float32
    assert Phase(approx_atan_accuracy=500).get_config()["approx_atan_accuracy"] == 500
    assert not hasattr(ApplyFilterbank(type="other", filterbank_kwargs={}), "filterbank")


def test_error_types_match_reference(golden):
    table = {
        "bad_data_format_value": lambda: STFT(input_data_format="weird"),
        "bad_data_format_type": lambda: STFT(output_data_format=3),
        "bad_window": lambda: STFT(window_name="bartlett"),
        "bad_db_ref": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), ref_value=0.0),
        "bad_db_amin": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), amin=-1.0),
        "bad_db_dr": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), dynamic_range=0.0),
        "bad_log_fmax": lambda: backend.filterbank_log(sample_rate=8000, n_freq=257, n_bins=120),
    }
    for label, fn in table.items():
        want = golden.errors[label]
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == want, label
    with pytest.raises(ValueError):
        InverseSTFT(input_data_format="nope")
    with pytest.raises(TypeError):
        composed.get_melspectrogram_layer(input_data_format=None)


def test_composed_helpers_build_the_reference_layer_lists(golden):
    m = composed.get_melspectrogram_layer(input_shape=(44100, 1), return_decibel=True)
    assert m.name == "melspectrogram"
    assert [type(l).__name__ for l in m.layers] == ["STFT", "Magnitude", "ApplyFilterbank",
                                                    "MagnitudeToDecibel"]
    assert m.layers[2].filterbank_kwargs == {"sample_rate": 22050, "n_freq": 1025, "n_mels": 128,
                                             "f_min": 0.0, "f_max": None, "htk": False,
                                             "norm": "slaney"}
    for name in golden_names("melspectrogram"):
        kw, _, _, extra = golden.get(name)
        assert [type(l).__name__ for l in composed.get_melspectrogram_layer(**kw).layers] == extra["layers"]
    m = composed.get_stft_magnitude_layer()
    assert m.name == "stft_magnitude" and [type(l).__name__ for l in m.layers] == ["STFT", "Magnitude"]
    m = composed.get_log_frequency_spectrogram_layer(return_decibel=True)
    assert m.layers[2].type == "log" and m.layers[2].filterbank.shape == (1025, 84)
    stft, istft = composed.get_perfectly_reconstructing_stft_istft(2048, 512, "channels_last",
                                                                   "channels_first")
    assert stft.pad_begin and stft.pad_end and stft.window_name == "hann_window"
    assert istft.input_data_format == "channels_first" and istft.output_data_format == "channels_last"


def test_sequential_protocol_and_config_round_trip():
    model = Sequential()
    model.add(Input(shape=(8000, 2)))
    model.add(STFT(n_fft=512, hop_length=256, name="stft"))
    model.add(Magnitude())
    assert [type(l).__name__ for l in model.layers] == ["STFT", "Magnitude"]
    clone = Sequential.from_config(model.get_config())
    assert [l.get_config() for l in clone.layers] == [l.get_config() for l in model.layers]
    with pytest.raises(TypeError):
        model.add("not a layer")
    with pytest.raises(TypeError):
        STFT(bogus_kwarg=1)
    # nested composed model inside a user model, as the reference tests do
    outer = Sequential([Input(shape=(8000, 2)), composed.get_melspectrogram_layer(n_fft=512)])
    assert [type(l).__name__ for l in outer._flat_layers()] == ["STFT", "Magnitude", "ApplyFilterbank"]


def test_default_data_format_follows_global_setting():
    try:
        backend.set_image_data_format("channels_first")
        assert STFT().input_data_format == "channels_first"
        assert ApplyFilterbank("mel", dict(sample_rate=22050, n_freq=257)).freq_axis == 3
    finally:
        backend.set_image_data_format("channels_last")
    with pytest.raises(ValueError):
        backend.set_image_data_format("nhwc")


# ------------------------------------------------------------------ host constants vs oracle
@pytest.mark.parametrize("sample_rate", [44100, 22050])
@pytest.mark.parametrize("n_freq", [1025, 257])
@pytest.mark.parametrize("n_mels", [32, 128])
@pytest.mark.parametrize("f_min", [0.0, 200])
@pytest.mark.parametrize("f_max_ratio", [1.0, 0.5])
@pytest.mark.parametrize("htk", [True, False])
@pytest.mark.parametrize("norm", [None, "slaney", 1.0])
def test_mel_filterbank_equals_oracle(sample_rate, n_freq, n_mels, f_min, f_max_ratio, htk, norm):
    # parametrisation of the reference's tests/test_backend.py:43-75 (tolerance there: rtol 1e-7)
    f_max = int(f_max_ratio * (sample_rate // 2))
    kw = dict(sample_rate=sample_rate, n_freq=n_freq, n_mels=n_mels, f_min=f_min, f_max=f_max,
              htk=htk, norm=norm)
    fb = backend.filterbank_mel(**kw)
    assert fb.dtype == np.float32 and fb.shape == (n_freq, n_mels)
    np.testing.assert_allclose(fb, o.filterbank_mel(**kw), rtol=1e-7)


@pytest.mark.parametrize("name", golden_names("filterbank_mel"))
def test_mel_filterbank_equals_reference_run(golden, name):
    kw, _, y, _ = golden.get(name)
    np.testing.assert_allclose(backend.filterbank_mel(**kw), y, rtol=1e-7)


@pytest.mark.parametrize("n_bins,bpo", [(32, 12), (84, 12), (48, 24)])
def test_log_filterbank_equals_oracle(n_bins, bpo):
    kw = dict(sample_rate=22050, n_freq=1025, n_bins=n_bins, bins_per_octave=bpo)
    np.testing.assert_allclose(backend.filterbank_log(**kw), o.filterbank_log(**kw), rtol=1e-7)


@pytest.mark.parametrize("n", [1, 7, 200, 511, 512, 2018, 2048])
def test_windows_equal_oracle(n):
    for name in ("hann_window", "hamming_window", "kaiser_window", "vorbis_window", None):
        np.testing.assert_allclose(backend.get_window_fn(name)(n), o.get_window(name, n), atol=1e-7)
        assert backend.get_window_fn(name)(n).dtype == np.float32
    if n % 2 == 0:
        np.testing.assert_allclose(backend.get_window_fn("kaiser_bessel_derived_window")(n),
                                   o.kaiser_bessel_derived_window(n), atol=1e-7)


@pytest.mark.parametrize("win,hop", [(2048, 512), (1024, 256), (400, 100), (511, 100), (512, 512)])
def test_inverse_window_equals_oracle(win, hop):
    got = backend.inverse_stft_window_fn(hop, backend.hann_window)(win)
    want = o.inverse_stft_window(win, hop, o.hann_window(win))
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=2e-6, atol=1e-7)
    assert (np.isfinite(got) == ok).all()


# ------------------------------------------------------------------ C ABI surface
def _header_functions():
    text = open(os.path.join(REPO, "include", "kapre_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kpr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from kapre_amd import build

    build.build()                      # hipcc cross-compiles for gfx950 without a GPU
    names = _header_functions()
    assert len(names) >= 14
    handle = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "missing export %s" % n
    assert set(names) == set(_ffi.EXPORTS), "ctypes table and header disagree"
    assert _ffi.lib().kpr_version() == 120


def test_host_only_abi_calls():
    L = _ffi.lib()
    g = _ffi.StftGeom(4, 1, 16000, 512, 512, 256, 0, 0, 0, 0)
    assert L.kpr_num_frames(ctypes.byref(g)) == 61
    g = _ffi.StftGeom(128, 1, 110250, 1024, 1024, 256, 1, 1, 1, 1)
    assert L.kpr_num_frames(ctypes.byref(g)) == 434
    g = _ffi.StftGeom(1, 1, 100, 512, 512, 256, 0, 0, 0, 0)
    assert L.kpr_num_frames(ctypes.byref(g)) == 0
    bad = _ffi.StftGeom(1, 0, 100, 512, 512, 256, 0, 0, 0, 0)
    assert L.kpr_num_frames(ctypes.byref(bad)) == -1 and b"channels" in L.kpr_last_error()
    # one (batch item) signal is addressed with 32-bit element offsets: 2^30 elements are refused
    huge = _ffi.StftGeom(1, 2, 1 << 29, 512, 512, 256, 0, 0, 1, 1)
    assert L.kpr_num_frames(ctypes.byref(huge)) == -1 and b"2^30" in L.kpr_last_error()
    ok = _ffi.StftGeom(1, 2, (1 << 29) - 1, 512, 512, 256, 0, 0, 1, 1)
    assert L.kpr_num_frames(ctypes.byref(ok)) > 0
    assert L.kpr_fft_fast_path(2048) == 1 and L.kpr_fft_fast_path(1000) == 0
    fb = backend.filterbank_mel(44100, 1025, 128)
    kr = _ffi.filterbank_kranges(fb).reshape(-1, 2)
    assert kr.shape == (8, 2) and (kr % 4 == 0).all() and kr[-1, 1] == 1028
    for t, (lo, hi) in enumerate(kr):
        blk = fb[:, 16 * t:16 * t + 16]
        assert not blk[:lo].any() and not blk[hi:].any()
        nz = np.nonzero(blk.any(axis=1))[0]
        assert lo <= nz[0] < lo + 4 and hi - 4 < nz[-1] + 1 <= hi
    dense = _ffi.filterbank_kranges(np.ones((257, 20), np.float32)).reshape(-1, 2)
    assert dense.tolist() == [[0, 260], [0, 260]]


def test_filterbank_pack_layout():
    """kpr_filterbank_pack: MFMA-fragment order copy of the filterbank.  Every nonzero entry must
    land exactly once at the documented position (include/kapre_hip.h)."""
    for shape, kw in (((1025, 128), dict(sample_rate=44100, n_freq=1025, n_mels=128)),
                      ((513, 80), dict(sample_rate=16000, n_freq=513, n_mels=80)),
                      ((257, 40), dict(sample_rate=22050, n_freq=257, n_mels=40, f_max=8000))):
        fb = backend.filterbank_mel(**kw)
        assert fb.shape == shape
        kr = _ffi.filterbank_kranges(fb)
        blob = _ffi.filterbank_pack(fb, kr)
        hdr = blob[:_ffi.PACK_HEADER_FLOATS].view(np.uint32)
        assert hdr[0] == 0x4b504642 and tuple(hdr[1:4]) == fb.shape + ((fb.shape[1] + 15) // 16,)
        pk = blob[_ffi.PACK_HEADER_FLOATS:_ffi.PACK_HEADER_FLOATS + int(hdr[4]) * 512]      # the MFMA fragments
        # round 4: the band plan of k_mel_pw follows the fragments (header words 6..11; tests/test_band_plan.py executes it)
        assert hdr[6] == _ffi.PACK_HEADER_FLOATS + pk.size and hdr[7] == (fb.shape[0] - 1) // 16 and not hdr[12:].any()
        assert hdr[6] + hdr[11] <= blob.size
        assert pk.size % 512 == 0
        assert np.count_nonzero(pk) == np.count_nonzero(fb)
        np.testing.assert_allclose(np.sort(pk[pk != 0]), np.sort(fb[fb != 0]), rtol=0, atol=0)
        # decode: chunk c of the stream, half g, lane l, s -> (row, filter); tiles in natural order,
        # each tile's rows padded to whole 32-row chunks inside [0, roundup(K, 32)]
        n_freq, n_filt = fb.shape
        cap = (n_freq + 31) // 32 * 32
        pos = 0
        rebuilt = np.zeros_like(fb)
        for t in range((n_filt + 15) // 16):
            lo, hi = int(kr[2 * t]) & ~7, int(kr[2 * t + 1])
            need = max(32, (hi - lo + 31) // 32 * 32)
            hi = min(cap, lo + need)
            lo = max(0, hi - need)
            for c in range((hi - lo) // 32):
                blk = pk[pos:pos + 512].reshape(2, 64, 4)
                pos += 512
                for g in range(2):
                    for l in range(64):
                        for s4 in range(4):
                            k, m = lo + 32 * c + 16 * g + 4 * s4 + (l >> 4), 16 * t + (l & 15)
                            if k < n_freq and m < n_filt:
                                rebuilt[k, m] = blk[g, l, s4]
                            else:
                                assert blk[g, l, s4] == 0.0
        assert pos == pk.size
        assert np.array_equal(rebuilt, fb)
    dense = np.arange(257 * 20, dtype=np.float32).reshape(257, 20) + 1
    blob = _ffi.filterbank_pack(dense, None)
    pk = blob[_ffi.PACK_HEADER_FLOATS:]
    assert np.count_nonzero(pk) == dense.size
    # the header carries a hash of the k-ranges: other ranges, other tag
    a = _ffi.filterbank_pack(fb, _ffi.filterbank_kranges(fb))[:8].view(np.uint32)
    b = _ffi.filterbank_pack(fb, None)[:8].view(np.uint32)
    assert a[5] != b[5] and a[1] == b[1]


def test_options_api():
    """kpr_set_option / kpr_get_option: the library's only process-wide switches (it reads no environment
    variables)."""
    L = _ffi.lib()
    assert _ffi.set_option("mel_variant", 4) == 0
    assert _ffi.set_option("mel_variant", 0) == 4
    # kernels removed in round 5: their option values are rejected, not silently remapped
    assert L.kpr_set_option(b"mel_variant", 1) == -1 and b"removed" in L.kpr_last_error()
    assert L.kpr_set_option(b"stft_variant", 2) == -1 and b"removed" in L.kpr_last_error()
    assert L.kpr_set_option(b"mel_variant", 9) == -1 and b"outside" in L.kpr_last_error()
    assert L.kpr_set_option(b"istft_path", 4) == 0 and L.kpr_set_option(b"istft_path", 0) == 0
    assert L.kpr_set_option(b"istft_path", 5) == -1 and b"outside" in L.kpr_last_error()
    assert L.kpr_set_option(b"no_such_switch", 1) == -1 and b"unknown option" in L.kpr_last_error()
    src = open(os.path.join(REPO, "kapre_amd", "csrc", "kapre_hip.hip")).read()
    assert "getenv" not in src


def test_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        composed.get_melspectrogram_layer(n_fft=512)(np.zeros((1, 4000, 1), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "kapre_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                for line in text.splitlines():
                    if re.match(r"\s*(import|from)\s+.*oracle", line):
                        raise AssertionError("%s imports the oracle: %s" % (f, line))
    code = ("import sys; import kapre_amd; "
            "bad=[m for m in sys.modules if 'oracle' in m or 'proto_stockham' in m]; "
            "assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=REPO)


def test_row4_layers_config_and_errors(golden):
    """Frame / Energy / Delta / LogmelToMFCC: get_config keys and construction-time errors equal the
    reference run (tests/golden json: the reference's own classes executed on the numpy stubs)."""
    from kapre_amd import Frame, Energy, LogmelToMFCC, Delta
    for name in golden.names("frame") + golden.names("energy") + golden.names("logmel_to_mfcc") + golden.names("delta"):
        kw, _, _, extra = golden.get(name)
        if "config" not in extra:
            continue
        cls = {"frame": Frame, "energy": Energy, "mfcc": LogmelToMFCC, "delta": Delta}[name.split("_")[0]]
        got = cls(**kw).get_config()
        want = extra["config"]
        for k, v in want.items():
            if k in ("name", "dtype", "trainable"):
                continue
            assert got[k] == v, (name, k, got[k], v)
        assert set(want) - {"name", "dtype", "trainable"} <= set(got)
    errs = golden.errors
    cases = {
        "delta_win_small": lambda: Delta(win_length=1),
        "delta_win_even": lambda: Delta(win_length=4),
        "delta_bad_mode": lambda: Delta(mode="wrap"),
        "frame_len_zero": lambda: Frame(frame_length=0, hop_length=1),
        "frame_hop_zero": lambda: Frame(frame_length=4, hop_length=0),
        "frame_hop_gt_len": lambda: Frame(frame_length=4, hop_length=8),
        "frame_bad_format": lambda: Frame(frame_length=4, hop_length=2, data_format="nope"),
    }
    for label, fn in cases.items():
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == errs[label], label
    L = _ffi.lib()
    assert L.kpr_frame_count(1000, 50, 25, 0) == 39 and L.kpr_frame_count(1000, 50, 25, 1) == 40
    assert L.kpr_frame_count(10, 50, 25, 0) == 0 and L.kpr_frame_count(10, 50, 0, 0) == -1
    from kapre_amd.signal import mfcc_matrix
    import kapre_oracle as o
    np.testing.assert_allclose(mfcc_matrix(128, 20), o.mfcc_matrix(128, 20), rtol=1e-6, atol=1e-7)


def test_fft_plan_classification_of_every_transform_size():
    """kpr_fft_plan (host only): the FFT family of each n_fft, and that the O(n_fft^2) DFT-as-GEMM fallback is
    left with exactly the sizes that have a prime factor above 64."""
    from kapre_amd import _ffi
    L = _ffi.lib()

    def largest_prime(n):
        p, f = 1, 2
        while f * f <= n:
            while n % f == 0:
                p, n = f, n // f
            f += 1
        return max(p, n) if n > 1 else p

    assert L.kpr_fft_plan(1, 1) < 0 and L.kpr_fft_plan(512, 0) < 0
    for n in (256, 512, 1024, 2048):
        assert L.kpr_fft_plan(n, n) == _ffi.FFT_POW2
    for n in (160, 200, 320, 400, 640, 800, 1000):
        assert L.kpr_fft_plan(n, n) == _ffi.FFT_MIXED_RADIX
    for n in (96, 120, 192, 240, 360, 384, 480, 600, 720, 768, 960):
        assert L.kpr_fft_plan(n, n) == _ffi.FFT_TWO_PASS
    for n in (12, 100, 300, 1022, 128, 64):
        assert L.kpr_fft_plan(n, n) == _ffi.FFT_BLUESTEIN
    assert L.kpr_fft_plan(4096, 4096) == L.kpr_fft_plan(8192, 4097) == _ffi.FFT_SUB_FFT
    for n in (15, 77, 1001, 1155, 1200, 1280, 1536, 2000, 3000, 6000, 513, 2050):
        assert L.kpr_fft_plan(n, n) == _ffi.FFT_GENERIC, n
    assert L.kpr_fft_plan(2049, 2049) == _ffi.FFT_DFT_GEMM            # 3 x 683
    assert L.kpr_fft_plan(1200, 1201) == L.kpr_fft_plan(1200, 1200) == _ffi.FFT_GENERIC     # win_length > n_fft: cropped frames, the size's own FFT
    assert L.kpr_fft_plan(400, 512) == _ffi.FFT_MIXED_RADIX and L.kpr_fft_plan(300, 999) == _ffi.FFT_BLUESTEIN
    for n in range(2, 4200):
        plan = L.kpr_fft_plan(n, n)
        assert plan >= 0
        if plan == _ffi.FFT_DFT_GEMM:
            assert largest_prime(n) > 64, n
        if largest_prime(n) <= 64:
            assert plan != _ffi.FFT_DFT_GEMM, n


# ---------------------------------------------------------------------------------------------
# the autograd entry (kapre_amd/autograd.py): host-side decisions that need no GPU
# ---------------------------------------------------------------------------------------------
def test_needs_grad_follows_torch_semantics():
    import torch
    from kapre_amd import autograd
    x = torch.ones(2, 100, 1, requires_grad=True)
    assert autograd.needs_grad(x)
    assert not autograd.needs_grad(x.detach())
    assert not autograd.needs_grad(np.ones((2, 100, 1), np.float32))
    with torch.no_grad():
        assert not autograd.needs_grad(x)


def test_gradient_request_without_a_gpu_fails_loudly():
    """A tensor that requires grad is never routed to a silent non-differentiable path: without a HIP device the layer
    raises the same 'no HIP device' error as the forward-only call (there is no CPU fallback in either direction)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    import kapre_amd as kapre
    x = torch.ones(1, 1000, 1, requires_grad=True)
    for layer in (kapre.STFT(n_fft=256, hop_length=64), kapre.MagnitudeToDecibel(), kapre.Delta(win_length=5)):
        with pytest.raises(RuntimeError, match='HIP device'):
            layer(x if not isinstance(layer, (kapre.MagnitudeToDecibel, kapre.Delta)) else torch.ones(1, 5, 7, 1, requires_grad=True))
