"""The C ABI used WITHOUT torch or Python in the data path: tests/abi/test_abi.cc (hipMalloc + raw hipStream_t,
built by kapre_amd/build.py) replays reference-run golden cases dumped to .bin files."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, golden_names
from kapre_amd import backend

EXE = os.path.join(REPO, "tests", "abi", "test_abi")
LAYOUT = {"channels_first": 0, "channels_last": 1, "default": 1}


def _geom(kw, x, time_axis_last):
    fmt_in = kw.get("input_data_format", "default")
    if LAYOUT[fmt_in] == 1:
        b, t, c = x.shape
    else:
        b, c, t = x.shape
    n_fft = kw.get("n_fft", 2048)
    win = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or win // 4
    return b, c, t, n_fft, win, hop


def test_binary_is_built_and_links_only_the_abi():
    assert os.path.exists(EXE), "tests/abi/test_abi missing: run python -m kapre_amd.build"
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libkapre_hip.so" in out and "torch" not in out and "python" not in out.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in golden_names("melspectrogram")][:4])
def test_mel_through_the_abi_without_torch(golden, name, tmp_path):
    kw, x, y, _ = golden.get(name)
    b, c, t, n_fft, win, hop = _geom(kw, x, False)
    n_mels = kw.get("n_mels", 128)
    fb = np.asarray(backend.filterbank_mel(kw.get("sample_rate", 22050), n_fft // 2 + 1, n_mels, kw.get("mel_f_min", 0.0),
                                           kw.get("mel_f_max"), kw.get("mel_htk", False), kw.get("mel_norm", "slaney")), np.float32)
    window = np.asarray(backend.get_window_fn(kw.get("window_name"))(win), np.float32)
    lay_out = LAYOUT[kw.get("output_data_format", "default")]
    n_frames = y.shape[1] if lay_out == 1 else y.shape[2]
    db = bool(kw.get("return_decibel"))
    meta = np.array([b, c, t, n_fft, win, hop, int(kw.get("pad_begin", False)), int(kw.get("pad_end", False)),
                     LAYOUT[kw.get("input_data_format", "default")], lay_out, n_mels, int(db), n_frames, 0, 0, 0], np.int64)
    dbp = np.array([kw.get("db_ref_value", 1.0), kw.get("db_amin", 1e-5), kw.get("db_dynamic_range", 80.0), 0], np.float32)
    path = tmp_path / "mel.bin"
    with open(path, "wb") as f:
        for a in (meta, dbp, x.astype(np.float32), window, fb, y.astype(np.float32)):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, "mel", str(path), "1e-3" if db else "1e-4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("istft")[:3])
def test_istft_through_the_abi_without_torch(golden, name, tmp_path):
    kw, s, y, _ = golden.get(name)
    n_fft = kw.get("n_fft", 2048)
    win = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or win // 4
    lay_in = LAYOUT[kw.get("input_data_format", "default")]
    lay_out = LAYOUT[kw.get("output_data_format", "default")]
    if lay_in == 1:
        b, n_frames, k, c = s.shape
    else:
        b, c, n_frames, k = s.shape
    fwd = backend.get_window_fn(kw.get("forward_window_name"))
    window = np.asarray(backend.inverse_stft_window_fn(hop, fwd)(win), np.float32)
    # kpr_stft_geom describes the SPECTROGRAM side in in_layout... for the inverse: in = spectrogram, out = waveform
    meta = np.array([b, c, 0, n_fft, win, hop, 0, 0, lay_out, lay_in, 0, 0, n_frames, 0, 0, 0], np.int64)
    spec = np.ascontiguousarray(s.astype(np.complex64)).view(np.float32)
    path = tmp_path / "istft.bin"
    with open(path, "wb") as f:
        for a in (meta, spec, window, y.astype(np.float32)):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, "istft", str(path), "1e-4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n_freq, n_mels, rows, batch", [(1025, 128, 83, 6), (201, 80, 300, 2), (513, 40, 7, 1)])
def test_apply_filterbank_through_the_abi_without_torch(n_freq, n_mels, rows, batch, tmp_path):
    """round 6: the stand-alone filterbank entry point from a torch-free C++ client -- the packed blob built through the ABI, the
    banded row kernel named by kpr_last_launches(), float64 numpy as the checker, a NaN bin in the last row"""
    rng = np.random.default_rng(n_freq)
    fb = np.asarray(backend.filterbank_mel(16000, n_freq, n_mels), np.float32)
    x = np.abs(rng.standard_normal((batch, 1, rows, n_freq))).astype(np.float32)
    want = (x.astype(np.float64) @ fb.astype(np.float64)).astype(np.float32)
    x[-1, 0, -1, n_freq // 3] = np.nan
    meta = np.array([batch, 1, 0, n_freq, 0, 1, 0, 0, 0, 0, n_mels, 0, rows, 0, 0, 0], np.int64)
    path = tmp_path / "fb.bin"
    with open(path, "wb") as f:
        for a in (meta, x, fb, want):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, "fb", str(path), "4e-6"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
