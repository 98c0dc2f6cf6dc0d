"""Backend operations -- host-side mirror of /root/reference/kapre/backend.py for the hot path.

Same names, argument meaning and error behaviour as the reference:
``get_window_fn`` (backend.py:58-100), ``validate_data_format_str`` (:103-123),
``magnitude_to_decibel`` (:126-194), ``filterbank_mel`` (:197-231), ``filterbank_log`` (:234-299).

Constants that the reference builds once on the host through TensorFlow/librosa (windows, the mel
and log filterbanks) are built here once on the host in numpy (float64 arithmetic, float32
result, following the published tf.signal / librosa definitions); everything that touches the
signal itself runs in the HIP kernels behind ``kapre_amd._ffi``.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Union

import numpy as np

from . import _ffi

_CH_FIRST_STR = 'channels_first'
_CH_LAST_STR = 'channels_last'
_CH_DEFAULT_STR = 'default'

# Keras' global image_data_format(); the reference resolves 'default' through it
# (time_frequency.py:142-144) and itself falls back to 'channels_last' (backend.py:36-37).
_IMAGE_DATA_FORMAT = _CH_LAST_STR


def image_data_format() -> str:
    return _IMAGE_DATA_FORMAT


def set_image_data_format(data_format: str) -> None:
    global _IMAGE_DATA_FORMAT
    if data_format not in (_CH_FIRST_STR, _CH_LAST_STR):
        raise ValueError('Unknown data_format: %r' % (data_format,))
    _IMAGE_DATA_FORMAT = data_format


def _get_image_data_format() -> str:
    return image_data_format()


def _get_floatx() -> str:
    return 'float32'


# --------------------------------------------------------------------------------------
# windows: tf.signal.*_window(window_length, periodic=True, dtype=float32)
# --------------------------------------------------------------------------------------
# (``dtype``: tf.signal window functions take the dtype of the framed signal; float64 layers ask for float64)
def _raised_cosine_window(window_length: int, a: float, b: float, dtype=np.float32) -> np.ndarray:
    # tf.signal: n = window_length + periodic*even - 1 (periodic only affects even lengths)
    if window_length == 1:
        return np.ones(1, dtype=dtype)
    even = 1 - window_length % 2
    n = float(window_length + even - 1)
    count = np.arange(window_length, dtype=np.float64)
    return (a - b * np.cos(2.0 * np.pi * count / n)).astype(dtype)


def hann_window(window_length: int, dtype=np.float32) -> np.ndarray:
    return _raised_cosine_window(int(window_length), 0.5, 0.5, dtype)


def hamming_window(window_length: int, dtype=np.float32) -> np.ndarray:
    return _raised_cosine_window(int(window_length), 0.54, 0.46, dtype)


def kaiser_window(window_length: int, beta: float = 12.0, dtype=np.float32) -> np.ndarray:
    window_length = int(window_length)
    if window_length == 1:
        return np.ones(1, dtype=dtype)
    halflen = (window_length - 1) / 2.0
    arg = np.arange(window_length, dtype=np.float64) - halflen
    arg = beta * np.sqrt(np.maximum(0.0, 1.0 - (arg / halflen) ** 2))
    return (np.i0(arg) / np.i0(beta)).astype(dtype)


def kaiser_bessel_derived_window(window_length: int, beta: float = 12.0, dtype=np.float32) -> np.ndarray:
    window_length = int(window_length)
    halflen = window_length // 2
    kw = kaiser_window(halflen + 1, beta, dtype=dtype).astype(np.float64)
    csum = np.cumsum(kw)
    half = np.sqrt(csum[:-1] / csum[-1])
    return np.concatenate([half, half[::-1]]).astype(dtype)


def vorbis_window(window_length: int, dtype=np.float32) -> np.ndarray:
    window_length = int(window_length)
    arg = np.arange(window_length, dtype=np.float64) + 0.5
    return np.sin(np.pi / 2.0 * np.sin(np.pi / window_length * arg) ** 2).astype(dtype)


_AVAILABLE_WINDOWS = {
    'hamming_window': hamming_window,
    'hann_window': hann_window,
    'kaiser_bessel_derived_window': kaiser_bessel_derived_window,
    'kaiser_window': kaiser_window,
    'vorbis_window': vorbis_window,
}


def get_window_fn(window_name: Optional[str] = None) -> Callable[[int], np.ndarray]:
    """Return a window function given its name (reference: backend.py:58-100).

    ``None`` -> hann.  Unknown names raise ``NotImplementedError`` like the reference.
    The returned callable maps a window length to a float32 numpy array.
    """
    if window_name is None:
        return hann_window
    if window_name not in _AVAILABLE_WINDOWS:
        raise NotImplementedError(
            'Window name %s is not supported now. Currently, %d windows are'
            'supported - %s'
            % (window_name, len(_AVAILABLE_WINDOWS), ', '.join(_AVAILABLE_WINDOWS.keys()))
        )
    return _AVAILABLE_WINDOWS[window_name]


def window_values(window_fn, length: int, dtype=np.float32) -> np.ndarray:
    """``window_fn(length)`` in ``dtype``: the built-in windows take the dtype (as tf.signal's do); a user
    callable that only accepts the length is evaluated as is and cast."""
    if dtype == np.float32:
        return np.asarray(window_fn(length), dtype=np.float32)
    try:
        return np.asarray(window_fn(length, dtype=dtype), dtype=dtype)
    except TypeError:
        return np.asarray(window_fn(length), dtype=dtype)


def inverse_stft_window_fn(frame_step: int, forward_window_fn: Callable[[int], np.ndarray]):
    """tf.signal.inverse_stft_window_fn (used at time_frequency.py:278-280)."""

    def _fn(frame_length: int, dtype=np.float32) -> np.ndarray:
        fw = np.asarray(window_values(forward_window_fn, frame_length, dtype), dtype=dtype)
        denom = np.square(fw)
        overlaps = -(-frame_length // frame_step)
        denom = np.pad(denom, (0, overlaps * frame_step - frame_length))
        denom = denom.reshape(overlaps, frame_step).sum(0, keepdims=True)
        denom = np.tile(denom, (overlaps, 1)).reshape(overlaps * frame_step)
        with np.errstate(divide='ignore', invalid='ignore'):
            return (fw / denom[:frame_length]).astype(dtype)

    return _fn


def validate_data_format_str(data_format: str) -> None:
    """Reference: backend.py:103-123 (TypeError for non-str, ValueError for unknown values)."""
    if not isinstance(data_format, str):
        raise TypeError(
            f'data_format must be a string, got {type(data_format).__name__}: {data_format}'
        )
    if data_format not in (_CH_DEFAULT_STR, _CH_FIRST_STR, _CH_LAST_STR):
        raise ValueError(
            f'data_format must be one of {[_CH_FIRST_STR, _CH_LAST_STR, _CH_DEFAULT_STR]}, '
            f'got: {data_format!r}'
        )


# --------------------------------------------------------------------------------------
# decibel
# --------------------------------------------------------------------------------------
def magnitude_to_decibel(x, ref_value: float = 1.0, amin: float = 1e-5,
                         dynamic_range: float = 80.0):
    """Decibel scaling on the GPU (reference: backend.py:126-194).

    ``10*log10(max(x, amin)) - 10*log10(max(amin, ref_value))`` clamped from below at
    (per batch item max) - dynamic_range; a rank-1 input is one item.  Raises ``ValueError`` for
    non-positive parameters exactly like the reference (:168-173).
    Accepts a numpy array or torch tensor; returns a float32 torch tensor on the GPU (float64 for a
    float64 input: the TF ops of the reference follow the dtype of their argument).
    """
    if ref_value <= 0:
        raise ValueError(f'ref_value must be positive, got: {ref_value}')
    if amin <= 0:
        raise ValueError(f'amin must be positive, got: {amin}')
    if dynamic_range <= 0:
        raise ValueError(f'dynamic_range must be positive, got: {dynamic_range}')
    import torch

    f64 = _ffi.is_f64(x)            # tf ops compute in the dtype of their input (float64 layers hand in float64)
    xt = _ffi.as_device_dtype(x, torch.float64) if f64 else _ffi.as_device_f32(x)
    out = torch.empty_like(xt)
    if xt.dim() > 1:
        n_items = xt.shape[0]
        item_size = xt.numel() // max(n_items, 1)
    else:
        n_items, item_size = 1, xt.numel()
    L = _ffi.lib()
    if f64:
        with torch.cuda.device(xt.device):
            _ffi.check(L.kpr_mag_to_db_f64(_ffi.ptr(xt), n_items, item_size, float(ref_value), float(amin),
                                           float(dynamic_range), _ffi.ptr(out), _ffi.current_stream_ptr()),
                       'kpr_mag_to_db_f64')
        return out
    ws_bytes = int(L.kpr_db_workspace_bytes(n_items))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xt.device)
    db = _ffi.DbParams(1, float(ref_value), float(amin), float(dynamic_range))
    with torch.cuda.device(xt.device):
        _ffi.check(L.kpr_mag_to_db_f32(_ffi.ptr(xt), n_items, item_size, ctypes.byref(db),
                                       _ffi.ptr(out), _ffi.ptr(ws), ws_bytes,
                                       _ffi.current_stream_ptr()), 'kpr_mag_to_db_f32')
    return out


# --------------------------------------------------------------------------------------
# filterbanks (host constants)
# --------------------------------------------------------------------------------------
def _hz_to_mel(frequencies, htk: bool):
    frequencies = np.asanyarray(frequencies, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + frequencies / 700.0)
    f_sp = 200.0 / 3
    mels = frequencies / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels, htk: bool):
    mels = np.asanyarray(mels, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def filterbank_mel(sample_rate: int, n_freq: int, n_mels: int = 128, f_min: float = 0.0,
                   f_max: Optional[float] = None, htk: bool = False,
                   norm: Union[str, int, float, None] = 'slaney') -> np.ndarray:
    """Mel filterbank, shape (n_freq, n_mels), float32 (reference: backend.py:197-231, which
    wraps ``librosa.filters.mel(sr, n_fft=(n_freq-1)*2, ...).astype(floatx).T``)."""
    n_fft = (n_freq - 1) * 2
    if f_max is None:
        f_max = float(sample_rate) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sample_rate)
    mel_pts = np.linspace(_hz_to_mel(f_min, htk), _hz_to_mel(f_max, htk), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts, htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if isinstance(norm, str):
        if norm == 'slaney':
            enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
            weights *= enorm[:, np.newaxis]
        else:
            raise ValueError(f'Unsupported norm={norm}')
    elif norm is not None:
        weights = _normalize(weights, norm=norm, axis=-1)
    return np.ascontiguousarray(weights.astype(_get_floatx()).T)


def _normalize(S: np.ndarray, norm, axis: int) -> np.ndarray:
    """librosa.util.normalize (fill=None): rows with norm below `tiny` are left unscaled."""
    mag = np.abs(S).astype(float)          # librosa computes the norm in float64
    if norm == np.inf:
        length = np.max(mag, axis=axis, keepdims=True)
    elif norm > 0:
        length = np.sum(mag ** norm, axis=axis, keepdims=True) ** (1.0 / norm)
    else:
        raise ValueError(f'Unsupported norm: {norm!r}')
    length = np.where(length < np.finfo(S.dtype).tiny, 1.0, length)
    return (S / length).astype(S.dtype)    # Snorm = np.empty_like(S); Snorm[:] = S / length


def filterbank_log(sample_rate: int, n_freq: int, n_bins: int = 84, bins_per_octave: int = 12,
                   f_min: Optional[float] = None, spread: float = 0.125) -> np.ndarray:
    """Log-frequency filterbank, shape (n_freq, n_bins), float32 (reference: backend.py:234-299)."""
    if f_min is None:
        f_min = 32.70319566
    f_max = f_min * 2 ** (n_bins / bins_per_octave)
    if f_max > sample_rate // 2:
        raise RuntimeError(
            'Maximum frequency of log filterbank should be lower or equal to the maximum'
            'frequency of the input (defined by its sample rate), '
            'but f_max=%f and maximum frequency is %f. \n'
            'Fix it by reducing n_bins, increasing bins_per_octave and/or reducing f_min.\n'
            'You can also do it by increasing sample_rate but it means you need to upsample'
            'the input audio data, too.' % (f_max, sample_rate)
        )
    sigma = float(spread) / bins_per_octave
    basis = np.zeros((n_bins, n_freq))
    fft_freqs = np.fft.rfftfreq(n=(n_freq - 1) * 2, d=1.0 / sample_rate)
    log_freqs = np.log2(fft_freqs[1:])
    for i in range(n_bins):
        c_freq = f_min * (2.0 ** (float(i) / bins_per_octave))
        basis[i, 1:] = np.exp(
            -0.5 * ((log_freqs - np.log2(c_freq)) / sigma) ** 2 - np.log2(sigma) - log_freqs
        )
    basis = _normalize(basis, norm=1, axis=1)
    return np.ascontiguousarray(basis.astype(_get_floatx()).T)
