"""CPU oracle for Kapre's time-frequency hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``kapre_amd/`` imports it (tests/test_host_api.py::test_product_never_imports_the_oracle enforces that).

What it is
----------
A float64 numpy restatement of the arithmetic that Kapre's layers delegate to TensorFlow
(``tf.signal.*``) and librosa (``librosa.filters.mel``).  Those two packages are NOT vendored
under /root/reference and are not installable here (pins: tensorflow>=2.16,<2.21 and
librosa>=0.11,<1.0 -- /root/reference/setup.py:25-29), so their published algorithms are
restated from their documented definitions; every function cites the Kapre call site
(/root/reference file:line) whose behaviour it reproduces.

Parity pinning status (see DESIGN.md "Oracle"):
  * Kapre-level glue (layouts, pad_begin length, tensordot axes, dB reduction axes, defaults)
    is pinned by executing the REAL reference modules (kapre/backend.py, time_frequency.py,
    composed.py) in this container on top of numpy stand-ins for tensorflow/librosa
    (oracle/ref_stubs, script oracle/make_golden.py) -> tests/golden/*.npz.
  * Known-answer vectors the reference's tests hold (tests/test_backend.py:20-22 dB inputs,
    frame-count formulas tests/test_time_frequency.py:32-39) are checked in tests/.
  * The L0 arithmetic itself (tf.signal / librosa) has no stored golden vectors upstream (the
    reference tests call librosa live); it is cross-checked against independent
    implementations available here: an explicit O(N^2) DFT, scipy.fft, scipy.signal.get_window,
    torch.stft/istft on CPU.  For that layer parity is "pinned by definition + cross-checks",
    i.e. UNPINNED against TF/librosa binaries.
"""
from __future__ import annotations

import math

import numpy as np

CH_FIRST = "channels_first"
CH_LAST = "channels_last"


# --------------------------------------------------------------------------------------
# windows  (kapre/backend.py:58-100 -> tf.signal.*_window(window_length, periodic=True))
# --------------------------------------------------------------------------------------
def _raised_cosine(n: int, a: float, b: float) -> np.ndarray:
    """tf.signal.hann_window / hamming_window with periodic=True.

    TF's rule: denominator D = n + periodic*even - 1, i.e. D = n for even n and D = n-1 for
    odd n (periodic only takes effect for even lengths); n == 1 -> [1].
    """
    if n == 1:
        return np.ones(1, dtype=np.float64)
    even = 1 - (n % 2)
    d = n + even - 1
    k = np.arange(n, dtype=np.float64)
    return a - b * np.cos(2.0 * np.pi * k / d)


def hann_window(n: int) -> np.ndarray:
    return _raised_cosine(n, 0.5, 0.5)


def hamming_window(n: int) -> np.ndarray:
    return _raised_cosine(n, 0.54, 0.46)


def kaiser_window(n: int, beta: float = 12.0) -> np.ndarray:
    """tf.signal.kaiser_window(window_length, beta=12.0): symmetric Kaiser window."""
    if n == 1:
        return np.ones(1, dtype=np.float64)
    k = np.arange(n, dtype=np.float64)
    half = (n - 1) / 2.0
    ratio = (k - half) / half
    arg = beta * np.sqrt(np.maximum(0.0, 1.0 - ratio * ratio))
    return np.i0(arg) / np.i0(beta)


def kaiser_bessel_derived_window(n: int, beta: float = 12.0) -> np.ndarray:
    """tf.signal.kaiser_bessel_derived_window: sqrt of the normalised cumulative Kaiser."""
    half = n // 2
    kw = kaiser_window(half + 1, beta)
    csum = np.cumsum(kw)
    half_w = np.sqrt(csum[:-1] / csum[-1])
    return np.concatenate([half_w, half_w[::-1]])


def vorbis_window(n: int) -> np.ndarray:
    """tf.signal.vorbis_window: sin(pi/2 * sin^2(pi (k+0.5)/n))."""
    k = np.arange(n, dtype=np.float64) + 0.5
    return np.sin(np.pi / 2.0 * np.sin(np.pi / n * k) ** 2)


_WINDOWS = {
    None: hann_window,
    "hann_window": hann_window,
    "hamming_window": hamming_window,
    "kaiser_window": kaiser_window,
    "kaiser_bessel_derived_window": kaiser_bessel_derived_window,
    "vorbis_window": vorbis_window,
}


def get_window(window_name, n: int) -> np.ndarray:
    """kapre/backend.py:58-100; unknown names raise NotImplementedError (:89-98)."""
    if window_name not in _WINDOWS:
        raise NotImplementedError("Window name %s is not supported" % window_name)
    return _WINDOWS[window_name](n)


# --------------------------------------------------------------------------------------
# framing (tf.signal.frame as used by tf.signal.stft; mirrored in-tree at
# kapre/tflite_compatible_stft.py:101 and :176-182)
# --------------------------------------------------------------------------------------
def num_frames(t: int, win: int, hop: int, pad_end: bool) -> int:
    if pad_end:
        return -(-t // hop)
    return max(0, 1 + (t - win) // hop)


def frame(x: np.ndarray, win: int, hop: int, pad_end: bool) -> np.ndarray:
    """(..., T) -> (..., F, win).  pad_end right-pads zeros so that ceil(T/hop) frames exist."""
    t = x.shape[-1]
    f = num_frames(t, win, hop, pad_end)
    need = (f - 1) * hop + win if f > 0 else 0
    if need > t:
        pad = [(0, 0)] * (x.ndim - 1) + [(0, need - t)]
        x = np.pad(x, pad)
    idx = np.arange(f)[:, None] * hop + np.arange(win)[None, :]
    return x[..., idx]


# --------------------------------------------------------------------------------------
# forward STFT: tf.signal.stft(signals, frame_length, frame_step, fft_length, window_fn, pad_end)
# called at kapre/time_frequency.py:174-182
# --------------------------------------------------------------------------------------
def rdft_matrix(n_fft: int) -> np.ndarray:
    """(n_fft, n_fft//2+1) complex128, exp(-2 pi i k n / n_fft) -- same definition as the
    reference's own primitive restatement kapre/tflite_compatible_stft.py:14-35."""
    k = np.arange(n_fft // 2 + 1)
    n = np.arange(n_fft)
    # reduce the product mod n_fft in integers before the trig call: exact angles
    kn = (n[:, None] * k[None, :]) % n_fft
    return np.exp(-2j * np.pi * kn / n_fft)


def tf_stft(x, win: int, hop: int, n_fft: int, window: np.ndarray, pad_end: bool,
            use_matrix: bool = False) -> np.ndarray:
    """(..., T) float -> (..., F, n_fft//2+1) complex128.

    rfft(frames * window, n_fft): frames are RIGHT-zero-padded to n_fft when win < n_fft and
    cropped to n_fft when win > n_fft (tf.signal.rfft semantics; right-padding mirrored at
    kapre/tflite_compatible_stft.py:62-69).  Unnormalised forward DFT.
    """
    x = np.asarray(x, dtype=np.float64)
    fr = frame(x, win, hop, pad_end) * np.asarray(window, dtype=np.float64)
    if win < n_fft:
        fr = np.pad(fr, [(0, 0)] * (fr.ndim - 1) + [(0, n_fft - win)])
    elif win > n_fft:
        fr = fr[..., :n_fft]
    if use_matrix:
        return fr @ rdft_matrix(n_fft)
    return np.fft.rfft(fr, n=n_fft, axis=-1)


def kapre_stft(x, n_fft=2048, win_length=None, hop_length=None, window_name=None,
               pad_begin=False, pad_end=False, input_data_format=CH_LAST,
               output_data_format=CH_LAST, use_matrix=False) -> np.ndarray:
    """STFT.call, kapre/time_frequency.py:146-187.

    x: (B,T,C) for channels_last, (B,C,T) for channels_first.
    returns complex128 (B,F,K,C) or (B,C,F,K).
    """
    if win_length is None:
        win_length = n_fft                      # time_frequency.py:126-127
    if hop_length is None:
        hop_length = win_length // 4            # time_frequency.py:128-129
    x = np.asarray(x, dtype=np.float64)
    if input_data_format == CH_LAST:
        x = np.transpose(x, (0, 2, 1))          # time_frequency.py:164-167
    if pad_begin:
        # NOTE n_fft - hop (code), not win_length - hop (docstring): time_frequency.py:169-172
        x = np.pad(x, [(0, 0), (0, 0), (int(n_fft - hop_length), 0)])
    w = get_window(window_name, win_length)
    s = tf_stft(x, win_length, hop_length, n_fft, w, pad_end, use_matrix=use_matrix)
    if output_data_format == CH_LAST:
        s = np.transpose(s, (0, 2, 3, 1))       # time_frequency.py:184-185
    return s


def magnitude(s) -> np.ndarray:
    """Magnitude.call -> tf.abs, kapre/time_frequency.py:351-359."""
    return np.abs(s)


def phase(s) -> np.ndarray:
    """Phase.call (accurate branch) -> tf.math.angle, kapre/time_frequency.py:402."""
    return np.angle(s)


# --------------------------------------------------------------------------------------
# inverse STFT
# --------------------------------------------------------------------------------------
def inverse_stft_window(win: int, hop: int, forward_window: np.ndarray) -> np.ndarray:
    """tf.signal.inverse_stft_window_fn(frame_step, forward_window_fn)(win)
    built at kapre/time_frequency.py:278-280:
      denom = w^2, padded to ceil(win/hop)*hop, reshaped (overlaps, hop), summed over overlaps,
      tiled back; w_inv = w / denom[:win].
    """
    w = np.asarray(forward_window, dtype=np.float64)
    overlaps = -(-win // hop)
    denom = np.pad(w * w, (0, overlaps * hop - win)).reshape(overlaps, hop).sum(0)
    denom = np.tile(denom, overlaps)[:win]
    with np.errstate(divide="ignore", invalid="ignore"):
        return w / denom


def tf_inverse_stft(s, win: int, hop: int, n_fft: int, synth_window: np.ndarray) -> np.ndarray:
    """tf.signal.inverse_stft as called at kapre/time_frequency.py:307-314.

    (..., F, K) complex -> (..., (F-1)*hop + win) float64.
    irfft(n_fft) (1/n_fft scaling; imaginary parts of DC/Nyquist ignored; K cropped / zero-
    padded to n_fft//2+1) -> crop to win if n_fft > win, right-pad if n_fft < win ->
    * synthesis window -> overlap-add with stride hop.
    """
    s = np.asarray(s, dtype=np.complex128)
    k_need = n_fft // 2 + 1
    if s.shape[-1] > k_need:
        s = s[..., :k_need]
    elif s.shape[-1] < k_need:
        s = np.pad(s, [(0, 0)] * (s.ndim - 1) + [(0, k_need - s.shape[-1])])
    fr = np.fft.irfft(s, n=n_fft, axis=-1)
    if n_fft > win:
        fr = fr[..., :win]
    elif n_fft < win:
        fr = np.pad(fr, [(0, 0)] * (fr.ndim - 1) + [(0, win - n_fft)])
    fr = fr * np.asarray(synth_window, dtype=np.float64)
    f = fr.shape[-2]
    t_out = (f - 1) * hop + win if f > 0 else 0
    out = np.zeros(fr.shape[:-2] + (t_out,), dtype=np.float64)
    for i in range(f):
        out[..., i * hop:i * hop + win] += fr[..., i, :]
    return out


def kapre_istft(s, n_fft=2048, win_length=None, hop_length=None, forward_window_name=None,
                input_data_format=CH_LAST, output_data_format=CH_LAST) -> np.ndarray:
    """InverseSTFT.call, kapre/time_frequency.py:289-319.
    s: (B,F,K,C) channels_last or (B,C,F,K) channels_first.  returns (B,T,C) or (B,C,T)."""
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    s = np.asarray(s)
    if input_data_format == CH_LAST:
        s = np.transpose(s, (0, 3, 1, 2))       # time_frequency.py:304-305
    w = get_window(forward_window_name, win_length)
    w_inv = inverse_stft_window(win_length, hop_length, w)
    y = tf_inverse_stft(s, win_length, hop_length, n_fft, w_inv)
    if output_data_format == CH_LAST:
        y = np.transpose(y, (0, 2, 1))          # time_frequency.py:316-317
    return y


# --------------------------------------------------------------------------------------
# mel filterbank: librosa.filters.mel (librosa 0.11) as called at kapre/backend.py:222-231
# --------------------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f: float, htk: bool) -> float:
    if htk:
        return 2595.0 * math.log10(1.0 + f / 700.0)
    if f >= _MIN_LOG_HZ:
        return _MIN_LOG_MEL + math.log(f / _MIN_LOG_HZ) / _LOGSTEP
    return f / _F_SP


def mel_to_hz(m: float, htk: bool) -> float:
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    if m >= _MIN_LOG_MEL:
        return _MIN_LOG_HZ * math.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return _F_SP * m


def filterbank_mel(sample_rate, n_freq, n_mels=128, f_min=0.0, f_max=None, htk=False,
                   norm="slaney") -> np.ndarray:
    """backend.filterbank_mel (kapre/backend.py:197-231) -> (n_freq, n_mels) float32.

    Written as plain per-element loops on purpose (independent of the vectorised product
    implementation in kapre_amd/backend.py).  Triangles are computed in float64, stored to a
    float32 array, then normalised (slaney: *= 2/(f[i+2]-f[i]) evaluated in float64 on the
    float32-rounded weight and rounded again; numeric p: divide by the float32 p-norm).
    """
    n_fft = (n_freq - 1) * 2
    if f_max is None:
        f_max = float(sample_rate) / 2
    m_lo, m_hi = hz_to_mel(float(f_min), htk), hz_to_mel(float(f_max), htk)
    mels = np.linspace(m_lo, m_hi, n_mels + 2)
    mel_f = np.array([mel_to_hz(float(m), htk) for m in mels], dtype=np.float64)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sample_rate)   # librosa.fft_frequencies
    w = np.zeros((n_mels, n_freq), dtype=np.float32)
    for i in range(n_mels):
        d_lo = mel_f[i + 1] - mel_f[i]
        d_hi = mel_f[i + 2] - mel_f[i + 1]
        for k in range(n_freq):
            lower = -(mel_f[i] - fftfreqs[k]) / d_lo
            upper = (mel_f[i + 2] - fftfreqs[k]) / d_hi
            w[i, k] = max(0.0, min(lower, upper))
    if isinstance(norm, str):
        if norm != "slaney":
            raise ValueError("Unsupported norm=%r" % norm)
        for i in range(n_mels):
            enorm = 2.0 / (mel_f[i + 2] - mel_f[i])
            w[i, :] = (w[i, :].astype(np.float64) * enorm).astype(np.float32)
    elif norm is not None:
        p = float(norm)
        tiny = np.finfo(np.float32).tiny
        for i in range(n_mels):
            row = np.abs(w[i, :]).astype(np.float64)      # librosa.util.normalize: float64 norm
            length = row.max() if p == np.inf else float(np.sum(row ** p) ** (1.0 / p))
            if length < tiny:
                length = 1.0
            w[i, :] = (w[i, :].astype(np.float64) / length).astype(np.float32)
    return np.ascontiguousarray(w.T)


def filterbank_log(sample_rate, n_freq, n_bins=84, bins_per_octave=12, f_min=None,
                   spread=0.125) -> np.ndarray:
    """backend.filterbank_log, kapre/backend.py:234-299 -> (n_freq, n_bins) float32."""
    if f_min is None:
        f_min = 32.70319566
    f_max = f_min * 2 ** (n_bins / bins_per_octave)
    if f_max > sample_rate // 2:                 # backend.py:266-275
        raise RuntimeError("Maximum frequency of log filterbank should be lower or equal to "
                           "the maximum frequency of the input")
    sigma = float(spread) / bins_per_octave
    basis = np.zeros((n_bins, n_freq))
    n_fft = (n_freq - 1) * 2
    freqs = np.arange(n_freq, dtype=np.float64) * (float(sample_rate) / n_fft)
    log_freqs = np.log2(freqs[1:])
    for i in range(n_bins):
        c_freq = f_min * (2.0 ** (float(i) / bins_per_octave))
        basis[i, 1:] = np.exp(-0.5 * ((log_freqs - np.log2(c_freq)) / sigma) ** 2
                              - np.log2(sigma) - log_freqs)
    # librosa.util.normalize(basis, norm=1, axis=1): rows with tiny norm are left unscaled
    length = np.sum(np.abs(basis), axis=1, keepdims=True)
    length[length < np.finfo(basis.dtype).tiny] = 1.0
    basis = (basis / length).astype(np.float32)
    return np.ascontiguousarray(basis.T)


def apply_filterbank(x, fb, data_format=CH_LAST) -> np.ndarray:
    """ApplyFilterbank.call, kapre/time_frequency.py:535-548.
    x: (B,F,K,C) ch_last or (B,C,F,K) ch_first; fb (K, M)."""
    x = np.asarray(x, dtype=np.float64)
    fb = np.asarray(fb, dtype=np.float64)
    axis = 3 if data_format == CH_FIRST else 2
    out = np.tensordot(x, fb, axes=(axis, 0))
    if data_format == CH_LAST:
        out = np.transpose(out, (0, 1, 3, 2))
    return out


# --------------------------------------------------------------------------------------
# decibel: backend.magnitude_to_decibel, kapre/backend.py:126-194
# --------------------------------------------------------------------------------------
def magnitude_to_decibel(x, ref_value=1.0, amin=1e-5, dynamic_range=80.0) -> np.ndarray:
    if ref_value <= 0:
        raise ValueError("ref_value must be positive, got: %r" % (ref_value,))
    if amin <= 0:
        raise ValueError("amin must be positive, got: %r" % (amin,))
    if dynamic_range <= 0:
        raise ValueError("dynamic_range must be positive, got: %r" % (dynamic_range,))
    x = np.asarray(x, dtype=np.float64)
    log10 = lambda v: np.log(v) / np.log(10.0)            # backend.py:175-176
    log_spec = 10.0 * log10(np.maximum(x, amin))
    log_spec = log_spec - 10.0 * log10(np.maximum(amin, ref_value))
    if x.ndim > 1:                                         # backend.py:178-181
        mx = log_spec.max(axis=tuple(range(1, x.ndim)), keepdims=True)
    else:
        mx = log_spec.max(keepdims=True)
    return np.maximum(log_spec, mx - dynamic_range)        # backend.py:190-192


# --------------------------------------------------------------------------------------
# composed: get_melspectrogram_layer / get_stft_magnitude_layer (kapre/composed.py:138-261, 32-135)
# --------------------------------------------------------------------------------------
def kapre_melspectrogram(x, n_fft=2048, win_length=None, hop_length=None, window_name=None,
                         pad_begin=False, pad_end=False, sample_rate=22050, n_mels=128,
                         mel_f_min=0.0, mel_f_max=None, mel_htk=False, mel_norm="slaney",
                         return_decibel=False, db_amin=1e-5, db_ref_value=1.0,
                         db_dynamic_range=80.0, input_data_format=CH_LAST,
                         output_data_format=CH_LAST) -> np.ndarray:
    s = kapre_stft(x, n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                   input_data_format, output_data_format)
    mag = magnitude(s)                                      # composed.py:239
    fb = filterbank_mel(sample_rate, n_fft // 2 + 1, n_mels, mel_f_min, mel_f_max, mel_htk,
                        mel_norm)
    out = apply_filterbank(mag, fb, output_data_format)     # composed.py:241-252
    if return_decibel:                                      # composed.py:255-259
        out = magnitude_to_decibel(out, db_ref_value, db_amin, db_dynamic_range)
    return out


def kapre_stft_magnitude(x, n_fft=2048, win_length=None, hop_length=None, window_name=None,
                         pad_begin=False, pad_end=False, return_decibel=False, db_amin=1e-5,
                         db_ref_value=1.0, db_dynamic_range=80.0, input_data_format=CH_LAST,
                         output_data_format=CH_LAST) -> np.ndarray:
    s = kapre_stft(x, n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                   input_data_format, output_data_format)
    out = magnitude(s)
    if return_decibel:
        out = magnitude_to_decibel(out, db_ref_value, db_amin, db_dynamic_range)
    return out


# --------------------------------------------------------------------------------------
# consumers / neighbours of the path: Frame, Energy, Delta, LogmelToMFCC
# --------------------------------------------------------------------------------------
def tf_frame(x, frame_length: int, frame_step: int, pad_end: bool = False, pad_value=0.0,
             axis: int = -1) -> np.ndarray:
    """tf.signal.frame(signal, frame_length, frame_step, pad_end, pad_value, axis): the `axis`
    dimension of length T is replaced by (F, frame_length); F = 1 + (T - L)//step without pad_end
    (0 when T < L), ceil(T/step) with pad_end (right-padded with pad_value)."""
    x = np.asarray(x)
    axis = axis % x.ndim
    t = x.shape[axis]
    f = num_frames(t, frame_length, frame_step, pad_end)
    need = (f - 1) * frame_step + frame_length if f > 0 else 0
    xm = np.moveaxis(x, axis, -1)
    if need > t:
        xm = np.pad(xm, [(0, 0)] * (xm.ndim - 1) + [(0, need - t)], constant_values=pad_value)
    idx = np.arange(f)[:, None] * frame_step + np.arange(frame_length)[None, :]
    fr = xm[..., idx]                                           # (..., F, L)
    return np.moveaxis(np.moveaxis(fr, -2, axis), -1, axis + 1)


def kapre_frame(x, frame_length, hop_length, pad_end=False, pad_value=0, data_format=CH_LAST):
    """kapre.Frame.call, signal.py:86-104: (b, t, ch) -> (b, frame, frame_length, ch);
    (b, ch, t) -> (b, ch, frame, frame_length)."""
    axis = 2 if data_format == CH_FIRST else 1
    return tf_frame(np.asarray(x, dtype=np.float64), frame_length, hop_length, pad_end, pad_value, axis)


def kapre_energy(x, sample_rate=22050, ref_duration=0.1, frame_length=2205, hop_length=1102,
                 pad_end=False, pad_value=0, data_format=CH_LAST):
    """kapre.Energy.call, signal.py:187-213: sum of squares per frame, scaled by
    ref_duration / (frame_length / sample_rate).  (b, t, ch) -> (b, frame, ch); (b, ch, t) ->
    (b, ch, frame)."""
    fr = kapre_frame(x, frame_length, hop_length, pad_end, pad_value, data_format)
    frame_axis = 2 if data_format == CH_LAST else 3
    return (ref_duration / (frame_length / sample_rate)) * np.sum(fr * fr, axis=frame_axis)


def kapre_delta(x, win_length=5, mode="symmetric", data_format=CH_LAST):
    """kapre.Delta.call, time_frequency.py:614-635: pad the time axis by n = (win-1)//2 with
    tf.pad(mode), correlate with the kernel [-n .. n] along time, divide by 2*sum(i^2).
    Time axis: 1 for channels_last (b, t, f, ch), 2 for channels_first (b, ch, t, f)."""
    x = np.asarray(x, dtype=np.float64)
    n = (win_length - 1) // 2
    denom = 2 * sum(i * i for i in range(1, n + 1))
    axis = 1 if data_format == CH_LAST else 2
    pad = [(0, 0)] * x.ndim
    pad[axis] = (n, n)
    xp = np.pad(x, pad, mode=mode.lower())
    t = x.shape[axis]
    out = np.zeros_like(x)
    for j in range(-n, n + 1):
        out += j * np.take(xp, np.arange(n + j, n + j + t), axis=axis)
    return out / denom


def mfccs_from_log_mel_spectrograms(log_mel):
    """tf.signal.mfccs_from_log_mel_spectrograms (TensorFlow, published definition):
    dct2 = tf.signal.dct(log_mel, type=2) = 2 * sum_n x[n] cos(pi (2n+1) k / (2N)) along the last
    axis (N = num_mel_bins), result = dct2 * rsqrt(2 N).  (HTK scaling; librosa's 'ortho' DCT
    differs by sqrt(2) in bin 0 -- the check the reference makes at tests/test_signal.py:103-106.)"""
    x = np.asarray(log_mel, dtype=np.float64)
    nm = x.shape[-1]
    n = np.arange(nm)
    basis = 2.0 * np.cos(np.pi * np.outer(2 * n + 1, np.arange(nm)) / (2.0 * nm))     # (n, k)
    return (x @ basis) / math.sqrt(2.0 * nm)


def kapre_logmel_to_mfcc(x, n_mfccs=20, data_format=CH_LAST):
    """kapre.LogmelToMFCC.call, signal.py:418-436: DCT over the mel axis, first n_mfccs bins.
    (b, time, mel, ch) -> (b, time, n_mfccs, ch); (b, ch, time, mel) -> (b, ch, time, n_mfccs)."""
    x = np.asarray(x, dtype=np.float64)
    if data_format == CH_LAST:
        x = np.transpose(x, (0, 1, 3, 2))
    y = mfccs_from_log_mel_spectrograms(x)[..., :n_mfccs]
    if data_format == CH_LAST:
        y = np.transpose(y, (0, 1, 3, 2))
    return y


def mfcc_matrix(n_mels: int, n_mfccs: int) -> np.ndarray:
    """(n_mels, n_mfccs) matrix M with mfcc = log_mel @ M (same numbers as above)."""
    n = np.arange(n_mels)
    return (2.0 * np.cos(np.pi * np.outer(2 * n + 1, np.arange(n_mfccs)) / (2.0 * n_mels))
            / math.sqrt(2.0 * n_mels))
