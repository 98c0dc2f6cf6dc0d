"""The handful of librosa calls the reference's tests compare Kapre against, restated on top of the
float64 oracle (oracle/kapre_oracle.py).  TEST INFRASTRUCTURE ONLY.

librosa is not installable in this image.  Each function names the librosa semantics it follows and the
oracle functions it is made of; the oracle's own anchoring to published librosa / TensorFlow numbers is in
tests/test_oracle_published.py.  Upstream only ever calls these with win_length == n_fft (where librosa's
centred window padding and tf.signal's right padding coincide) and Hann / Hamming windows of even length
(where scipy's periodic window, which librosa uses, equals tf.signal's)."""
import numpy as np
import scipy.signal

import kapre_oracle as o


def stft(y, n_fft=2048, hop_length=None, win_length=None, window='hann', center=True, pad_mode='constant'):
    """librosa.stft: (1 + n_fft/2, n_frames) complex; hop defaults to win_length // 4; periodic window
    (scipy get_window(fftbins=True)); center=True pads n_fft // 2 on both sides (librosa >= 0.10: zeros)."""
    win_length = win_length or n_fft
    hop_length = hop_length or win_length // 4
    assert win_length == n_fft, "upstream never uses win_length != n_fft against librosa"
    w = scipy.signal.get_window(window, win_length, fftbins=True)
    y = np.asarray(y, np.float64)
    if center:
        y = np.pad(y, n_fft // 2, mode=pad_mode)
    return o.tf_stft(y, win_length, hop_length, n_fft, w, False).T


def magphase(d):
    mag = np.abs(d)
    with np.errstate(invalid='ignore', divide='ignore'):
        ph = np.where(mag > 0, d / np.where(mag > 0, mag, 1.0), 1.0 + 0j)
    return mag.astype(np.complex128), ph


def melspectrogram(y, sr, n_fft, hop_length, win_length, center, power, n_mels, fmin, fmax):
    """librosa.feature.melspectrogram: mel_basis (n_mels, 1 + n_fft/2) @ |stft| ** power."""
    s = np.abs(stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center)) ** power
    fb = o.filterbank_mel(sr, n_fft // 2 + 1, n_mels, fmin, fmax).astype(np.float64)      # (K, M), Kapre's transpose
    return fb.T @ s


def power_to_db(s, ref=1.0, amin=1e-10, top_db=80.0):
    """librosa.power_to_db: 10 log10(max(amin, S)) - 10 log10(max(amin, ref)), floored at max - top_db
    (the max over the WHOLE array: upstream applies it per batch item / per row)."""
    s = np.asarray(s, np.float64)
    db = 10.0 * np.log10(np.maximum(amin, s)) - 10.0 * np.log10(np.maximum(amin, ref))
    if top_db is not None:
        db = np.maximum(db, db.max() - top_db)
    return db
