"""Minimal numpy stand-in for the handful of TensorFlow symbols that Kapre's hot path touches.
Test infrastructure only -- see ../README.md.  Everything computes in float64/complex128."""
import os
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, "..", "..")))
import kapre_oracle as _o  # noqa: E402

Tensor = np.ndarray
float16, float32, float64 = np.float16, np.float32, np.float64
int32, int64 = np.int32, np.int64
complex64, complex128 = np.complex64, np.complex128


def function(fn=None, **_kw):
    if fn is None:
        return lambda f: f
    return fn


def constant(value, dtype=None):
    return np.asarray(value, dtype=dtype)


def convert_to_tensor(value, dtype=None):
    return np.asarray(value, dtype=dtype)


def Variable(value):
    return np.asarray(value)


def cast(x, dtype):
    if isinstance(dtype, str):
        dtype = np.dtype(dtype)
    return np.asarray(x).astype(dtype)


def transpose(x, perm=None):
    return np.transpose(x, perm)


def pad(x, paddings, mode="CONSTANT"):
    mode = {"CONSTANT": "constant", "SYMMETRIC": "symmetric", "REFLECT": "reflect"}[mode.upper()]
    return np.pad(x, [tuple(int(v) for v in p) for p in np.asarray(paddings)], mode=mode)


def abs(x):  # noqa: A001
    return np.abs(x)


def tensordot(a, b, axes):
    return np.tensordot(a, b, axes=axes)


def shape(x):
    return np.asarray(np.shape(x))


def print(*args, **kwargs):  # noqa: A001  (kapre.Energy.call logs shapes with tf.print)
    return None


def reshape(x, shp):
    return np.reshape(x, shp)


# ---- primitive ops used by the reference's own restatement of tf.signal.stft
# (kapre/tflite_compatible_stft.py: matmul-DFT + gather framing).  Plain numpy, no oracle code:
# oracle/make_golden_l0.py runs that file on these to pin framing / right-padding / rDFT.
def matmul(a, b):
    return np.matmul(a, b)


def stack(values, axis=0):
    return np.stack(values, axis=axis)


def concat(values, axis=0):
    return np.concatenate([np.asarray(v) for v in values], axis=axis)


def zeros(shp, dtype=np.float32):
    return np.zeros([int(v) for v in np.atleast_1d(shp)], dtype=dtype)


def rank(x):
    return np.ndim(x)


def slice(x, begin, size):  # noqa: A001
    idx = tuple(np.s_[int(b):int(b) + int(n)] for b, n in zip(begin, size))
    return np.asarray(x)[idx]


def gather(params, indices, axis=0):
    return np.take(params, indices, axis=axis)


def where(cond, a, b):
    return np.where(cond, a, b)


def logical_and(a, b):
    return np.logical_and(a, b)


def equal(a, b):
    return np.equal(a, b)


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001
    return np.arange(start, limit, delta, dtype=dtype)


class _Math(types.ModuleType):
    log = staticmethod(np.log)
    maximum = staticmethod(np.maximum)
    angle = staticmethod(np.angle)
    real = staticmethod(np.real)
    imag = staticmethod(np.imag)

    @staticmethod
    def reduce_max(x, axis=None, keepdims=False):
        return np.max(x, axis=axis, keepdims=keepdims)

    @staticmethod
    def reduce_sum(x, axis=None, keepdims=False):
        return np.sum(x, axis=axis, keepdims=keepdims)

    square = staticmethod(np.square)


math = _Math("tensorflow.math")


class _Signal(types.ModuleType):
    @staticmethod
    def hann_window(n, periodic=True, dtype=None):
        return _o.hann_window(int(n))

    @staticmethod
    def hamming_window(n, periodic=True, dtype=None):
        return _o.hamming_window(int(n))

    @staticmethod
    def kaiser_window(n, beta=12.0, dtype=None):
        return _o.kaiser_window(int(n), beta)

    @staticmethod
    def kaiser_bessel_derived_window(n, beta=12.0, dtype=None):
        return _o.kaiser_bessel_derived_window(int(n), beta)

    @staticmethod
    def vorbis_window(n, dtype=None):
        return _o.vorbis_window(int(n))

    @staticmethod
    def stft(signals, frame_length, frame_step, fft_length=None, window_fn=None, pad_end=False,
             name=None):
        w = window_fn(frame_length, dtype=np.float64) if window_fn else np.ones(frame_length)
        return _o.tf_stft(signals, int(frame_length), int(frame_step), int(fft_length), w,
                          bool(pad_end))

    @staticmethod
    def frame(signal, frame_length, frame_step, pad_end=False, pad_value=0, axis=-1, name=None):
        return _o.tf_frame(signal, int(frame_length), int(frame_step), bool(pad_end), pad_value, int(axis))

    @staticmethod
    def mfccs_from_log_mel_spectrograms(log_mel_spectrograms, name=None):
        return _o.mfccs_from_log_mel_spectrograms(log_mel_spectrograms)

    @staticmethod
    def inverse_stft_window_fn(frame_step, forward_window_fn=None, name=None):
        def _fn(frame_length, dtype=None):
            fw = forward_window_fn(frame_length, dtype=dtype)
            return _o.inverse_stft_window(int(frame_length), int(frame_step), fw)
        return _fn

    @staticmethod
    def inverse_stft(stfts, frame_length, frame_step, fft_length=None, window_fn=None, name=None):
        w = window_fn(frame_length, dtype=np.float64) if window_fn else np.ones(frame_length)
        return _o.tf_inverse_stft(stfts, int(frame_length), int(frame_step), int(fft_length), w)


signal = _Signal("tensorflow.signal")

from . import keras  # noqa: E402,F401
