"""Signal layers: the consumers / neighbours of the time-frequency path that SURVEY 8f row 4 names
(reference: /root/reference/kapre/signal.py).  Frame, Energy and LogmelToMFCC keep the reference's
constructor signatures, validation, get_config() keys and output shapes; the arithmetic runs in
libkapre_hip.so (kpr_frame_f32 / kpr_energy_f32 / kpr_apply_filterbank_f32 with a DCT-II matrix).
MuLawEncoding / MuLawDecoding are out of scope."""
import ctypes
import math

import numpy as np

from . import _ffi, autograd, backend
from .backend import _CH_FIRST_STR, _CH_LAST_STR, _CH_DEFAULT_STR
from .keras_shim import Layer, register_keras_serializable

__all__ = ['Frame', 'Energy', 'LogmelToMFCC']


def _resolve_format(fmt):
    return backend.image_data_format() if fmt == _CH_DEFAULT_STR else fmt


def _waveform_dims(x, data_format):
    if x.dim() != 3:
        raise ValueError('expected a rank-3 waveform batch, got shape %s' % (tuple(x.shape),))
    if data_format == _CH_FIRST_STR:
        b, c, t = x.shape
    else:
        b, t, c = x.shape
    return int(b), int(c), int(t)


@register_keras_serializable(package='Kapre')
class Frame(Layer):
    """Frame the input audio signal -- ``tf.signal.frame`` (reference: signal.py:22-119).

    (batch, time, ch) -> (batch, n_frame, frame_length, ch) for ``channels_last``;
    (batch, ch, time) -> (batch, ch, n_frame, frame_length) for ``channels_first``."""

    def __init__(self, frame_length, hop_length, pad_end=False, pad_value=0, data_format='default',
                 **kwargs):
        super(Frame, self).__init__(**kwargs)
        backend.validate_data_format_str(data_format)
        if frame_length <= 0:
            raise ValueError(f'frame_length must be positive, got: {frame_length}')
        if hop_length <= 0:
            raise ValueError(f'hop_length must be positive, got: {hop_length}')
        if frame_length < hop_length:
            raise ValueError(f'frame_length ({frame_length}) must be >= hop_length ({hop_length})')
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.pad_end = pad_end
        self.pad_value = pad_value
        self.data_format_str = data_format
        self.data_format = _resolve_format(data_format)
        self.time_axis = 2 if self.data_format == _CH_FIRST_STR else 1

    def compute_output_shape(self, input_shape):
        """(b, t, ch) -> (b, frame, frame_length, ch); (b, ch, t) -> (b, ch, frame, frame_length) (tf.signal.frame)"""
        t = input_shape[self.time_axis]
        n = None
        if t is not None:
            t, fl, hop = int(t), int(self.frame_length), int(self.hop_length)
            n = -(-t // hop) if self.pad_end else max(0, 1 + (t - fl) // hop)
        if self.data_format == _CH_FIRST_STR:
            return (input_shape[0], input_shape[1], n, int(self.frame_length))
        return (input_shape[0], n, int(self.frame_length), input_shape[2])

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.frame(self, autograd.prep(x, 'float32'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        x = _ffi.as_device_f32(x)
        b, c, t = _waveform_dims(x, self.data_format)
        L = _ffi.lib()
        f = int(L.kpr_frame_count(t, self.frame_length, self.hop_length, int(bool(self.pad_end))))
        shape = ((b, f, self.frame_length, c) if self.data_format == _CH_LAST_STR
                 else (b, c, f, self.frame_length))
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _ffi.check(L.kpr_frame_f32(_ffi.ptr(x), b, c, t, _ffi.layout(self.data_format),
                                       self.frame_length, self.hop_length, int(bool(self.pad_end)),
                                       float(self.pad_value), _ffi.ptr(out), _ffi.current_stream_ptr()),
                       'kpr_frame_f32')
        return out

    def get_config(self):
        config = super(Frame, self).get_config()
        config.update({'frame_length': self.frame_length, 'hop_length': self.hop_length,
                       'pad_end': self.pad_end, 'pad_value': self.pad_value,
                       'data_format': self.data_format_str})
        return config


@register_keras_serializable(package='Kapre')
class Energy(Layer):
    """Energy of each frame, normalised to ``ref_duration`` (reference: signal.py:122-240).

    (batch, time, ch) -> (batch, n_frame, ch); (batch, ch, time) -> (batch, ch, n_frame).  The
    frames are never materialised: one kernel sums the squares of each frame's samples."""

    def __init__(self, sample_rate=22050, ref_duration=0.1, frame_length=2205, hop_length=1102,
                 pad_end=False, pad_value=0, data_format='default', **kwargs):
        super(Energy, self).__init__(**kwargs)
        backend.validate_data_format_str(data_format)
        self.sample_rate = sample_rate
        self.ref_duration = ref_duration
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.pad_end = pad_end
        self.pad_value = pad_value
        self.data_format_str = data_format
        self.data_format = _resolve_format(data_format)
        self.time_axis = 2 if self.data_format == _CH_FIRST_STR else 1

    def compute_output_shape(self, input_shape):
        """(b, t, ch) -> (b, frame, ch); (b, ch, t) -> (b, ch, frame)"""
        t = input_shape[self.time_axis]
        n = None
        if t is not None:
            t, fl, hop = int(t), int(self.frame_length), int(self.hop_length)
            n = -(-t // hop) if self.pad_end else max(0, 1 + (t - fl) // hop)
        if self.data_format == _CH_FIRST_STR:
            return (input_shape[0], input_shape[1], n)
        return (input_shape[0], n, input_shape[2])

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.energy(self, autograd.prep(x, 'float32'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        x = _ffi.as_device_f32(x)
        b, c, t = _waveform_dims(x, self.data_format)
        L = _ffi.lib()
        f = int(L.kpr_frame_count(t, self.frame_length, self.hop_length, int(bool(self.pad_end))))
        _ffi.check(0 if f >= 0 else -1, 'kpr_frame_count')
        shape = (b, f, c) if self.data_format == _CH_LAST_STR else (b, c, f)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        nor_coeff = self.ref_duration / (self.frame_length / self.sample_rate)
        with torch.cuda.device(x.device):
            _ffi.check(L.kpr_energy_f32(_ffi.ptr(x), b, c, t, _ffi.layout(self.data_format),
                                        self.frame_length, self.hop_length, int(bool(self.pad_end)),
                                        float(self.pad_value), float(nor_coeff), _ffi.ptr(out),
                                        _ffi.current_stream_ptr()), 'kpr_energy_f32')
        return out

    def get_config(self):
        config = super(Energy, self).get_config()
        config.update({'sample_rate': self.sample_rate, 'ref_duration': self.ref_duration,
                       'frame_length': self.frame_length, 'hop_length': self.hop_length,
                       'pad_end': self.pad_end, 'pad_value': self.pad_value,
                       'data_format': self.data_format_str})
        return config


def mfcc_matrix(n_mels: int, n_mfccs: int) -> np.ndarray:
    """(n_mels, n_mfccs) float32 matrix of ``tf.signal.mfccs_from_log_mel_spectrograms``:
    unnormalised DCT-II ``2 cos(pi (2n+1) k / (2N))`` scaled by ``rsqrt(2N)`` (HTK convention; the
    reference notes the sqrt(2) difference to librosa's orthonormal DCT in bin 0, signal.py:370-377).
    Built in float64, stored as floatx."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(min(n_mfccs, n_mels), dtype=np.float64)
    m = 2.0 * np.cos(np.pi * np.outer(2.0 * n + 1.0, k) / (2.0 * n_mels)) / math.sqrt(2.0 * n_mels)
    return m.astype(np.float32)


@register_keras_serializable(package='Kapre')
class LogmelToMFCC(Layer):
    """MFCC from a log-melspectrogram (reference: signal.py:364-447): DCT-II over the mel axis,
    first ``n_mfccs`` coefficients.  (b, time, mel, ch) -> (b, time, n_mfccs, ch) or
    (b, ch, time, mel) -> (b, ch, time, n_mfccs).  The DCT is a (n_mels x n_mfccs) matrix applied
    by the same MFMA GEMM kernel as ApplyFilterbank."""

    def __init__(self, n_mfccs=20, data_format='default', **kwargs):
        super(LogmelToMFCC, self).__init__(**kwargs)
        backend.validate_data_format_str(data_format)
        self.n_mfccs = n_mfccs
        self.data_format_str = data_format
        self.data_format = _resolve_format(data_format)
        self.permutation = (0, 1, 3, 2) if self.data_format == _CH_LAST_STR else None
        self._mats = {}

    def _matrix(self, n_mels, device):
        key = (n_mels, str(device))
        if key not in self._mats:
            import torch
            self._mats[key] = torch.from_numpy(mfcc_matrix(n_mels, self.n_mfccs)).to(device)
        return self._mats[key]

    def compute_output_shape(self, input_shape):
        """the mel axis (3 for channels_first, 2 for channels_last) becomes n_mfccs"""
        shape = list(input_shape)
        shape[3 if self.data_format == _CH_FIRST_STR else 2] = int(self.n_mfccs)
        return tuple(shape)

    def call(self, log_melgrams):
        if autograd.needs_grad(log_melgrams):
            x = autograd.prep(log_melgrams, 'float32')
            n_mels = int(x.shape[2] if self.data_format == _CH_LAST_STR else x.shape[3]) if x.dim() == 4 else 0
            if n_mels:
                mat_t = self._matrix(n_mels, x.device).t().contiguous()      # (n_mfccs, n_mels): the backward GEMM
                return autograd.matrix(self, x, mat_t, self.data_format)
        return self._forward(log_melgrams)

    def _forward(self, log_melgrams):
        import torch

        x = _ffi.as_device_f32(log_melgrams)
        if x.dim() != 4:
            raise ValueError('LogmelToMFCC expects a rank-4 input, got shape %s' % (tuple(x.shape),))
        if self.data_format == _CH_LAST_STR:
            b, f, m, c = x.shape
        else:
            b, c, f, m = x.shape
        mat = self._matrix(int(m), x.device)
        n_out = int(mat.shape[1])                      # min(n_mfccs, n_mels), as the slice upstream
        shape = (b, f, n_out, c) if self.data_format == _CH_LAST_STR else (b, c, f, n_out)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().kpr_apply_filterbank_f32(
                _ffi.ptr(x), b, c, f, m, _ffi.layout(self.data_format), _ffi.ptr(mat), n_out,
                ctypes.c_void_p(0), _ffi.ptr(out), _ffi.current_stream_ptr()),
                'kpr_apply_filterbank_f32 (mfcc)')
        return out

    def get_config(self):
        config = super(LogmelToMFCC, self).get_config()
        config.update({'n_mfccs': self.n_mfccs, 'data_format': self.data_format_str})
        return config
