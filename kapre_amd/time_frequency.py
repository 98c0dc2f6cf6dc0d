"""Time-frequency layers -- MI355X implementation behind Kapre's own layer API.

Mirror of /root/reference/kapre/time_frequency.py for the hot path: ``STFT`` (:61-203),
``InverseSTFT`` (:207-333), ``Magnitude`` (:337-359), ``Phase`` (:363-411),
``MagnitudeToDecibel`` (:415-465), ``ApplyFilterbank`` (:469-559).  Constructor signatures,
defaults, ``get_config`` keys and raised exception types are the reference's; ``call`` runs the
hand-written gfx950 kernels of libkapre_hip.so through ``kapre_amd._ffi`` (ctypes).  Inputs may be
numpy arrays or torch tensors; outputs are torch tensors on the GPU (complex64 / float32).

``fuse_and_run`` is the peephole optimiser used by ``Sequential``: the chains
``STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel]``, ``STFT -> Magnitude
[-> MagnitudeToDecibel]`` and ``STFT -> Phase`` each become one kernel launch (plus the clamp
pass when decibel scaling is on).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi, autograd, backend
from .backend import _CH_FIRST_STR, _CH_LAST_STR, _CH_DEFAULT_STR
from .keras_shim import Layer, register_keras_serializable

__all__ = [
    'STFT',
    'InverseSTFT',
    'Magnitude',
    'Phase',
    'MagnitudeToDecibel',
    'ApplyFilterbank',
    'Delta',
]


def _resolve_format(fmt):
    return backend.image_data_format() if fmt == _CH_DEFAULT_STR else fmt


def _forget_packed(ptr):
    try:
        _ffi.lib().kpr_filterbank_forget(ctypes.c_void_p(ptr))
    except Exception:        # interpreter shutdown: the library may be gone already
        pass


class _DeviceConstants:
    """Per-device cache of small constant tensors (windows, filterbanks)."""

    def __init__(self):
        self._cache = {}

    def get(self, key, device, make_numpy):
        import torch

        k = (key, str(device))
        t = self._cache.get(k)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(make_numpy())).to(device)
            self._cache[k] = t
        return t


def _workspace(nbytes: int, device):
    import torch

    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class _WorkspacePool:
    """Scratch buffers of the cached (plan-based) calls: one per (device, stream), grown
    geometrically and NEVER freed -- a captured hipGraph keeps replaying the pointer it recorded, so
    a buffer that was ever handed out stays alive (outgrown ones are parked in ``_retired``; total
    memory stays below twice the largest request).  Calls on one stream are ordered, so sharing the
    buffer between plans of that stream is safe."""

    def __init__(self):
        self._live = {}
        self._retired = []

    def get(self, nbytes: int, device, stream_ptr: int):
        key = (str(device), int(stream_ptr))
        buf = self._live.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                self._retired.append(buf)
            grow = 0 if buf is None else 2 * buf.numel()
            buf = self._live[key] = _workspace(max(int(nbytes), grow, 4096), device)
        return buf


_PLAN_WORKSPACES = _WorkspacePool()


@register_keras_serializable(package='Kapre')
class STFT(Layer):
    """Short-time Fourier transform layer (reference: time_frequency.py:61-203).

    ``output_data_format == 'channels_last'`` -> (batch, time, freq, channel);
    ``'channels_first'`` -> (batch, channel, time, freq); dtype complex64.

    Args are the reference's: ``n_fft=2048, win_length=None (-> n_fft), hop_length=None
    (-> win_length // 4), window_name=None (-> hann), pad_begin=False, pad_end=False,
    input_data_format='default', output_data_format='default', **kwargs``.
    ``pad_begin`` pads ``n_fft - hop_length`` zeros on the left (code at :169-172).
    """

    def __init__(
        self,
        n_fft=2048,
        win_length=None,
        hop_length=None,
        window_name=None,
        pad_begin=False,
        pad_end=False,
        input_data_format='default',
        output_data_format='default',
        **kwargs,
    ):
        super(STFT, self).__init__(**kwargs)

        for data_format in (input_data_format, output_data_format):
            backend.validate_data_format_str(data_format)   # reference order: validate first (:115-116)
        if isinstance(input_data_format, dict):
            input_data_format = input_data_format['config']
        if isinstance(output_data_format, dict):
            output_data_format = output_data_format['config']

        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = win_length // 4

        self.n_fft = n_fft
        self.win_length = win_length
        self.hop_length = hop_length
        self.window_name = window_name
        self.window_fn = backend.get_window_fn(window_name)   # NotImplementedError if unknown
        self.pad_begin = pad_begin
        self.pad_end = pad_end

        self.input_data_format_original = input_data_format
        self.output_data_format_original = output_data_format
        self.output_data_format = _resolve_format(output_data_format)
        self.input_data_format = _resolve_format(input_data_format)
        self._consts = _DeviceConstants()

    # -- helpers -----------------------------------------------------------------------------
    def _geom(self, x) -> _ffi.StftGeom:
        if x.dim() != 3:
            raise ValueError('STFT expects a rank-3 input (batch, time, ch) / (batch, ch, time), '
                             'got shape %s' % (tuple(x.shape),))
        if self.input_data_format == _CH_LAST_STR:
            b, t, c = x.shape
        else:
            b, c, t = x.shape
        return _ffi.StftGeom(b, c, t, int(self.n_fft), int(self.win_length), int(self.hop_length),
                             int(bool(self.pad_begin)), int(bool(self.pad_end)),
                             _ffi.layout(self.input_data_format),
                             _ffi.layout(self.output_data_format))

    def _window(self, device):
        return self._consts.get(('window', int(self.win_length), id(self.window_fn)), device,
                                lambda: self.window_fn(int(self.win_length)))

    def _plan_version(self):
        """Everything of the layer a cached fused-call plan depends on (attributes are plain and
        may be reassigned by the caller, as with any Keras layer)."""
        return (int(self.n_fft), int(self.win_length), int(self.hop_length), bool(self.pad_begin),
                bool(self.pad_end), self.input_data_format, self.output_data_format,
                id(self.window_fn))

    def _out_shape(self, g: _ffi.StftGeom, n_frames: int, q: int):
        if self.output_data_format == _CH_LAST_STR:
            return (g.batch, n_frames, q, g.channels)
        return (g.batch, g.channels, n_frames, q)

    def _run(self, x, mode: int):
        import torch

        if self._f64:
            return self._run_f64(x, mode)
        x = _ffi.as_device_f32(x)
        L = _ffi.lib()
        g = self._geom(x)
        n_frames = int(L.kpr_num_frames(ctypes.byref(g)))
        if n_frames < 0:
            _ffi.check(-1, 'kpr_num_frames')
        k = int(self.n_fft) // 2 + 1
        dtype = torch.complex64 if mode == _ffi.OUT_COMPLEX else torch.float32
        out = torch.empty(self._out_shape(g, n_frames, k), dtype=dtype, device=x.device)
        with torch.cuda.device(x.device):
            ws_bytes = int(L.kpr_stft_workspace_bytes(ctypes.byref(g), mode))
            ws = _workspace(ws_bytes, x.device)
            _ffi.check(L.kpr_stft_f32(_ffi.ptr(x), ctypes.byref(g), _ffi.ptr(self._window(x.device)),
                                      _ffi.ptr(out), mode, _ffi.ptr(ws), ws_bytes,
                                      _ffi.current_stream_ptr()), 'kpr_stft_f32')
        return out

    def _run_f64(self, x, mode: int):
        """dtype='float64' layer: float64 in, complex128 / float64 out (reference :155)."""
        import torch

        x = _ffi.as_device_dtype(x, torch.float64)
        L = _ffi.lib()
        g = self._geom(x)
        n_frames = int(L.kpr_num_frames(ctypes.byref(g)))
        if n_frames < 0:
            _ffi.check(-1, 'kpr_num_frames')
        k = int(self.n_fft) // 2 + 1
        dtype = torch.complex128 if mode == _ffi.OUT_COMPLEX else torch.float64
        out = torch.empty(self._out_shape(g, n_frames, k), dtype=dtype, device=x.device)
        win = self._consts.get(('window64', int(self.win_length), id(self.window_fn)), x.device,
                               lambda: backend.window_values(self.window_fn, int(self.win_length), np.float64))
        with torch.cuda.device(x.device):
            _ffi.check(L.kpr_stft_f64(_ffi.ptr(x), ctypes.byref(g), _ffi.ptr(win), _ffi.ptr(out), mode,
                                      _ffi.current_stream_ptr()), 'kpr_stft_f64')
        return out

    def compute_output_shape(self, input_shape):
        """(b, t, ch) / (b, ch, t) -> (b, frame, n_fft // 2 + 1, ch) / (b, ch, frame, n_fft // 2 + 1); frames as tf.signal.stft counts
        them after the left padding of n_fft - hop_length (reference: time_frequency.py:164-185)"""
        b, t, c = (input_shape[0], input_shape[1], input_shape[2]) if self.input_data_format == _CH_LAST_STR else \
                  (input_shape[0], input_shape[2], input_shape[1])
        frames = None
        if t is not None:
            t = int(t) + (int(self.n_fft) - int(self.hop_length) if self.pad_begin else 0)
            frames = -(-t // int(self.hop_length)) if self.pad_end else max(0, 1 + (t - int(self.win_length)) // int(self.hop_length))
        k = int(self.n_fft) // 2 + 1
        return (b, frames, k, c) if self.output_data_format == _CH_LAST_STR else (b, c, frames, k)

    def call(self, x):
        """(batch, time, ch) or (batch, ch, time) float -> complex64 STFT (reference :146-187).
        Differentiable when ``x`` is a torch tensor that requires grad (kapre_amd/autograd.py)."""
        if autograd.needs_grad(x):
            return autograd.stft(self, autograd.prep(x, 'float64' if self._f64 else 'float32'))
        return self._run(x, _ffi.OUT_COMPLEX)

    def get_config(self):
        config = super(STFT, self).get_config()
        config.update(
            {
                'n_fft': self.n_fft,
                'win_length': self.win_length,
                'hop_length': self.hop_length,
                'window_name': self.window_name,
                'pad_begin': self.pad_begin,
                'pad_end': self.pad_end,
                'input_data_format': self.input_data_format_original,
                'output_data_format': self.output_data_format_original,
            }
        )
        return config


def _complex_f64(x, layer_f64: bool) -> bool:
    """Compute precision for a layer that consumes a complex tensor: the INPUT's.  Keras autocasting only touches
    floating-point tensors, so the reference's ``tf.abs`` / ``tf.math.angle`` / ``tf.signal.inverse_stft`` run in the
    precision of the complex input whatever the layer's own dtype is (complex128 -> float64 out, complex64 -> float32
    out).  Non-complex inputs follow the layer dtype."""
    name = str(getattr(x, 'dtype', '')).replace('torch.', '')
    if name == 'complex128':
        return True
    if name == 'complex64':
        return False
    return layer_f64


@register_keras_serializable(package='Kapre')
class InverseSTFT(Layer):
    """Inverse STFT layer (reference: time_frequency.py:207-333).

    Input (batch, time, freq, ch) / (batch, ch, time, freq) complex64; output
    (batch, time, ch) / (batch, ch, time) float32 of length ``(n_frames-1)*hop + win_length``
    (the caller trims, as the reference notes at :213-214).
    """

    def __init__(
        self,
        n_fft=2048,
        win_length=None,
        hop_length=None,
        forward_window_name=None,
        input_data_format='default',
        output_data_format='default',
        **kwargs,
    ):
        super(InverseSTFT, self).__init__(**kwargs)

        for data_format in (input_data_format, output_data_format):
            backend.validate_data_format_str(data_format)
        if isinstance(input_data_format, dict):
            input_data_format = input_data_format['config']
        if isinstance(output_data_format, dict):
            output_data_format = output_data_format['config']

        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = win_length // 4

        self.n_fft = n_fft
        self.win_length = win_length
        self.hop_length = hop_length
        self.forward_window_name = forward_window_name
        self.window_fn = backend.inverse_stft_window_fn(
            frame_step=hop_length, forward_window_fn=backend.get_window_fn(forward_window_name)
        )

        self.input_data_format_original = input_data_format
        self.output_data_format_original = output_data_format
        self.output_data_format = _resolve_format(output_data_format)
        self.input_data_format = _resolve_format(input_data_format)
        self._consts = _DeviceConstants()

    def compute_output_shape(self, input_shape):
        """(b, frame, freq, ch) / (b, ch, frame, freq) -> (b, (frame - 1) hop + win, ch) / (b, ch, ...): untrimmed, as upstream"""
        b, f, c = (input_shape[0], input_shape[1], input_shape[3]) if self.input_data_format == _CH_LAST_STR else \
                  (input_shape[0], input_shape[2], input_shape[1])
        t = None if f is None else ((int(f) - 1) * int(self.hop_length) + int(self.win_length) if int(f) > 0 else 0)
        return (b, t, c) if self.output_data_format == _CH_LAST_STR else (b, c, t)

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.istft(self, autograd.prep(x, 'complex128' if _complex_f64(x, self._f64) else 'complex64'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        f64 = _complex_f64(x, self._f64)
        x = _ffi.as_device_dtype(x, torch.complex128) if f64 else _ffi.as_device_c64(x)
        if x.dim() != 4:
            raise ValueError('InverseSTFT expects a rank-4 input, got shape %s' % (tuple(x.shape),))
        if self.input_data_format == _CH_LAST_STR:
            b, f, k, c = x.shape
        else:
            b, c, f, k = x.shape
        k_need = int(self.n_fft) // 2 + 1
        if k != k_need:
            # tf.signal.irfft crops / zero-pads the frequency axis to n_fft//2+1
            axis = 2 if self.input_data_format == _CH_LAST_STR else 3
            if k > k_need:
                x = x.narrow(axis, 0, k_need).contiguous()
            else:
                pad = [0, 0] * (x.dim() - 1 - axis) + [0, k_need - k]
                x = torch.nn.functional.pad(torch.view_as_real(x), [0, 0] + pad)
                x = torch.view_as_complex(x.contiguous())
        # StftGeom: in_layout = waveform layout, out_layout = spectrogram layout
        g = _ffi.StftGeom(b, c, 0, int(self.n_fft), int(self.win_length), int(self.hop_length), 0, 0,
                          _ffi.layout(self.output_data_format), _ffi.layout(self.input_data_format))
        t_out = (f - 1) * int(self.hop_length) + int(self.win_length) if f > 0 else 0
        shape = (b, t_out, c) if self.output_data_format == _CH_LAST_STR else (b, c, t_out)
        L = _ffi.lib()
        if f64:
            out = torch.empty(shape, dtype=torch.float64, device=x.device)
            win = self._consts.get('synth64', x.device,
                                   lambda: backend.window_values(self.window_fn, int(self.win_length), np.float64))
            with torch.cuda.device(x.device):
                ws_bytes = int(L.kpr_istft_f64_workspace_bytes(ctypes.byref(g), f))
                ws = _workspace(ws_bytes, x.device)
                _ffi.check(L.kpr_istft_f64(_ffi.ptr(x), ctypes.byref(g), f, _ffi.ptr(win), _ffi.ptr(out),
                                           _ffi.ptr(ws), ws_bytes, _ffi.current_stream_ptr()),
                           'kpr_istft_f64')
            return out
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        win = self._consts.get('synth', x.device, lambda: self.window_fn(int(self.win_length)))
        with torch.cuda.device(x.device):
            ws_bytes = int(L.kpr_istft_workspace_bytes(ctypes.byref(g), f))
            ws = _workspace(ws_bytes, x.device)
            _ffi.check(L.kpr_istft_f32(_ffi.ptr(x), ctypes.byref(g), f, _ffi.ptr(win), _ffi.ptr(out),
                                       _ffi.ptr(ws), ws_bytes, _ffi.current_stream_ptr()),
                       'kpr_istft_f32')
        return out

    def get_config(self):
        config = super(InverseSTFT, self).get_config()
        config.update(
            {
                'n_fft': self.n_fft,
                'win_length': self.win_length,
                'hop_length': self.hop_length,
                'forward_window_name': self.forward_window_name,
                'input_data_format': self.input_data_format_original,
                'output_data_format': self.output_data_format_original,
            }
        )
        return config


@register_keras_serializable(package='Kapre')
class Magnitude(Layer):
    """Magnitude of a complex input -> float32 (reference: time_frequency.py:337-359)."""

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.magnitude(self, autograd.prep(x, 'complex128' if _complex_f64(x, self._f64) else 'complex64'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        if _complex_f64(x, self._f64):
            x = _ffi.as_device_dtype(x, torch.complex128)
            out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
            with torch.cuda.device(x.device):
                _ffi.check(_ffi.lib().kpr_abs_c128(_ffi.ptr(x), x.numel(), _ffi.ptr(out),
                           _ffi.current_stream_ptr()), 'kpr_abs_c128')
            return out
        x = _ffi.as_device_c64(x)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().kpr_abs_c64(_ffi.ptr(x), x.numel(), _ffi.ptr(out),
                                              _ffi.current_stream_ptr()), 'kpr_abs_c64')
        return out


@register_keras_serializable(package='Kapre')
class Phase(Layer):
    """Phase (radian) of a complex input (reference: time_frequency.py:363-411).

    ``approx_atan_accuracy`` is kept for config compatibility; the TFLite continued-fraction
    approximation it selects upstream is a TFLite deployment feature and is not reproduced:
    the accurate ``atan2`` is always used.
    """

    def __init__(self, approx_atan_accuracy=None, **kwargs):
        super(Phase, self).__init__(**kwargs)
        self.approx_atan_accuracy = approx_atan_accuracy

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.phase(self, autograd.prep(x, 'complex128' if _complex_f64(x, self._f64) else 'complex64'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        if _complex_f64(x, self._f64):
            x = _ffi.as_device_dtype(x, torch.complex128)
            out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
            with torch.cuda.device(x.device):
                _ffi.check(_ffi.lib().kpr_angle_c128(_ffi.ptr(x), x.numel(), _ffi.ptr(out),
                           _ffi.current_stream_ptr()), 'kpr_angle_c128')
            return out
        x = _ffi.as_device_c64(x)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().kpr_angle_c64(_ffi.ptr(x), x.numel(), _ffi.ptr(out),
                                                _ffi.current_stream_ptr()), 'kpr_angle_c64')
        return out

    def get_config(self):
        config = super(Phase, self).get_config()
        config.update({'approx_atan_accuracy': self.approx_atan_accuracy})
        return config


@register_keras_serializable(package='Kapre')
class MagnitudeToDecibel(Layer):
    """Decibel scaling layer wrapping ``backend.magnitude_to_decibel``
    (reference: time_frequency.py:415-465)."""

    def __init__(self, ref_value=1.0, amin=1e-5, dynamic_range=80.0, **kwargs):
        super(MagnitudeToDecibel, self).__init__(**kwargs)
        self.ref_value = ref_value
        self.amin = amin
        self.dynamic_range = dynamic_range

    def call(self, x):
        if autograd.needs_grad(x):
            self._db_params()          # parameter validation before anything is recorded
            return autograd.decibel(self, autograd.prep(x, 'float64' if self._f64 else 'float32'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        # Keras autocast: the layer dtype decides the compute dtype, then the backend follows its input
        x = _ffi.as_device_dtype(x, torch.float64) if self._f64 else _ffi.as_device_f32(x)
        return backend.magnitude_to_decibel(
            x, ref_value=self.ref_value, amin=self.amin, dynamic_range=self.dynamic_range
        )

    def _db_params(self) -> _ffi.DbParams:
        # same validation (and order) as backend.magnitude_to_decibel, backend.py:168-173
        if self.ref_value <= 0:
            raise ValueError(f'ref_value must be positive, got: {self.ref_value}')
        if self.amin <= 0:
            raise ValueError(f'amin must be positive, got: {self.amin}')
        if self.dynamic_range <= 0:
            raise ValueError(f'dynamic_range must be positive, got: {self.dynamic_range}')
        return _ffi.DbParams(1, float(self.ref_value), float(self.amin), float(self.dynamic_range))

    def get_config(self):
        config = super(MagnitudeToDecibel, self).get_config()
        config.update(
            {
                'amin': self.amin,
                'dynamic_range': self.dynamic_range,
                'ref_value': self.ref_value,
            }
        )
        return config


@register_keras_serializable(package='Kapre')
class ApplyFilterbank(Layer):
    """Apply a (n_freq, n_filterbanks) filterbank along the frequency axis
    (reference: time_frequency.py:469-559).

    ``type`` is ``'mel'`` or ``'log'``; ``filterbank_kwargs`` go to ``backend.filterbank_mel`` /
    ``backend.filterbank_log``.  As upstream, any other ``type`` leaves ``self.filterbank``
    unset (AttributeError on use).
    """

    def __init__(
        self,
        type,
        filterbank_kwargs,
        data_format='default',
        **kwargs,
    ):
        super(ApplyFilterbank, self).__init__(**kwargs)
        backend.validate_data_format_str(data_format)
        if isinstance(data_format, dict):
            data_format = data_format['config']

        self.type = type
        self.filterbank_kwargs = filterbank_kwargs
        self._consts = _DeviceConstants()
        self._kranges = None
        self._fb_version = 0

        if type == 'log':
            self.filterbank = _log_filterbank = backend.filterbank_log(**filterbank_kwargs)
        elif type == 'mel':
            self.filterbank = _mel_filterbank = backend.filterbank_mel(**filterbank_kwargs)

        self.data_format_original = data_format
        self.data_format = _resolve_format(data_format)

        if self.data_format == _CH_FIRST_STR:
            self.freq_axis = 3
        else:
            self.freq_axis = 2

    @property
    def filterbank(self):
        """The (n_freq, n_filterbanks) float32 matrix.  ASSIGNING a new array (what
        ``dist.broadcast_constants`` does) bumps ``_fb_version`` and drops every device copy,
        packed copy and k-range table derived from the old one, here and in the fused-call plans
        (they are keyed on the version).  Mutating the array in place is not tracked.  As upstream,
        a ``type`` other than 'mel' / 'log' leaves the attribute unset (AttributeError on use)."""
        try:
            return self.__dict__['_filterbank']
        except KeyError:
            raise AttributeError("'ApplyFilterbank' object has no attribute 'filterbank'") from None

    @filterbank.setter
    def filterbank(self, value):
        self.__dict__['_filterbank'] = value
        self._fb_version += 1
        self._kranges = None
        self._consts._cache.clear()

    def _fb_device(self, device):
        return self._consts.get('fb', device, lambda: np.asarray(self.filterbank, np.float32))

    def _fb_packed_device(self, device):
        """Filterbank in MFMA-fragment order (kpr_filterbank_pack), built once per device.  When the tensor is released
        (layer dropped, filterbank replaced) the library is told to forget the address: its header check is cached per
        address, and the allocator hands freed addresses out again."""
        fresh = []
        t = self._consts.get('fb_packed', device, lambda: fresh.append(1) or _ffi.filterbank_pack(
            np.asarray(self.filterbank, np.float32), self._fb_kranges()))
        if fresh:
            import weakref
            weakref.finalize(t, _forget_packed, t.data_ptr())
        return t

    def _fb_kranges(self):
        """Host int32 [lo, hi) row ranges per 16-filter tile (exact zeros outside)."""
        if self._kranges is None:
            self._kranges = _ffi.filterbank_kranges(np.asarray(self.filterbank, np.float32))
        return self._kranges

    def _fb_transposed_device(self, device, f64: bool):
        """(n_filt, n_freq) copy of the matrix: the backward pass is the same GEMM with it."""
        return self._consts.get('fbT64' if f64 else 'fbT', device,
                                lambda: np.ascontiguousarray(
                                    np.asarray(self.filterbank, np.float64 if f64 else np.float32).T))

    def compute_output_shape(self, input_shape):
        """the frequency axis (3 for channels_first, 2 for channels_last) becomes the number of filters"""
        shape = list(input_shape)
        shape[self.freq_axis] = int(self.filterbank.shape[1])
        return tuple(shape)

    def call(self, x):
        if autograd.needs_grad(x):
            x = autograd.prep(x, 'float64' if self._f64 else 'float32')
            return autograd.matrix(self, x, self._fb_transposed_device(x.device, self._f64), self.data_format)
        return self._forward(x)

    def _forward(self, x):
        import torch

        f64 = self._f64
        x = _ffi.as_device_dtype(x, torch.float64) if f64 else _ffi.as_device_f32(x)
        if x.dim() != 4:
            raise ValueError('ApplyFilterbank expects a rank-4 input, got shape %s'
                             % (tuple(x.shape),))
        if self.data_format == _CH_LAST_STR:
            b, f, k, c = x.shape
        else:
            b, c, f, k = x.shape
        n_freq, n_filt = self.filterbank.shape
        if k != n_freq:
            raise ValueError('frequency axis has %d bins but the filterbank expects %d'
                             % (k, n_freq))
        shape = (b, f, n_filt, c) if self.data_format == _CH_LAST_STR else (b, c, f, n_filt)
        if f64:
            # the filterbank itself is floatx (float32) upstream as well (backend.py:231, :296); cast like TF does
            out = torch.empty(shape, dtype=torch.float64, device=x.device)
            fb64 = self._consts.get('fb64', x.device, lambda: np.asarray(self.filterbank, np.float64))
            with torch.cuda.device(x.device):
                _ffi.check(_ffi.lib().kpr_apply_filterbank_f64(
                    _ffi.ptr(x), b, c, f, n_freq, _ffi.layout(self.data_format), _ffi.ptr(fb64), n_filt,
                    _ffi.ptr(out), _ffi.current_stream_ptr()), 'kpr_apply_filterbank_f64')
            return out
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        kr = self._fb_kranges()
        packed = None
        thin = n_filt <= 64 and n_freq <= 512 and n_freq % 4 == 0          # the thin GEMM takes these
        if not thin and n_freq <= 1025:
            try:
                packed = self._fb_packed_device(x.device)      # wide banded matrix: MFMA consumer path
            except RuntimeError:
                packed = None                                   # band too wide to pack: generic GEMM
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().kpr_apply_filterbank_packed_f32(
                _ffi.ptr(x), b, c, f, n_freq, _ffi.layout(self.data_format),
                _ffi.ptr(self._fb_device(x.device)), _ffi.ptr(packed), n_filt,
                kr.ctypes.data_as(ctypes.c_void_p), _ffi.ptr(out), _ffi.current_stream_ptr()),
                'kpr_apply_filterbank_packed_f32')
        return out

    def get_config(self):
        config = super(ApplyFilterbank, self).get_config()
        config.update(
            {
                'type': self.type,
                'filterbank_kwargs': self.filterbank_kwargs,
                'data_format': self.data_format_original,
            }
        )
        return config


@register_keras_serializable(package='Kapre')
class Delta(Layer):
    """Delta: a local estimate of the derivative along the time axis
    (reference: time_frequency.py:561-644).  ``tf.pad(mode)`` of ``(win_length - 1) // 2`` frames
    on both sides, correlation with ``[-n .. n]``, division by ``2 * sum(i^2)`` -- one kernel.
    Input / output: (b, t, f, ch) for ``channels_last``, (b, ch, t, f) for ``channels_first``."""

    def __init__(self, win_length=5, mode='symmetric', data_format='default', **kwargs):
        super(Delta, self).__init__(**kwargs)
        backend.validate_data_format_str(data_format)
        if isinstance(data_format, dict):
            data_format = data_format['config']
        if not win_length >= 3:
            raise ValueError(
                'win_length should be equal or bigger than 3, but it is %d' % win_length)
        if win_length % 2 != 1:
            raise ValueError('win_length should be an odd number, but it is %d' % win_length)
        if mode.lower() not in ('symmetric', 'reflect', 'constant'):
            raise ValueError(
                'mode.lower() should be one of {}'.format(str(('symmetric', 'reflect', 'constant')))
                + 'but it is {}'.format(mode))
        self.data_format_original = data_format
        self.data_format = _resolve_format(data_format)
        self.win_length = win_length
        self.mode = mode
        self.n = (self.win_length - 1) // 2
        self.denom = 2 * sum([_n ** 2 for _n in range(1, self.n + 1, 1)])

    def call(self, x):
        if autograd.needs_grad(x):
            return autograd.delta(self, autograd.prep(x, 'float32'))
        return self._forward(x)

    def _forward(self, x):
        import torch

        x = _ffi.as_device_f32(x)
        if x.dim() != 4:
            raise ValueError('Delta expects a rank-4 input, got shape %s' % (tuple(x.shape),))
        if self.data_format == _CH_LAST_STR:
            b, t, f, c = x.shape
        else:
            b, c, t, f = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().kpr_delta_f32(
                _ffi.ptr(x), b, c, t, f, _ffi.layout(self.data_format), self.win_length,
                _ffi.PAD_MODES[self.mode.lower()], _ffi.ptr(out), _ffi.current_stream_ptr()),
                'kpr_delta_f32')
        return out

    def get_config(self):
        config = super(Delta, self).get_config()
        config.update({'win_length': self.win_length, 'mode': self.mode,
                       'data_format': self.data_format_original})
        return config


# --------------------------------------------------------------------------------------------
# fused execution
# --------------------------------------------------------------------------------------------
class _MelPlan:
    """Everything of one fused call that does not depend on the data: geometry, output shape,
    workspace and the ctypes argument objects.  Cached per (input shape, device, stream, dB)."""

    __slots__ = ('g', 'g_ref', 'out_shape', 'n_filt', 'db', 'db_ref', 'ws', 'ws_ptr', 'ws_bytes',
                 'win', 'win_ptr', 'fb', 'fb_ptr', 'fbp', 'fbp_ptr', 'kr', 'kr_ptr', 'stream',
                 'fb_layer', 'db_layer')


def _mel_plan(stft, fb_layer, db_layer, x, stream_ptr):
    import torch

    L = _ffi.lib()
    plan = _MelPlan()
    plan.g = stft._geom(x)
    plan.g_ref = ctypes.byref(plan.g)
    n_frames = int(L.kpr_num_frames(plan.g_ref))
    if n_frames < 0:
        _ffi.check(-1, 'kpr_num_frames')
    n_freq, n_filt = fb_layer.filterbank.shape
    if n_freq != int(stft.n_fft) // 2 + 1:
        raise ValueError('filterbank has %d frequency rows but the STFT produces %d bins'
                         % (n_freq, int(stft.n_fft) // 2 + 1))
    plan.n_filt = n_filt
    plan.db = db_layer._db_params() if db_layer is not None else _ffi.DbParams(0, 1.0, 1e-5, 80.0)
    plan.db_ref = ctypes.byref(plan.db)
    plan.out_shape = stft._out_shape(plan.g, n_frames, n_filt)
    plan.fb_layer, plan.db_layer = fb_layer, db_layer       # keep the key's objects alive (no id() reuse)
    plan.win = stft._window(x.device)
    plan.win_ptr = _ffi.ptr(plan.win)
    plan.fb = fb_layer._fb_device(x.device)
    plan.fb_ptr = _ffi.ptr(plan.fb)
    try:
        plan.fbp = fb_layer._fb_packed_device(x.device)
    except RuntimeError:
        plan.fbp = None                 # more filter tiles than the packed schedule holds: generic product
    plan.fbp_ptr = _ffi.ptr(plan.fbp)
    # without a packed filterbank kpr_mel_f32 takes its two-kernel path, which stages the spectrum in the workspace
    plan.ws_bytes = int(L.kpr_mel_workspace_bytes(plan.g_ref, n_filt, plan.db_ref) if plan.fbp is not None
                        else L.kpr_mel_workspace_bytes_unpacked(plan.g_ref, n_filt))
    if plan.ws_bytes < 0:
        _ffi.check(-1, 'kpr_mel_workspace_bytes')
    plan.ws = _PLAN_WORKSPACES.get(plan.ws_bytes, x.device, stream_ptr)
    plan.ws_ptr = _ffi.ptr(plan.ws)
    plan.kr = fb_layer._fb_kranges()
    plan.kr_ptr = plan.kr.ctypes.data_as(ctypes.c_void_p)
    plan.stream = ctypes.c_void_p(stream_ptr)
    return plan


def fused_melspectrogram(stft: STFT, fb_layer: ApplyFilterbank, db_layer, x):
    """STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel] in one launch (kpr_mel_f32)."""
    import torch

    x = _ffi.as_device_f32(x)
    dev = x.device
    stream_ptr = torch.cuda.current_stream(dev).cuda_stream
    db_key = None if db_layer is None else (db_layer.ref_value, db_layer.amin, db_layer.dynamic_range)
    key = (tuple(x.shape), dev.index, stream_ptr, db_key, id(fb_layer), fb_layer._fb_version,
           stft._plan_version())
    cache = stft.__dict__.setdefault('_mel_plans', {})
    plan = cache.get(key)
    if plan is None or plan.fb_layer is not fb_layer:
        if len(cache) >= 64:
            cache.pop(next(iter(cache)))        # oldest plan; its scratch lives in the pool, not in the plan
        plan = cache[key] = _mel_plan(stft, fb_layer, db_layer, x, stream_ptr)
    out = torch.empty(plan.out_shape, dtype=torch.float32, device=dev)
    L = _ffi.lib()
    if torch.cuda.current_device() != dev.index:
        with torch.cuda.device(dev):
            rc = L.kpr_mel_f32(x.data_ptr(), plan.g_ref, plan.win_ptr, plan.fb_ptr, plan.fbp_ptr,
                               plan.n_filt, plan.kr_ptr, plan.db_ref, out.data_ptr(), plan.ws_ptr,
                               plan.ws_bytes, plan.stream)
    else:
        rc = L.kpr_mel_f32(x.data_ptr(), plan.g_ref, plan.win_ptr, plan.fb_ptr, plan.fbp_ptr,
                           plan.n_filt, plan.kr_ptr, plan.db_ref, out.data_ptr(), plan.ws_ptr,
                           plan.ws_bytes, plan.stream)
    if rc == -4 and plan.fbp is not None:
        # KPR_E_WORKSPACE: a bank whose schedule none of the fused kernels holds (dense matrices): the two-launch path stages the
        # spectrum and needs kpr_mel_workspace_bytes_unpacked() -- the plan is upgraded once and keeps the larger workspace
        ws_bytes = int(L.kpr_mel_workspace_bytes_unpacked(plan.g_ref, plan.n_filt))
        if ws_bytes > plan.ws_bytes:
            plan.ws_bytes = ws_bytes
            plan.ws = _PLAN_WORKSPACES.get(plan.ws_bytes, dev, stream_ptr)
            plan.ws_ptr = _ffi.ptr(plan.ws)
            with torch.cuda.device(dev):
                rc = L.kpr_mel_f32(x.data_ptr(), plan.g_ref, plan.win_ptr, plan.fb_ptr, plan.fbp_ptr,
                                   plan.n_filt, plan.kr_ptr, plan.db_ref, out.data_ptr(), plan.ws_ptr,
                                   plan.ws_bytes, plan.stream)
    if rc:
        _ffi.check(rc, 'kpr_mel_f32')
    return out


def _run_mel_group(group, x):
    db_layer = group[3] if len(group) == 4 else None
    return fused_melspectrogram(group[0], group[2], db_layer, x)


def _run_stft_mag(group, x):
    return group[0]._run(x, _ffi.OUT_MAGNITUDE)


def _run_stft_phase(group, x):
    return group[0]._run(x, _ffi.OUT_PHASE)


def fuse_and_run(layers, x):
    """Run a flat list of layers, fusing the Kapre chains that have a single-kernel form.  When ``x`` carries gradient
    a fused group runs as one autograd node (forward = the fused launch, backward = recomputation through the
    individual layers, kapre_amd/autograd.py)."""
    i, n = 0, len(layers)
    while i < n:
        layer = layers[i]
        group, runner = None, None
        # (layers of different compute dtypes are never fused; the single-kernel mel chain is float32 only)
        if type(layer) is STFT and i + 1 < n and layers[i + 1]._f64 == layer._f64:
            nxt = layers[i + 1]
            if type(nxt) is Magnitude:
                # STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel]
                if (not layer._f64 and i + 2 < n and type(layers[i + 2]) is ApplyFilterbank
                        and not layers[i + 2]._f64
                        and not (i + 3 < n and type(layers[i + 3]) is MagnitudeToDecibel and layers[i + 3]._f64)
                        and hasattr(layers[i + 2], 'filterbank')
                        and layers[i + 2].data_format == layer.output_data_format):
                    step = 4 if i + 3 < n and type(layers[i + 3]) is MagnitudeToDecibel else 3
                    group, runner = layers[i:i + step], _run_mel_group
                else:
                    # STFT -> Magnitude (magnitude written straight from the FFT kernel)
                    group, runner = layers[i:i + 2], _run_stft_mag
            elif type(nxt) is Phase:
                group, runner = layers[i:i + 2], _run_stft_phase
        if group is not None:
            x = autograd.chain(group, x, runner) if autograd.needs_grad(x) else runner(group, x)
            i += len(group)
            continue
        x = layer(x)
        i += 1
    return x
