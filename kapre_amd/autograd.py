"""Backward passes of the hot-path layers: Kapre's layers are TensorFlow graphs, so a front end placed
inside ``model.fit`` / ``tf.GradientTape`` is differentiable (reference: time_frequency.py:146-187,
:289-319, :351-359, :402, :535-548; backend.py:186-192).  Here a layer called on a torch tensor that
``requires_grad`` returns a tensor with a ``grad_fn`` whose backward runs on the same HIP library:

* the LINEAR layers are their own adjoints' kernels -- ``STFT^T`` is an inverse-STFT launch
  (window ``n_fft * w``, interior bins halved), ``InverseSTFT^T`` an STFT launch (window
  ``2 / n_fft * w_synth``, edge bins halved), ``ApplyFilterbank^T`` / ``LogmelToMFCC^T`` the same
  GEMM with the transposed matrix; ``Frame`` / ``Energy`` / ``Delta`` have gather-form adjoint kernels;
* ``Magnitude`` / ``Phase`` / ``MagnitudeToDecibel`` have elementwise backward kernels
  (``csrc/kpr_grad_kernels.h``) that follow TensorFlow's registered gradients, including the part of
  the decibel gradient that reaches an item's maximum through the dynamic-range floor;
* a fused chain (one forward launch) is differentiated by recomputation: backward re-runs the chain
  layer by layer with these functions and differentiates that.

Complex cotangents use the convention torch and TensorFlow share (``dL/dRe + i dL/dIm``).
torch.autograd is the tape; every arithmetic step is a HIP launch of this package.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi

_CH_LAST_STR = 'channels_last'


def needs_grad(x) -> bool:
    """True when the caller expects the layer output to be differentiable w.r.t. ``x``."""
    try:
        import torch
    except ImportError:         # pragma: no cover
        return False
    return isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled()


def prep(x, dtype_name: str):
    """Device / dtype / layout normalisation of a tensor that carries gradient, done with torch ops so that it is
    recorded (the forward-only path does the same through _ffi.as_device_* outside the tape)."""
    import torch

    _ffi.require_gpu()
    dtype = getattr(torch, dtype_name)
    if not x.is_cuda:
        x = x.to(torch.device('cuda', torch.cuda.current_device()))
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def _stream():
    return _ffi.current_stream_ptr()


def _edge_scale(spec, n_fft: int, channels_last: bool, s_edge: float, s_mid: float, out=None):
    """``spec`` (complex, frequency axis 2 for channels_last, 3 otherwise) times s_edge on DC / Nyquist, s_mid elsewhere."""
    import torch

    f64 = spec.dtype == torch.complex128
    if out is None:
        out = torch.empty_like(spec)
    k = spec.shape[2] if channels_last else spec.shape[3]
    inner = spec.shape[3] if channels_last else 1
    fn = _ffi.lib().kpr_spec_edge_scale_c128 if f64 else _ffi.lib().kpr_spec_edge_scale_c64
    with torch.cuda.device(spec.device):
        _ffi.check(fn(_ffi.ptr(spec), spec.numel(), int(k), int(inner), int(n_fft), float(s_edge), float(s_mid),
                      _ffi.ptr(out), _stream()), 'kpr_spec_edge_scale')
    return out


def _run_istft(spec, n_fft, win_length, hop, window, wave_fmt, spec_fmt):
    """kpr_istft_f32 / f64 on a contiguous complex spectrogram with an arbitrary 'synthesis' window tensor."""
    import torch

    f64 = spec.dtype == torch.complex128
    if spec_fmt == _CH_LAST_STR:
        b, f, k, c = spec.shape
    else:
        b, c, f, k = spec.shape
    g = _ffi.StftGeom(b, c, 0, int(n_fft), int(win_length), int(hop), 0, 0, _ffi.layout(wave_fmt), _ffi.layout(spec_fmt))
    t_out = (f - 1) * int(hop) + int(win_length) if f > 0 else 0
    shape = (b, t_out, c) if wave_fmt == _CH_LAST_STR else (b, c, t_out)
    out = torch.empty(shape, dtype=torch.float64 if f64 else torch.float32, device=spec.device)
    if out.numel() == 0:
        return out
    L = _ffi.lib()
    with torch.cuda.device(spec.device):
        if f64:
            ws_bytes = int(L.kpr_istft_f64_workspace_bytes(ctypes.byref(g), f))
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=spec.device)
            _ffi.check(L.kpr_istft_f64(_ffi.ptr(spec), ctypes.byref(g), f, _ffi.ptr(window), _ffi.ptr(out),
                                       _ffi.ptr(ws), ws_bytes, _stream()), 'kpr_istft_f64')
        else:
            ws_bytes = int(L.kpr_istft_workspace_bytes(ctypes.byref(g), f))
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=spec.device)
            _ffi.check(L.kpr_istft_f32(_ffi.ptr(spec), ctypes.byref(g), f, _ffi.ptr(window), _ffi.ptr(out),
                                       _ffi.ptr(ws), ws_bytes, _stream()), 'kpr_istft_f32')
    return out


def _run_stft(wave, n_fft, win_length, hop, window, wave_fmt, spec_fmt):
    """kpr_stft_f32 / f64 (complex output, no padding) on a contiguous waveform with an arbitrary window tensor."""
    import torch

    f64 = wave.dtype == torch.float64
    if wave_fmt == _CH_LAST_STR:
        b, t, c = wave.shape
    else:
        b, c, t = wave.shape
    g = _ffi.StftGeom(b, c, t, int(n_fft), int(win_length), int(hop), 0, 0, _ffi.layout(wave_fmt), _ffi.layout(spec_fmt))
    L = _ffi.lib()
    n_frames = int(L.kpr_num_frames(ctypes.byref(g)))
    if n_frames < 0:
        _ffi.check(-1, 'kpr_num_frames')
    k = int(n_fft) // 2 + 1
    shape = (b, n_frames, k, c) if spec_fmt == _CH_LAST_STR else (b, c, n_frames, k)
    out = torch.empty(shape, dtype=torch.complex128 if f64 else torch.complex64, device=wave.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(wave.device):
        if f64:
            _ffi.check(L.kpr_stft_f64(_ffi.ptr(wave), ctypes.byref(g), _ffi.ptr(window), _ffi.ptr(out),
                                      _ffi.OUT_COMPLEX, _stream()), 'kpr_stft_f64')
        else:
            ws_bytes = int(L.kpr_stft_workspace_bytes(ctypes.byref(g), _ffi.OUT_COMPLEX))
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=wave.device)
            _ffi.check(L.kpr_stft_f32(_ffi.ptr(wave), ctypes.byref(g), _ffi.ptr(window), _ffi.ptr(out),
                                      _ffi.OUT_COMPLEX, _ffi.ptr(ws), ws_bytes, _stream()), 'kpr_stft_f32')
    return out


def _scaled_window(layer, key, device, f64: bool, scale: float):
    """``scale * layer.window_fn(win_length)`` on the device, cached on the layer."""
    from . import backend

    def make():
        w = backend.window_values(layer.window_fn, int(layer.win_length), np.float64)
        return (w * scale).astype(np.float64 if f64 else np.float32)

    return layer._consts.get((key, f64, int(layer.win_length), int(layer.n_fft), id(layer.window_fn)), device, make)


# ---------------------------------------------------------------------------------------------
# STFT
# ---------------------------------------------------------------------------------------------
def stft_vjp(layer, gspec, x_shape):
    """Cotangent of the waveform from the cotangent of ``STFT.call``'s complex output.

    X[f, k] = sum_n w[n] x[f hop + n - pad_left] e^{-2 pi i k n / N}, k = 0 .. N/2, so
    dL/dx[t] = sum_f w[n] (Re G[f,0] + (-1)^n Re G[f,N/2] + sum_{0<k<N/2} Re(G[f,k] e^{+2 pi i k n / N})), n = t + pad_left - f hop:
    N * irfft of (G with the interior bins halved), windowed by w and overlap-added -- one inverse-STFT launch."""
    import torch

    n_fft, win, hop = int(layer.n_fft), int(layer.win_length), int(layer.hop_length)
    ch_last_spec = layer.output_data_format == _CH_LAST_STR
    f64 = gspec.dtype == torch.complex128
    gspec = gspec.contiguous()
    half = _edge_scale(gspec, n_fft, ch_last_spec, 1.0, 0.5)
    window = _scaled_window(layer, 'vjp_window', gspec.device, f64, float(n_fft))
    ola = _run_istft(half, n_fft, win, hop, window, layer.input_data_format, layer.output_data_format)
    gx = torch.zeros(x_shape, dtype=ola.dtype, device=ola.device)
    pad_left = n_fft - hop if layer.pad_begin else 0
    t_axis = 1 if layer.input_data_format == _CH_LAST_STR else 2
    t_in, t_ola = x_shape[t_axis], ola.shape[t_axis]
    n = min(t_in, t_ola - pad_left)
    if n > 0:
        gx.narrow(t_axis, 0, n).copy_(ola.narrow(t_axis, pad_left, n))
    return gx


def istft_vjp(layer, gwave, n_frames):
    """Cotangent of the (n_fft/2+1)-bin spectrogram from the cotangent of ``InverseSTFT.call``'s waveform.

    y = overlap-add of w_s[n] irfft(X[f])[n]; irfft reads Re / Im of the interior bins with weight 2/N and the real
    parts of DC / Nyquist with weight 1/N, so G[f] = rfft(w_s * gy[f hop : f hop + win]) * (2/N, edges 1/N):
    one STFT launch with window 2 w_s / N, then the edge bins halved."""
    n_fft, win, hop = int(layer.n_fft), int(layer.win_length), int(layer.hop_length)
    import torch

    f64 = gwave.dtype == torch.float64
    gwave = gwave.contiguous()
    window = _scaled_window(layer, 'vjp_window', gwave.device, f64, 2.0 / n_fft)
    spec = _run_stft(gwave, n_fft, win, hop, window, layer.output_data_format, layer.input_data_format)
    f_axis = 1 if layer.input_data_format == _CH_LAST_STR else 2
    if spec.shape[f_axis] != n_frames:     # cannot happen: (F - 1) hop + win samples frame into exactly F frames
        raise RuntimeError('InverseSTFT backward: %d frames, expected %d' % (spec.shape[f_axis], n_frames))
    return _edge_scale(spec, n_fft, layer.input_data_format == _CH_LAST_STR, 0.5, 1.0, out=spec)


def _functions():
    """The torch.autograd.Function classes (built on first use: torch is imported lazily everywhere in this package)."""
    global _FN
    if _FN is not None:
        return _FN
    import torch
    from torch.autograd.function import once_differentiable

    class STFTFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer):
            ctx.layer, ctx.x_shape = layer, tuple(x.shape)
            return layer._run(x.detach(), _ffi.OUT_COMPLEX)

        @staticmethod
        def backward(ctx, g):
            return stft_vjp(ctx.layer, g, ctx.x_shape), None

    class ISTFTFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer):
            ctx.layer, ctx.x_shape = layer, tuple(x.shape)
            return layer._forward(x.detach())

        @staticmethod
        def backward(ctx, g):
            layer, shape = ctx.layer, ctx.x_shape
            ch_last = layer.input_data_format == _CH_LAST_STR
            f_axis, k_axis = (1, 2) if ch_last else (2, 3)
            gs = istft_vjp(layer, g, shape[f_axis])
            k_have, k_in = gs.shape[k_axis], shape[k_axis]
            if k_in > k_have:        # the forward cropped the frequency axis: the cropped bins get no gradient
                full = torch.zeros(shape, dtype=gs.dtype, device=gs.device)
                full.narrow(k_axis, 0, k_have).copy_(gs)
                gs = full
            elif k_in < k_have:      # the forward zero-padded it
                gs = gs.narrow(k_axis, 0, k_in).contiguous()
            return gs, None

    class CplxToRealFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer, phase):
            ctx.phase = phase
            y = layer._forward(x.detach())
            # the forward may have cast its input (complex128 -> layer precision never happens; numpy never gets here)
            ctx.save_for_backward(x.detach())
            return y

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            f64 = x.dtype == torch.complex128
            x = x.contiguous()
            g = g.contiguous().to(torch.float64 if f64 else torch.float32)
            gx = torch.empty_like(x)
            L = _ffi.lib()
            fn = ((L.kpr_angle_c128_bwd if f64 else L.kpr_angle_c64_bwd) if ctx.phase
                  else (L.kpr_abs_c128_bwd if f64 else L.kpr_abs_c64_bwd))
            with torch.cuda.device(x.device):
                _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(g), x.numel(), _ffi.ptr(gx), _stream()), 'kpr_abs/angle_bwd')
            return gx, None, None

    class MatrixFn(torch.autograd.Function):
        """y = x . M over the frequency axis (ApplyFilterbank, LogmelToMFCC); backward = the same GEMM with M^T."""

        @staticmethod
        def forward(ctx, x, layer, matrix_t, data_format):
            ctx.matrix_t, ctx.data_format = matrix_t, data_format
            return layer._forward(x.detach())

        @staticmethod
        def backward(ctx, g):
            mt = ctx.matrix_t                         # (n_out, n_in) on the device, dtype of the layer
            f64 = mt.dtype == torch.float64
            g = g.contiguous().to(mt.dtype)
            if ctx.data_format == _CH_LAST_STR:
                b, f, m, c = g.shape
            else:
                b, c, f, m = g.shape
            n_in = int(mt.shape[1])
            shape = (b, f, n_in, c) if ctx.data_format == _CH_LAST_STR else (b, c, f, n_in)
            gx = torch.empty(shape, dtype=mt.dtype, device=g.device)
            L = _ffi.lib()
            with torch.cuda.device(g.device):
                if f64:
                    _ffi.check(L.kpr_apply_filterbank_f64(_ffi.ptr(g), b, c, f, m, _ffi.layout(ctx.data_format),
                                                          _ffi.ptr(mt), n_in, _ffi.ptr(gx), _stream()),
                               'kpr_apply_filterbank_f64 (backward)')
                else:
                    _ffi.check(L.kpr_apply_filterbank_f32(_ffi.ptr(g), b, c, f, m, _ffi.layout(ctx.data_format),
                                                          _ffi.ptr(mt), n_in, ctypes.c_void_p(0), _ffi.ptr(gx),
                                                          _stream()), 'kpr_apply_filterbank_f32 (backward)')
            return gx, None, None, None

    class DbFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer):
            ctx.params = (float(layer.ref_value), float(layer.amin), float(layer.dynamic_range))
            y = layer._forward(x.detach())
            ctx.save_for_backward(x.detach().to(y.dtype).contiguous())
            return y

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            ref, amin, dyn = ctx.params
            g = g.contiguous().to(x.dtype)
            gx = torch.empty_like(x)
            if x.dim() > 1:
                n_items = x.shape[0]
                item = x.numel() // max(n_items, 1)
            else:
                n_items, item = 1, x.numel()
            L = _ffi.lib()
            with torch.cuda.device(x.device):
                if x.dtype == torch.float64:
                    _ffi.check(L.kpr_mag_to_db_bwd_f64(_ffi.ptr(x), _ffi.ptr(g), n_items, item, ref, amin, dyn,
                                                       _ffi.ptr(gx), _stream()), 'kpr_mag_to_db_bwd_f64')
                else:
                    db = _ffi.DbParams(1, ref, amin, dyn)
                    _ffi.check(L.kpr_mag_to_db_bwd_f32(_ffi.ptr(x), _ffi.ptr(g), n_items, item, ctypes.byref(db),
                                                       _ffi.ptr(gx), _stream()), 'kpr_mag_to_db_bwd_f32')
            return gx, None

    class ChainFn(torch.autograd.Function):
        """A fused run of layers (one forward launch); backward recomputes the chain layer by layer with the
        differentiable functions above and differentiates that (activation recomputation)."""

        @staticmethod
        def forward(ctx, x, layers, run_fused):
            ctx.layers = layers
            ctx.save_for_backward(x.detach())
            return run_fused(layers, x.detach())

        @staticmethod
        @once_differentiable
        def backward(ctx, g):
            """Recomputation is chunked over the batch (every layer of the path treats batch items independently -- the
            decibel maximum is per item, backend.py:178-192): the unfused chain materialises the complex spectrogram and
            the magnitudes, several GB at once for a long batch, ~256 MB per chunk here.  Not differentiable
            a second time (the chain's backward passes are forward launches, not autograd graphs).
            A chunk is recomputed at a smaller batch than the forward ran at and may therefore take another kernel of the
            same arithmetic family (launch-size dispatch): values agree to float32 round-off, so a decibel value that sits
            EXACTLY on the clamp or on a tie of the item maximum may fall on the other side in the recomputation; the
            gradient at such boundaries is that of the recomputed values (tests/test_autograd.py pins ties with inputs that
            are exact in every kernel)."""
            (x,) = ctx.saved_tensors
            if x.numel() == 0:                                   # an empty batch has an empty gradient (no launch)
                return torch.zeros_like(x), None, None
            n = x.shape[0] if x.dim() > 0 else 1
            per_item = max(1, x.numel() // max(n, 1)) * x.element_size()
            first = next((l for l in ctx.layers if getattr(l, 'n_fft', None)), None)    # the chain's STFT, wherever it sits
            n_fft, hop = getattr(first, 'n_fft', None), getattr(first, 'hop_length', None)
            if n_fft and hop:                                   # complex spectrum + magnitude of one item
                per_item = per_item * (n_fft // 2 + 1) * 3 // max(1, hop)
            else:
                per_item *= 4
            step = int(max(1, min(n, (256 << 20) // max(1, per_item))))
            gx = torch.empty_like(x)
            for i0 in range(0, max(n, 1), step):
                with torch.enable_grad():
                    xr = x[i0:i0 + step].detach().requires_grad_(True)
                    y = xr
                    for layer in ctx.layers:
                        y = layer(y)
                    (gc,) = torch.autograd.grad(y, xr, g[i0:i0 + step].to(y.dtype))
                gx[i0:i0 + step] = gc
            return gx, None, None

    class FrameFn(torch.autograd.Function):
        """Frame (energy=False) and Energy (energy=True): both are sums over the frames that cover a sample."""

        @staticmethod
        def forward(ctx, x, layer, energy):
            ctx.layer, ctx.energy, ctx.x_shape = layer, energy, tuple(x.shape)
            xd = x.detach()
            if energy:
                ctx.save_for_backward(xd)
            return layer._forward(xd)

        @staticmethod
        def backward(ctx, g):
            layer = ctx.layer
            g = g.contiguous().to(torch.float32)
            fmt = layer.data_format
            L = _ffi.lib()
            if ctx.energy:
                (x,) = ctx.saved_tensors
            shape = tuple(x.shape) if ctx.energy else ctx.x_shape
            if fmt == _CH_LAST_STR:
                b, t, c = shape
            else:
                b, c, t = shape
            gx = torch.empty(shape, dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                if ctx.energy:
                    scale = layer.ref_duration / (layer.frame_length / layer.sample_rate)
                    _ffi.check(L.kpr_energy_bwd_f32(_ffi.ptr(x), _ffi.ptr(g), b, c, t, _ffi.layout(fmt),
                                                    int(layer.frame_length), int(layer.hop_length),
                                                    int(bool(layer.pad_end)), float(scale), _ffi.ptr(gx), _stream()),
                               'kpr_energy_bwd_f32')
                else:
                    _ffi.check(L.kpr_frame_bwd_f32(_ffi.ptr(g), b, c, t, _ffi.layout(fmt), int(layer.frame_length),
                                                   int(layer.hop_length), int(bool(layer.pad_end)), _ffi.ptr(gx),
                                                   _stream()), 'kpr_frame_bwd_f32')
            return gx, None, None

    class DeltaFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, layer):
            ctx.layer = layer
            return layer._forward(x.detach())

        @staticmethod
        def backward(ctx, g):
            layer = ctx.layer
            g = g.contiguous().to(torch.float32)
            if layer.data_format == _CH_LAST_STR:
                b, t, f, c = g.shape
            else:
                b, c, t, f = g.shape
            gx = torch.empty_like(g)
            with torch.cuda.device(g.device):
                _ffi.check(_ffi.lib().kpr_delta_bwd_f32(_ffi.ptr(g), b, c, t, f, _ffi.layout(layer.data_format),
                                                        int(layer.win_length), _ffi.PAD_MODES[layer.mode.lower()],
                                                        _ffi.ptr(gx), _stream()), 'kpr_delta_bwd_f32')
            return gx, None

    _FN = dict(stft=STFTFn, istft=ISTFTFn, c2r=CplxToRealFn, matrix=MatrixFn, db=DbFn, chain=ChainFn,
               frame=FrameFn, delta=DeltaFn)
    return _FN


_FN = None


def stft(layer, x):
    return _functions()['stft'].apply(x, layer)


def istft(layer, x):
    return _functions()['istft'].apply(x, layer)


def magnitude(layer, x):
    return _functions()['c2r'].apply(x, layer, 0)


def phase(layer, x):
    return _functions()['c2r'].apply(x, layer, 1)


def matrix(layer, x, matrix_t, data_format):
    return _functions()['matrix'].apply(x, layer, matrix_t, data_format)


def decibel(layer, x):
    return _functions()['db'].apply(x, layer)


def chain(layers, x, run_fused):
    return _functions()['chain'].apply(x, tuple(layers), run_fused)


def frame(layer, x):
    return _functions()['frame'].apply(x, layer, False)


def energy(layer, x):
    return _functions()['frame'].apply(x, layer, True)


def delta(layer, x):
    return _functions()['delta'].apply(x, layer)
