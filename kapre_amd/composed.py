"""Composed layers -- mirror of /root/reference/kapre/composed.py for the hot path.

``get_stft_magnitude_layer`` (composed.py:32-135), ``get_melspectrogram_layer`` (:138-261),
``get_log_frequency_spectrogram_layer`` (:264-385), ``get_perfectly_reconstructing_stft_istft``
(:388-417), ``get_stft_mag_phase`` (:420-511): same signatures, defaults and layer order as the
reference.  The returned ``Sequential`` exposes the individual layers through ``.layers`` (users
may re-add them to their own model, composed.py:1-13) and, when called, runs the whole chain as a
single fused HIP launch (kapre_amd.time_frequency.fuse_and_run).
"""
from .keras_shim import Sequential, Layer
from .time_frequency import (
    STFT,
    InverseSTFT,
    Magnitude,
    Phase,
    MagnitudeToDecibel,
    ApplyFilterbank,
)
from . import backend
from .backend import _CH_FIRST_STR, _CH_LAST_STR, _CH_DEFAULT_STR


def get_stft_magnitude_layer(
    input_shape=None,
    n_fft=2048,
    win_length=None,
    hop_length=None,
    window_name=None,
    pad_begin=False,
    pad_end=False,
    return_decibel=False,
    db_amin=1e-5,
    db_ref_value=1.0,
    db_dynamic_range=80.0,
    input_data_format='default',
    output_data_format='default',
    name='stft_magnitude',
):
    """``Sequential([STFT, Magnitude, (MagnitudeToDecibel)])`` (reference: composed.py:32-135)."""
    backend.validate_data_format_str(input_data_format)
    backend.validate_data_format_str(output_data_format)

    stft_kwargs = {}
    if input_shape is not None:
        stft_kwargs['input_shape'] = input_shape

    waveform_to_stft = STFT(
        **stft_kwargs,
        n_fft=n_fft,
        win_length=win_length,
        hop_length=hop_length,
        window_name=window_name,
        pad_begin=pad_begin,
        pad_end=pad_end,
        input_data_format=input_data_format,
        output_data_format=output_data_format,
    )

    stft_to_stftm = Magnitude()

    layers = [waveform_to_stft, stft_to_stftm]
    if return_decibel:
        mag_to_decibel = MagnitudeToDecibel(
            ref_value=db_ref_value, amin=db_amin, dynamic_range=db_dynamic_range
        )
        layers.append(mag_to_decibel)

    return Sequential(layers, name=name)


def get_melspectrogram_layer(
    input_shape=None,
    n_fft=2048,
    win_length=None,
    hop_length=None,
    window_name=None,
    pad_begin=False,
    pad_end=False,
    sample_rate=22050,
    n_mels=128,
    mel_f_min=0.0,
    mel_f_max=None,
    mel_htk=False,
    mel_norm='slaney',
    return_decibel=False,
    db_amin=1e-5,
    db_ref_value=1.0,
    db_dynamic_range=80.0,
    input_data_format='default',
    output_data_format='default',
    name='melspectrogram',
):
    """``Sequential([STFT, Magnitude, ApplyFilterbank('mel'), (MagnitudeToDecibel)])``
    (reference: composed.py:138-261).  The filterbank is applied to the magnitude (power 1) and
    its layer uses ``output_data_format`` (composed.py:250-252)."""
    backend.validate_data_format_str(input_data_format)
    backend.validate_data_format_str(output_data_format)

    stft_kwargs = {}
    if input_shape is not None:
        stft_kwargs['input_shape'] = input_shape

    waveform_to_stft = STFT(
        **stft_kwargs,
        n_fft=n_fft,
        win_length=win_length,
        hop_length=hop_length,
        window_name=window_name,
        pad_begin=pad_begin,
        pad_end=pad_end,
        input_data_format=input_data_format,
        output_data_format=output_data_format,
    )

    stft_to_stftm = Magnitude()

    kwargs = {
        'sample_rate': sample_rate,
        'n_freq': n_fft // 2 + 1,
        'n_mels': n_mels,
        'f_min': mel_f_min,
        'f_max': mel_f_max,
        'htk': mel_htk,
        'norm': mel_norm,
    }
    stftm_to_melgram = ApplyFilterbank(
        type='mel', filterbank_kwargs=kwargs, data_format=output_data_format
    )

    layers = [waveform_to_stft, stft_to_stftm, stftm_to_melgram]
    if return_decibel:
        mag_to_decibel = MagnitudeToDecibel(
            ref_value=db_ref_value, amin=db_amin, dynamic_range=db_dynamic_range
        )
        layers.append(mag_to_decibel)

    return Sequential(layers, name=name)


def get_log_frequency_spectrogram_layer(
    input_shape=None,
    n_fft=2048,
    win_length=None,
    hop_length=None,
    window_name=None,
    pad_begin=False,
    pad_end=False,
    sample_rate=22050,
    log_n_bins=84,
    log_f_min=None,
    log_bins_per_octave=12,
    log_spread=0.125,
    return_decibel=False,
    db_amin=1e-5,
    db_ref_value=1.0,
    db_dynamic_range=80.0,
    input_data_format='default',
    output_data_format='default',
    name='log_frequency_spectrogram',
):
    """``Sequential([STFT, Magnitude, ApplyFilterbank('log'), (MagnitudeToDecibel)])``
    (reference: composed.py:264-385)."""
    backend.validate_data_format_str(input_data_format)
    backend.validate_data_format_str(output_data_format)

    stft_kwargs = {}
    if input_shape is not None:
        stft_kwargs['input_shape'] = input_shape

    waveform_to_stft = STFT(
        **stft_kwargs,
        n_fft=n_fft,
        win_length=win_length,
        hop_length=hop_length,
        window_name=window_name,
        pad_begin=pad_begin,
        pad_end=pad_end,
        input_data_format=input_data_format,
        output_data_format=output_data_format,
    )

    stft_to_stftm = Magnitude()

    _log_filterbank = {
        'sample_rate': sample_rate,
        'n_freq': n_fft // 2 + 1,
        'n_bins': log_n_bins,
        'bins_per_octave': log_bins_per_octave,
        'f_min': log_f_min,
        'spread': log_spread,
    }
    stftm_to_loggram = ApplyFilterbank(
        type='log', filterbank_kwargs=_log_filterbank, data_format=output_data_format
    )

    layers = [waveform_to_stft, stft_to_stftm, stftm_to_loggram]
    if return_decibel:
        mag_to_decibel = MagnitudeToDecibel(
            ref_value=db_ref_value, amin=db_amin, dynamic_range=db_dynamic_range
        )
        layers.append(mag_to_decibel)

    return Sequential(layers, name=name)


def get_perfectly_reconstructing_stft_istft(
    n_fft,
    hop_length,
    waveform_data_format,
    stft_data_format,
    stft_name=None,
    istft_name=None,
):
    """A matched ``(STFT, InverseSTFT)`` pair (reference: composed.py:388-417): hann window,
    ``win_length=n_fft``, ``pad_begin=True``, ``pad_end=True``; the caller trims
    ``[n_fft-hop : n_fft-hop+len_src]`` of the reconstruction."""
    stft = STFT(
        n_fft=n_fft,
        win_length=n_fft,
        hop_length=hop_length,
        window_name='hann_window',
        pad_begin=True,
        pad_end=True,
        input_data_format=waveform_data_format,
        output_data_format=stft_data_format,
        name=stft_name,
    )

    istft = InverseSTFT(
        n_fft=n_fft,
        win_length=n_fft,
        hop_length=hop_length,
        forward_window_name='hann_window',
        input_data_format=stft_data_format,
        output_data_format=waveform_data_format,
        name=istft_name,
    )
    return stft, istft


class _MagPhase(Layer):
    """Functional-model stand-in returned by get_stft_mag_phase: STFT once, then magnitude
    (optionally in decibel) and phase concatenated along the channel axis."""

    def __init__(self, stft, mag, phase, db, ch_axis, name):
        super().__init__(name=name)
        self.stft, self.mag, self.phase, self.db, self.ch_axis = stft, mag, phase, db, ch_axis
        self.layers = [stft, mag, phase] + ([db] if db is not None else [])

    def call(self, x):
        import torch

        s = self.stft(x)
        m = self.mag(s)
        if self.db is not None:
            m = self.db(m)
        return torch.cat([m, self.phase(s)], dim=self.ch_axis)


def get_stft_mag_phase(
    input_shape,
    n_fft=2048,
    win_length=None,
    hop_length=None,
    window_name=None,
    pad_begin=False,
    pad_end=False,
    return_decibel=False,
    db_amin=1e-5,
    db_ref_value=1.0,
    db_dynamic_range=80.0,
    input_data_format='default',
    output_data_format='default',
    name='stft_mag_phase',
):
    """Magnitude and phase of the STFT concatenated on the channel axis
    (reference: composed.py:420-511)."""
    backend.validate_data_format_str(input_data_format)
    backend.validate_data_format_str(output_data_format)

    waveform_to_stft = STFT(
        n_fft=n_fft,
        win_length=win_length,
        hop_length=hop_length,
        window_name=window_name,
        pad_begin=pad_begin,
        pad_end=pad_end,
        input_data_format=input_data_format,
        output_data_format=output_data_format,
    )
    db = None
    if return_decibel:
        db = MagnitudeToDecibel(ref_value=db_ref_value, amin=db_amin,
                                dynamic_range=db_dynamic_range)
    if output_data_format == _CH_DEFAULT_STR:
        output_data_format = backend.image_data_format()
    ch_axis = 1 if output_data_format == _CH_FIRST_STR else 3
    return _MagPhase(waveform_to_stft, Magnitude(), Phase(), db, ch_axis, name)
