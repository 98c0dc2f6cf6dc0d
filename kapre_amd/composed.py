"""Composed layers -- mirror of /root/reference/kapre/composed.py for the hot path.

``get_stft_magnitude_layer`` (composed.py:32-135), ``get_melspectrogram_layer`` (:138-261),
``get_log_frequency_spectrogram_layer`` (:264-385), ``get_perfectly_reconstructing_stft_istft``
(:388-417), ``get_stft_mag_phase`` (:420-511): same signatures, defaults and layer order as the
reference.  The returned ``Sequential`` exposes the individual layers through ``.layers`` (users
may re-add them to their own model, composed.py:1-13) and, when called, runs the whole chain as a
single fused HIP launch (kapre_amd.time_frequency.fuse_and_run).

The public functions keep the reference's signatures and defaults verbatim (they are the drop-in
contract); everything below them is this package's own: one ``_spectrogram_chain`` builder that all
four analysis helpers share.
"""
from .keras_shim import Sequential, Layer, register_keras_serializable
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




def _spectrogram_chain(name, input_shape, stft_args, filterbank=None, decibel=None):
    """Build ``Sequential([STFT, Magnitude, (ApplyFilterbank), (MagnitudeToDecibel)])``.

    ``stft_args``: keyword arguments of ``STFT`` (both data formats are validated here first, as every
    reference helper does before constructing anything).  ``filterbank``: ``None`` or ``(type,
    filterbank_kwargs)`` -- the layer takes the STFT's *output* data format (composed.py:250-252,
    :374-376).  ``decibel``: ``None`` or ``(ref_value, amin, dynamic_range)``."""
    for key in ('input_data_format', 'output_data_format'):
        backend.validate_data_format_str(stft_args[key])
    if input_shape is not None:
        stft_args = dict(stft_args, input_shape=input_shape)
    chain = Sequential(name=name)
    chain.add(STFT(**stft_args))
    chain.add(Magnitude())
    if filterbank is not None:
        fb_type, fb_kwargs = filterbank
        chain.add(ApplyFilterbank(type=fb_type, filterbank_kwargs=fb_kwargs,
                                  data_format=stft_args['output_data_format']))
    if decibel is not None:
        ref_value, amin, dynamic_range = decibel
        chain.add(MagnitudeToDecibel(ref_value=ref_value, amin=amin, dynamic_range=dynamic_range))
    return chain


def _stft_args(n_fft, win_length, hop_length, window_name, pad_begin, pad_end, input_data_format,
               output_data_format):
    return dict(n_fft=n_fft, win_length=win_length, hop_length=hop_length, window_name=window_name,
                pad_begin=pad_begin, pad_end=pad_end, input_data_format=input_data_format,
                output_data_format=output_data_format)


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
    return _spectrogram_chain(
        name, input_shape,
        _stft_args(n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                   input_data_format, output_data_format),
        decibel=(db_ref_value, db_amin, db_dynamic_range) if return_decibel else None)


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
    mel = dict(sample_rate=sample_rate, n_freq=n_fft // 2 + 1, n_mels=n_mels, f_min=mel_f_min,
               f_max=mel_f_max, htk=mel_htk, norm=mel_norm)
    return _spectrogram_chain(
        name, input_shape,
        _stft_args(n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                   input_data_format, output_data_format),
        filterbank=('mel', mel),
        decibel=(db_ref_value, db_amin, db_dynamic_range) if return_decibel else None)


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
    log = dict(sample_rate=sample_rate, n_freq=n_fft // 2 + 1, n_bins=log_n_bins,
               bins_per_octave=log_bins_per_octave, f_min=log_f_min, spread=log_spread)
    return _spectrogram_chain(
        name, input_shape,
        _stft_args(n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                   input_data_format, output_data_format),
        filterbank=('log', log),
        decibel=(db_ref_value, db_amin, db_dynamic_range) if return_decibel else None)


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
    shared = dict(n_fft=n_fft, win_length=n_fft, hop_length=hop_length)
    analysis = STFT(window_name='hann_window', pad_begin=True, pad_end=True,
                    input_data_format=waveform_data_format, output_data_format=stft_data_format,
                    name=stft_name, **shared)
    synthesis = InverseSTFT(forward_window_name='hann_window', input_data_format=stft_data_format,
                            output_data_format=waveform_data_format, name=istft_name, **shared)
    return analysis, synthesis


@register_keras_serializable(package='Kapre', name='StftMagPhase')
class _MagPhase(Layer):
    """Functional-model stand-in returned by get_stft_mag_phase: STFT once, then magnitude
    (optionally in decibel) and phase concatenated along ``ch_axis``."""

    def __init__(self, stft, db, ch_axis, name):
        super().__init__(name=name)
        self.stft, self.mag, self.phase, self.db, self.ch_axis = stft, Magnitude(), Phase(), db, ch_axis
        self.layers = [self.stft, self.mag, self.phase] + ([db] if db is not None else [])

    def compute_output_shape(self, input_shape):
        shape = list(self.stft.compute_output_shape(input_shape))
        if shape[self.ch_axis] is not None:
            shape[self.ch_axis] = 2 * shape[self.ch_axis]
        return tuple(shape)

    def get_config(self):
        config = super().get_config()
        config.update({'stft': self.stft.get_config(), 'decibel': self.db.get_config() if self.db is not None else None,
                       'ch_axis': self.ch_axis,
                       'input_shape': list(self._input_shape_arg) if self._input_shape_arg is not None else None})
        return config

    @classmethod
    def from_config(cls, config):
        db = MagnitudeToDecibel.from_config(config['decibel']) if config.get('decibel') else None
        model = cls(STFT.from_config(config['stft']), db, config['ch_axis'], config['name'])
        if config.get('input_shape') is not None:                     # (upstream: a functional Model, which knows its Input)
            model._input_shape_arg = tuple(config['input_shape'])
        return model

    def call(self, x):
        import torch

        spectrum = self.stft(x)
        mag = self.mag(spectrum)
        if self.db is not None:
            mag = self.db(mag)
        return torch.cat([mag, self.phase(spectrum)], dim=self.ch_axis)


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
    (reference: composed.py:420-511).

    As upstream (composed.py:504), the concatenation axis is 1 only when ``output_data_format`` is
    literally ``'channels_first'`` and 3 otherwise -- ``'default'`` is NOT resolved through
    ``image_data_format()`` here, so a process whose Keras default is channels_first gets the
    reference's axis-3 concatenation, quirk included."""
    args = _stft_args(n_fft, win_length, hop_length, window_name, pad_begin, pad_end,
                      input_data_format, output_data_format)
    for key in ('input_data_format', 'output_data_format'):
        backend.validate_data_format_str(args[key])
    db = None
    if return_decibel:
        db = MagnitudeToDecibel(ref_value=db_ref_value, amin=db_amin, dynamic_range=db_dynamic_range)
    if input_shape is not None:
        args = dict(args, input_shape=input_shape)
    model = _MagPhase(STFT(**args), db, 1 if output_data_format == _CH_FIRST_STR else 3, name)
    model._input_shape_arg = tuple(input_shape) if input_shape is not None else None
    return model
