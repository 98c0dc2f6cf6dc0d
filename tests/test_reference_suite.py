"""-m gpu: the reference's OWN hot-path tests, parametrisation for parametrisation
(/root/reference/tests/test_time_frequency.py:72-125, :128-185, :188-267, :359-371, :390-444, :447-534,
:537-591 and tests/test_backend.py:13-31), with `import kapre_amd as kapre` where upstream imports
kapre, the numpy shim where it uses keras, and the float64 oracle standing in for the live
librosa calls (librosa / TensorFlow are not installable here; tests/librosa_standin.py restates the
few librosa functions upstream compares against and says which oracle functions they are made of).

Every assertion keeps upstream's tolerance; where north_star asks more (1e-4 relative), both hold.
"""
import numpy as np
import pytest

import kapre_oracle as o
import librosa_standin as librosa
from conftest import speech

import kapre_amd as kapre
from kapre_amd import STFT, Magnitude, Phase, Delta, backend
from kapre_amd.backend import _CH_FIRST_STR, _CH_LAST_STR
from kapre_amd.composed import (get_melspectrogram_layer, get_log_frequency_spectrogram_layer,
                                get_stft_mag_phase, get_perfectly_reconstructing_stft_istft,
                                get_stft_magnitude_layer)
from kapre_amd.keras_shim import Sequential, Input

pytestmark = pytest.mark.gpu


class K:
    image_data_format = staticmethod(backend.image_data_format)


def get_audio(data_format, n_ch, length=8000, batch_size=1):
    """tests/utils.py:13-35, verbatim semantics: the speech fixture tiled over channels and batch."""
    src = speech(length)
    src_mono = src.copy()
    len_src = len(src)
    src = np.expand_dims(src, axis=1)
    if n_ch != 1:
        src = np.tile(src, [1, n_ch])
    if data_format == 'default':
        data_format = K.image_data_format()
    if data_format == 'channels_last':
        input_shape = (len_src, n_ch)
    else:
        src = np.transpose(src)
        input_shape = (n_ch, len_src)
    batch_src = np.repeat([src], batch_size, axis=0)
    return src_mono, batch_src, input_shape


def allclose_phase(a, b, atol=1e-3):
    np.testing.assert_allclose(np.sin(a), np.sin(b), atol=atol)
    np.testing.assert_allclose(np.cos(a), np.cos(b), atol=atol)


def allclose_complex_numbers(a, b, atol=1e-3):
    np.testing.assert_equal(np.shape(a), np.shape(b))
    np.testing.assert_allclose(np.abs(a), np.abs(b), rtol=1e-5, atol=atol)
    np.testing.assert_allclose(np.real(a), np.real(b), rtol=1e-5, atol=atol)
    np.testing.assert_allclose(np.imag(a), np.imag(b), rtol=1e-5, atol=atol)


def north_star(got, want, rel=1e-4):
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(np.asarray(got) - want))) <= rel * scale


def _stft_model(input_shape, following_layer=None, **stft_kw):
    model = Sequential()
    model.add(Input(shape=input_shape))
    model.add(STFT(name='stft', **stft_kw))
    if following_layer is not None:
        model.add(following_layer)
    return model


# ---------------------------------------------------------------- test_time_frequency.py:72-125
@pytest.mark.parametrize('n_fft', [1000])
@pytest.mark.parametrize('hop_length', [None, 256])
@pytest.mark.parametrize('n_ch', [1, 2, 6])
@pytest.mark.parametrize('data_format', ['default', 'channels_first', 'channels_last'])
@pytest.mark.parametrize('batch_size', [1, 10])
def test_spectrogram_correctness(n_fft, hop_length, n_ch, data_format, batch_size):
    src_mono, batch_src, input_shape = get_audio(data_format=data_format, n_ch=n_ch, batch_size=batch_size)
    win_length = n_fft
    S_ref = librosa.stft(y=src_mono, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=False).T
    S_ref = np.tile(np.expand_dims(S_ref, axis=2), [1, 1, n_ch])
    if data_format == 'channels_first':
        S_ref = np.transpose(S_ref, (2, 0, 1))
    kw = dict(n_fft=n_fft, win_length=win_length, hop_length=hop_length, window_name=None, pad_end=False,
              input_data_format=data_format, output_data_format=data_format)
    out = _stft_model(input_shape, **kw).predict(batch_src)
    assert out.shape[0] == batch_size and out.dtype == np.complex64
    for item in out:                                   # upstream checks item 0; every item here
        allclose_complex_numbers(S_ref, item)
        north_star(item, S_ref)
    S = _stft_model(input_shape, Magnitude(), **kw).predict(batch_src)[0]
    np.testing.assert_allclose(np.abs(S_ref), S, atol=2e-4)
    S = _stft_model(input_shape, Phase(), **kw).predict(batch_src)[0]
    allclose_phase(np.angle(out[0]), S)


# ---------------------------------------------------------------- :128-185
@pytest.mark.parametrize('data_format', ['channels_first', 'channels_last'])
@pytest.mark.parametrize('window_name', [None, 'hann_window', 'hamming_window'])
def test_spectrogram_correctness_more(data_format, window_name):
    n_fft, hop_length, n_ch = 512, 256, 2
    src_mono, batch_src, input_shape = get_audio(data_format=data_format, n_ch=n_ch)
    win_length = n_fft
    S_ref = librosa.stft(y=src_mono, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=False,
                         window=window_name.replace('_window', '') if window_name else 'hann').T
    S_ref = np.tile(np.expand_dims(S_ref, axis=2), [1, 1, n_ch])
    if data_format == 'channels_first':
        S_ref = np.transpose(S_ref, (2, 0, 1))
    kw = dict(n_fft=n_fft, win_length=win_length, hop_length=hop_length, window_name=window_name, pad_end=False,
              input_data_format=data_format, output_data_format=data_format)
    S_complex = _stft_model(input_shape, **kw).predict(batch_src)[0]
    allclose_complex_numbers(S_ref, S_complex)
    north_star(S_complex, S_ref)
    S = _stft_model(input_shape, Magnitude(), **kw).predict(batch_src)[0]
    np.testing.assert_allclose(np.abs(S_ref), S, atol=2e-4)
    S = _stft_model(input_shape, Phase(), **kw).predict(batch_src)[0]
    allclose_phase(np.angle(S_complex), S)


# ---------------------------------------------------------------- :188-267
@pytest.mark.parametrize('n_fft', [512])
@pytest.mark.parametrize('sr', [22050])
@pytest.mark.parametrize('hop_length', [None, 256])
@pytest.mark.parametrize('n_ch', [2])
@pytest.mark.parametrize('data_format', ['default', 'channels_first', 'channels_last'])
@pytest.mark.parametrize('amin', [1e-5, 1e-3])
@pytest.mark.parametrize('dynamic_range', [120.0, 80.0])
@pytest.mark.parametrize('n_mels', [40])
@pytest.mark.parametrize('mel_f_min', [0.0])
@pytest.mark.parametrize('mel_f_max', [8000])
def test_melspectrogram_correctness(n_fft, sr, hop_length, n_ch, data_format, amin, dynamic_range, n_mels,
                                    mel_f_min, mel_f_max):
    src_mono, batch_src, input_shape = get_audio(data_format=data_format, n_ch=n_ch)
    win_length = n_fft

    def _get_melgram_model(return_decibel, amin, dynamic_range):
        melgram_model = get_melspectrogram_layer(
            n_fft=n_fft, sample_rate=sr, n_mels=n_mels, mel_f_min=mel_f_min, mel_f_max=mel_f_max,
            win_length=win_length, hop_length=hop_length, input_data_format=data_format,
            output_data_format=data_format, return_decibel=return_decibel, db_amin=amin,
            db_dynamic_range=dynamic_range)
        model = Sequential()
        model.add(Input(shape=input_shape))
        model.add(melgram_model)
        return model

    S_ref = librosa.melspectrogram(y=src_mono, sr=sr, n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                                   center=False, power=1.0, n_mels=n_mels, fmin=mel_f_min, fmax=mel_f_max).T
    S_ref = np.tile(np.expand_dims(S_ref, axis=2), [1, 1, n_ch])
    if data_format == 'channels_first':
        S_ref = np.transpose(S_ref, (2, 0, 1))
    S = _get_melgram_model(return_decibel=False, amin=None, dynamic_range=120.0).predict(batch_src)[0]
    np.testing.assert_allclose(S_ref, S, atol=1e-4)
    north_star(S, S_ref)
    S = _get_melgram_model(return_decibel=True, amin=amin, dynamic_range=dynamic_range).predict(batch_src)[0]
    S_ref_db = librosa.power_to_db(S_ref, ref=1.0, amin=amin, top_db=dynamic_range)
    np.testing.assert_allclose(S_ref_db, S, rtol=3e-3)
    assert np.abs(S_ref_db - S).max() <= 1e-3          # and to a millibel in absolute terms


# ---------------------------------------------------------------- :359-371
@pytest.mark.parametrize('data_format', ['default', 'channels_first', 'channels_last'])
def test_log_spectrogram_runnable(data_format):
    src_mono, batch_src, input_shape = get_audio(data_format=data_format, n_ch=1)
    for db in (True, False):
        layer = get_log_frequency_spectrogram_layer(input_shape, return_decibel=db,
                                                    input_data_format=data_format, output_data_format=data_format)
        out = layer.predict(batch_src)
        assert np.isfinite(out).all() and out.shape[0] == 1 and 84 in out.shape


def test_log_spectrogram_fail():
    """upstream marks this xfail: log_n_bins=200 puts f_max above Nyquist -> RuntimeError (backend.py:266-275)"""
    src_mono, batch_src, input_shape = get_audio(data_format='channels_last', n_ch=1)
    with pytest.raises(RuntimeError):
        get_log_frequency_spectrogram_layer(input_shape, return_decibel=True, log_n_bins=200)


# ---------------------------------------------------------------- :374-387
def test_delta():
    specgrams = np.reshape(np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32), (1, -1, 1, 1))
    delta_model = Sequential()
    delta_model.add(Input(shape=(4, 1, 1)))
    delta_model.add(Delta(win_length=3, data_format='channels_last'))
    delta_kapre = delta_model(specgrams).cpu().numpy()
    delta_ref = np.reshape(np.array([0.5, 1.0, 1.0, 0.5], dtype=np.float32), (1, -1, 1, 1))
    np.testing.assert_allclose(delta_kapre, delta_ref)


# ---------------------------------------------------------------- :390-444
@pytest.mark.parametrize('data_format', ['default', 'channels_first', 'channels_last'])
def test_mag_phase(data_format):
    n_ch = 1
    n_fft, hop_length, win_length = 512, 256, 512
    src_mono, batch_src, input_shape = get_audio(data_format=data_format, n_ch=n_ch)
    mag_phase_layer = get_stft_mag_phase(input_shape=input_shape, n_fft=n_fft, win_length=win_length,
                                         hop_length=hop_length, input_data_format=data_format,
                                         output_data_format=data_format)
    model = Sequential()
    model.add(Input(shape=input_shape))
    model.add(mag_phase_layer)
    mag_phase_kapre = model(batch_src)[0].cpu().numpy()
    ch_axis = 0 if data_format == 'channels_first' else 2
    if data_format == 'default':
        ch_axis = 0 if K.image_data_format() == 'channels_first' else 2
    mag_phase_ref = np.stack(librosa.magphase(librosa.stft(y=src_mono, n_fft=n_fft, hop_length=hop_length,
                                                           win_length=win_length, center=False).T), axis=ch_axis)
    np.testing.assert_equal(mag_phase_kapre.shape, mag_phase_ref.shape)
    np.testing.assert_allclose(np.take(mag_phase_kapre, [0], axis=ch_axis),
                               np.take(np.abs(mag_phase_ref), [0], axis=ch_axis), atol=2e-4)
    # upstream leaves the phase as a todo; the second channel IS the phase here
    allclose_phase(np.take(mag_phase_kapre, [1], axis=ch_axis),
                   np.angle(np.take(mag_phase_ref, [1], axis=ch_axis)), atol=2e-2)


# ---------------------------------------------------------------- :447-534
@pytest.mark.parametrize('waveform_data_format', ['default', 'channels_first', 'channels_last'])
@pytest.mark.parametrize('stft_data_format', ['default', 'channels_first', 'channels_last'])
@pytest.mark.parametrize('hop_ratio', [0.5, 0.25, 0.125])
def test_perfectly_reconstructing_stft_istft(waveform_data_format, stft_data_format, hop_ratio):
    n_ch = 1
    src_mono, batch_src, input_shape = get_audio(data_format=waveform_data_format, n_ch=n_ch)
    _waveform_data_format = K.image_data_format() if waveform_data_format == 'default' else waveform_data_format
    time_axis = 1 if _waveform_data_format == 'channels_first' else 0
    len_src = input_shape[time_axis]
    n_fft = 2048
    hop_length = int(2048 * hop_ratio)
    n_added_frames = int(1 / hop_ratio) - 1
    stft, istft = get_perfectly_reconstructing_stft_istft(n_fft=n_fft, hop_length=hop_length,
                                                          waveform_data_format=waveform_data_format,
                                                          stft_data_format=stft_data_format)
    # [STFT -> ISTFT]
    model = Sequential([Input(shape=input_shape, dtype=batch_src.dtype), stft, istft])
    recon_waveform = model(batch_src).cpu().numpy()
    len_pad_begin = n_fft - hop_length
    if _waveform_data_format == 'channels_first':
        recon_waveform = recon_waveform[:, :, len_pad_begin: len_pad_begin + len_src]
    else:
        recon_waveform = recon_waveform[:, len_pad_begin: len_pad_begin + len_src, :]
    np.testing.assert_allclose(batch_src, recon_waveform, atol=1e-5)

    # [ISTFT -> STFT]: a consistent spectrogram survives the trip
    S = librosa.stft(y=src_mono, n_fft=n_fft, hop_length=hop_length).T.astype(np.complex64)   # centred, as upstream
    _stft_data_format = K.image_data_format() if stft_data_format == 'default' else stft_data_format
    S = np.expand_dims(S, (0, 1)) if _stft_data_format == 'channels_first' else np.expand_dims(S, (0, -1))
    model = Sequential([Input(shape=S.shape[1:], dtype=S.dtype), istft, stft])
    recon_S = model(S).cpu().numpy()
    n = n_added_frames
    first = _stft_data_format == 'channels_first'
    if n > 0:
        S = S[:, :, n:-n, :] if first else S[:, n:-n, :, :]
    n_added_frames += n
    if n_added_frames > 0:
        recon_S = (recon_S[:, :, n_added_frames:-n_added_frames, :] if first
                   else recon_S[:, n_added_frames:-n_added_frames, :, :])
    np.testing.assert_equal(S.shape, recon_S.shape)
    allclose_complex_numbers(S, recon_S)


# ---------------------------------------------------------------- :537-591 + tests/test_signal.py:108-153 (save / load)
def save_load_compare(layer, input_batch, assertion_callback, save_format='tf', layer_class=None, training=None,
                      input_shape=None):
    """tests/utils.py:59-112, line for line with `tf.keras` replaced by kapre_amd.keras_shim: the model is SAVED to a file
    (`.keras` for save_format 'tf', `.h5` for 'h5') and LOADED back (with custom_objects for 'h5', as upstream), and the two
    models' predictions are compared."""
    import os
    import tempfile
    from kapre_amd import keras_shim as keras

    if not isinstance(layer, keras.Model):
        model = keras.Sequential()
        if input_shape is not None:
            model.add(keras.Input(shape=input_shape))
        model.add(layer)
    else:
        model = layer

    if training is None:
        result_original = model.predict(input_batch)
    else:
        result_original = model(input_batch, training=training)

    os_temp_dir = tempfile.gettempdir()
    model_temp_dir = tempfile.TemporaryDirectory(dir=os_temp_dir)

    if save_format == 'tf':
        model_path = os.path.join(model_temp_dir.name, 'model.keras')
    elif save_format == 'h5':
        model_path = os.path.join(model_temp_dir.name, 'model.h5')
    else:
        raise ValueError
    model.save(model_path)

    if save_format == 'h5':
        new_model = keras.load_model(model_path, custom_objects={layer.__class__.__name__: layer_class})
    else:
        new_model = keras.load_model(model_path)

    if training is None:
        result_new = new_model.predict(input_batch)
    else:
        result_new = new_model(input_batch, training=training)

    assertion_callback(result_original, result_new)
    # (beyond upstream: the loaded model has the same layers, configs and static shapes)
    import json
    assert [type(a_) for a_ in model.layers] == [type(b_) for b_ in new_model.layers]
    for a_, b_ in zip(model.layers, new_model.layers):
        assert json.loads(json.dumps(a_.get_config())) == json.loads(json.dumps(b_.get_config()))
    try:
        shape = model.output_shape
    except AttributeError:
        shape = None
    if shape is not None:
        assert new_model.output_shape == shape
        assert tuple(result_new.shape[1:]) == tuple(shape[1:])

    model_temp_dir.cleanup()

    return model


@pytest.mark.parametrize('save_format', ['tf', 'h5'])
def test_save_load(save_format):
    """tests/test_time_frequency.py:537-591 (ConcatenateFrequencyMap is outside SURVEY 8's scope)"""
    src_mono, batch_src, input_shape = get_audio(data_format='channels_last', n_ch=1)
    # test STFT save/load
    save_load_compare(STFT(pad_begin=True), batch_src, allclose_complex_numbers, save_format, STFT, input_shape=input_shape)

    if save_format == 'tf':
        # test melspectrogram save/load
        save_load_compare(get_melspectrogram_layer(input_shape=input_shape, return_decibel=True), batch_src,
                          np.testing.assert_allclose, save_format)
        # test log frequency spectrogram save/load
        save_load_compare(get_log_frequency_spectrogram_layer(input_shape=input_shape, return_decibel=True), batch_src,
                          np.testing.assert_allclose, save_format)
        # test stft_mag_phase  (a functional Model upstream; its stand-in layer here, saved inside a Sequential)
        save_load_compare(get_stft_mag_phase(input_shape=input_shape, return_decibel=True), batch_src,
                          np.testing.assert_allclose, save_format, input_shape=input_shape)
        # test stft mag
        save_load_compare(get_stft_magnitude_layer(input_shape=input_shape), batch_src, np.testing.assert_allclose, save_format)


@pytest.mark.parametrize('data_format', ['default', 'channels_first', 'channels_last'])
@pytest.mark.parametrize('save_format', ['tf', 'h5'])
def test_save_load_signal(data_format, save_format):
    """tests/test_signal.py:108-153 (the mu-law layers are outside SURVEY 8's scope)"""
    from kapre_amd import Frame, Energy, LogmelToMFCC
    src_mono, batch_src, input_shape = get_audio(data_format='channels_last', n_ch=1)
    # test Frame save/load
    save_load_compare(Frame(frame_length=128, hop_length=64, input_shape=input_shape), batch_src, np.testing.assert_allclose,
                      save_format, Frame)
    # test Energy save/load
    save_load_compare(Energy(frame_length=128, hop_length=64, input_shape=input_shape), batch_src, np.testing.assert_allclose,
                      save_format, Energy)
    # test mfcc layer
    expand_dim = (0, 3) if data_format in (_CH_LAST_STR, 'default') else (0, 1)
    # (upstream: librosa.power_to_db(librosa.feature.melspectrogram(y=src_mono).T); the oracle in librosa's place)
    power = np.abs(o.kapre_stft(src_mono.reshape(1, -1, 1), 2048, 2048, 512))[0, :, :, 0] ** 2
    logmel = o.magnitude_to_decibel((power @ o.filterbank_mel(22050, 1025, 128))[None])[0]
    save_load_compare(LogmelToMFCC(n_mfccs=10), np.expand_dims(logmel.astype(np.float32), expand_dim), np.testing.assert_allclose,
                      save_format, LogmelToMFCC)


# ---------------------------------------------------------------- tests/test_backend.py:13-31
@pytest.mark.parametrize('dynamic_range', [80.0, 120.0])
@pytest.mark.parametrize('dtype', ['float16', 'float32', 'float64'])
def test_magnitude_to_decibel(dynamic_range, dtype):
    x = np.array([[1e-20, 1e-5, 1e-3, 5e-2], [0.3, 1.0, 20.5, 9999]], dtype=dtype)
    amin = 1e-5
    got = backend.magnitude_to_decibel(x, ref_value=1.0, amin=amin, dynamic_range=dynamic_range).cpu().numpy()
    want = np.stack([librosa.power_to_db(row, ref=1.0, amin=amin, top_db=dynamic_range) for row in x])
    if dtype == 'float16':
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)
    else:
        np.testing.assert_allclose(got, want, atol=1e-5 if dtype == 'float64' else 2e-5)   # upstream TOL = 1e-5 (fp64 graph)
        np.testing.assert_allclose(got, [[-50.0, -50.0, -30.0, -13.0103], [-5.2288, 0.0, 13.1175, 39.9996]], atol=2e-4)
