"""CPU tests of the Keras-protocol shim's persistence and static shapes (VERDICT r04 missing 3; SURVEY 8b lists `.output_shape`):
`model.save(path)` / `keras_shim.load_model(path, custom_objects=...)` as the reference's tests/utils.py:59-112 uses them, and
`compute_output_shape` / `input_shape` / `output_shape` against the reference's frame-count formulas
(tests/test_time_frequency.py:32-39).  No kernel runs here: building a layer needs no GPU."""
import json
import os
import zipfile

import numpy as np
import pytest

import kapre_amd as kapre
from kapre_amd import keras_shim as keras
from kapre_amd import STFT, InverseSTFT, Magnitude, Phase, MagnitudeToDecibel, ApplyFilterbank, Delta, Frame, Energy, LogmelToMFCC


def _num_frame_valid(nsp_src, nsp_win, len_hop):          # tests/test_time_frequency.py:32-34
    return (nsp_src - (nsp_win - len_hop)) // len_hop


def _num_frame_same(nsp_src, len_hop):                    # :37-39
    return int(np.ceil(float(nsp_src) / len_hop))


@pytest.mark.parametrize("n_fft, hop, win", [(512, 128, 512), (1000, 250, 1000), (2048, 512, 2018), (400, 160, 400)])
@pytest.mark.parametrize("fmt", ["channels_last", "channels_first"])
@pytest.mark.parametrize("pad_end", [False, True])
def test_stft_output_shape_is_the_reference_frame_count(n_fft, hop, win, fmt, pad_end):
    t, ch = 8000, 2
    shape = (t, ch) if fmt == "channels_last" else (ch, t)
    layer = STFT(n_fft=n_fft, win_length=win, hop_length=hop, pad_end=pad_end, input_data_format=fmt, output_data_format=fmt,
                 input_shape=shape)
    frames = _num_frame_same(t, hop) if pad_end else _num_frame_valid(t, win, hop)
    k = n_fft // 2 + 1
    want = (None, frames, k, ch) if fmt == "channels_last" else (None, ch, frames, k)
    assert layer.input_shape == (None,) + shape
    assert layer.output_shape == want
    assert layer.compute_output_shape((7,) + shape) == (7,) + want[1:]
    # unknown time axis stays unknown
    unk = (None, None, ch) if fmt == "channels_last" else (None, ch, None)
    assert layer.compute_output_shape(unk)[1 if fmt == "channels_last" else 2] is None
    # pad_begin: n_fft - hop more samples on the left (the code at time_frequency.py:169-172, not the docstring)
    pb = STFT(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=True, pad_end=pad_end, input_data_format=fmt,
              output_data_format=fmt)
    fpb = pb.compute_output_shape((None,) + shape)[1 if fmt == "channels_last" else 2]
    t2 = t + n_fft - hop
    assert fpb == (_num_frame_same(t2, hop) if pad_end else _num_frame_valid(t2, win, hop))


def test_chain_shapes():
    m = kapre.get_melspectrogram_layer(input_shape=(44100, 1), n_fft=2048, hop_length=512, n_mels=128, return_decibel=True)
    assert m.input_shape == (None, 44100, 1) and m.output_shape == (None, 83, 128, 1)
    lf = kapre.get_log_frequency_spectrogram_layer(input_shape=(2, 22050), n_fft=1024, hop_length=256, input_data_format="channels_first",
                                                   output_data_format="channels_first")
    assert lf.output_shape == (None, 2, 83, 84)
    mp = kapre.get_stft_mag_phase(input_shape=(8000, 2), n_fft=512)
    assert mp.output_shape == (None, 59, 257, 4)                      # magnitude and phase concatenated on the channel axis
    st, ist = kapre.get_perfectly_reconstructing_stft_istft(1024, 256, "channels_last", "channels_last")
    rt = keras.Sequential([keras.Input(shape=(110250, 1)), st, ist])
    assert rt.output_shape == (None, 433 * 256 + 1024, 1)              # untrimmed, as upstream (the caller trims)
    assert InverseSTFT(n_fft=512, hop_length=128, input_data_format="channels_first", output_data_format="channels_last") \
        .compute_output_shape((3, 2, 10, 257)) == (3, 9 * 128 + 512, 2)
    for layer in (Magnitude(), Phase(), MagnitudeToDecibel(), Delta()):
        assert layer.compute_output_shape((None, 83, 1025, 2)) == (None, 83, 1025, 2)
    fb = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=22050, n_freq=257, n_mels=40), data_format="channels_first")
    assert fb.compute_output_shape((None, 2, 59, 257)) == (None, 2, 59, 40)
    assert Frame(128, 64, input_shape=(1000, 1)).output_shape == (None, 14, 128, 1)
    assert Frame(128, 64, pad_end=True, data_format="channels_first").compute_output_shape((2, 3, 1000)) == (2, 3, 16, 128)
    assert Energy(frame_length=128, hop_length=64, input_shape=(1000, 1)).output_shape == (None, 14, 1)
    assert LogmelToMFCC(n_mfccs=13).compute_output_shape((None, 83, 128, 1)) == (None, 83, 13, 1)
    with pytest.raises(AttributeError):
        STFT().output_shape                                            # no input shape known: as a never-built Keras layer


@pytest.mark.parametrize("ext", [".keras", ".h5"])
def test_save_load_files(tmp_path, ext):
    m = kapre.get_melspectrogram_layer(input_shape=(22050, 2), n_fft=1024, hop_length=256, n_mels=64, return_decibel=True, mel_htk=True,
                                       mel_f_max=8000.0, pad_end=True)
    path = os.path.join(str(tmp_path), "model" + ext)
    m.save(path)
    if ext == ".keras":                                                # the archive layout of Keras 3
        assert zipfile.is_zipfile(path)
        with zipfile.ZipFile(path) as z:
            assert {"config.json", "metadata.json"} <= set(z.namelist())
            cfg = json.loads(z.read("config.json").decode())
            assert cfg["class_name"] == "Sequential" and cfg["config"]["layers"][0]["class_name"] == "InputLayer"
    n = keras.load_model(path)
    assert type(n) is keras.Sequential and n.name == m.name
    assert [type(l) for l in n.layers] == [type(l) for l in m.layers]
    for a, b in zip(m.layers, n.layers):
        assert json.loads(json.dumps(a.get_config())) == json.loads(json.dumps(b.get_config()))
    assert n.input_shape == m.input_shape and n.output_shape == m.output_shape == (None, 87, 64, 2)
    np.testing.assert_array_equal(n.layers[2].filterbank, m.layers[2].filterbank)
    with pytest.raises(FileExistsError):
        m.save(path, overwrite=False)


def test_load_model_custom_objects_and_unknown_classes(tmp_path):
    class MySTFT(STFT):                                                # not registered: only custom_objects knows it
        pass

    model = keras.Sequential([keras.Input(shape=(4000, 1)), MySTFT(n_fft=256, hop_length=64)])
    path = os.path.join(str(tmp_path), "m.h5")
    model.save(path)
    with pytest.raises(ValueError, match="Unknown layer"):
        keras.load_model(path)
    loaded = keras.load_model(path, custom_objects={"MySTFT": MySTFT})
    assert type(loaded.layers[0]) is MySTFT and loaded.output_shape == (None, 59, 129, 1)
    # the reference's h5 branch passes {class name: class} for registered layers as well (tests/utils.py:98-101)
    m2 = keras.Sequential([Frame(frame_length=128, hop_length=64, input_shape=(1000, 1))])
    p2 = os.path.join(str(tmp_path), "f.h5")
    m2.save(p2)
    assert type(keras.load_model(p2, custom_objects={"Frame": Frame}).layers[0]) is Frame


def test_mag_phase_model_round_trip(tmp_path):
    a = kapre.get_stft_mag_phase(input_shape=(8000, 2), n_fft=512, hop_length=128, return_decibel=True, output_data_format="channels_first",
                                 input_data_format="channels_last")
    path = os.path.join(str(tmp_path), "mp.keras")
    keras.save_model(a, path)
    b = keras.load_model(path)
    assert type(b) is type(a) and b.ch_axis == 1 and b.output_shape == a.output_shape == (None, 4, 59, 257)
    assert json.loads(json.dumps(a.get_config())) == json.loads(json.dumps(b.get_config()))
