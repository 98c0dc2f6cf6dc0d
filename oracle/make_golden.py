#!/usr/bin/env python3
"""Generate tests/golden/kapre_ref_*.npz by running the REAL reference modules.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    python oracle/make_golden.py

/root/reference/kapre/{backend,time_frequency,composed,signal}.py are imported unmodified; their
`tensorflow` / `librosa` imports resolve to oracle/ref_stubs (numpy stand-ins, see its README).
`kapre/__init__.py` is bypassed (it would pull in tflite/augmentation modules that need far more
of TensorFlow) by registering an empty package object whose __path__ is the reference dir.

Every case stores the inputs, the keyword arguments and the float64 output of the reference's own
layer code, so the -m gpu parity tests and the oracle tests can replay it without /root/reference.
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("KAPRE_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")


def load_reference():
    sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
    sys.path.insert(0, HERE)
    pkg = types.ModuleType("kapre")
    pkg.__path__ = [os.path.join(REF, "kapre")]
    sys.modules["kapre"] = pkg
    backend = importlib.import_module("kapre.backend")
    tfq = importlib.import_module("kapre.time_frequency")
    composed = importlib.import_module("kapre.composed")
    sig = importlib.import_module("kapre.signal")
    return backend, tfq, composed, sig


def audio(n_ch, length, data_format, batch=1, offset=0):
    """Same construction as the reference's tests/utils.py:13-35 (tile over channels, repeat over
    batch), but with per-item/per-channel gains so that layout mistakes cannot cancel out."""
    src = np.load(os.path.join(OUT, "speech_test_file.npz"))["audio_data"].astype(np.float32)
    src = src[offset:offset + length]
    x = np.stack([src * np.float32(1.0 - 0.125 * c) for c in range(n_ch)], axis=1)  # (T, C)
    xb = np.stack([np.roll(x, 37 * b, axis=0) * np.float32(1.0 + 0.25 * b) for b in range(batch)])
    if data_format == "channels_first":
        xb = np.transpose(xb, (0, 2, 1))
    return np.ascontiguousarray(xb.astype(np.float32))


def main():
    backend, tfq, composed, sig = load_reference()
    cases = {}

    def add(name, kind, kwargs, x, y, extra=None):
        cases[name] = dict(kind=kind, kwargs=kwargs, x=x, y=np.asarray(y), extra=extra or {})

    # ---- STFT (reference params: tests/test_time_frequency.py:72-185, 270-337) ----
    stft_cases = [
        ("stft_512_256_cl", dict(n_fft=512, hop_length=256, input_data_format="channels_last",
                                 output_data_format="channels_last"), 2, 3000, 2),
        ("stft_512_256_cf_hamming", dict(n_fft=512, hop_length=256, window_name="hamming_window",
                                         input_data_format="channels_first",
                                         output_data_format="channels_first"), 2, 3000, 1),
        ("stft_512_256_cf_to_cl", dict(n_fft=512, hop_length=256,
                                       input_data_format="channels_first",
                                       output_data_format="channels_last"), 3, 2000, 1),
        ("stft_1000_default_hop", dict(n_fft=1000, input_data_format="channels_last",
                                       output_data_format="channels_last"), 1, 3000, 1),
        ("stft_1000_win512_padend", dict(n_fft=1000, win_length=512, hop_length=256, pad_end=True,
                                         input_data_format="channels_last",
                                         output_data_format="channels_last"), 1, 3000, 1),
        ("stft_2048_512_padbegin_padend", dict(n_fft=2048, hop_length=512, pad_begin=True,
                                               pad_end=True, window_name="hann_window",
                                               input_data_format="channels_last",
                                               output_data_format="channels_first"), 1, 5000, 1),
        ("stft_256_win200_hop80", dict(n_fft=256, win_length=200, hop_length=80, pad_begin=True,
                                       input_data_format="channels_first",
                                       output_data_format="channels_first"), 2, 1500, 2),
        ("stft_512_win511_odd", dict(n_fft=512, win_length=511, hop_length=100,
                                     input_data_format="channels_last",
                                     output_data_format="channels_last"), 1, 2000, 1),
    ]
    for name, kw, n_ch, length, batch in stft_cases:
        x = audio(n_ch, length, kw["input_data_format"], batch)
        layer = tfq.STFT(**kw)
        y = layer(x.astype(np.float64))
        add(name, "stft", kw, x, y, {"config": layer.get_config()})

    # ---- Magnitude / stft_magnitude layer ----
    kw = dict(n_fft=512, hop_length=256, return_decibel=True, db_amin=1e-5, db_dynamic_range=60.0,
              input_data_format="channels_last", output_data_format="channels_first")
    x = audio(2, 3000, "channels_last", 2)
    add("stftmag_512_db", "stft_magnitude", kw, x,
        composed.get_stft_magnitude_layer(**kw)(x.astype(np.float64)))
    kw = dict(n_fft=1024, hop_length=160, win_length=400, pad_end=True,
              input_data_format="channels_first", output_data_format="channels_last")
    x = audio(1, 4000, "channels_first", 1)
    add("stftmag_1024_win400", "stft_magnitude", kw, x,
        composed.get_stft_magnitude_layer(**kw)(x.astype(np.float64)))

    # ---- melspectrogram (reference params: tests/test_time_frequency.py:188-267) ----
    mel_cases = [
        ("mel_512_h128_cl", dict(n_fft=512, sample_rate=22050, n_mels=40, mel_f_min=0.0,
                                 mel_f_max=8000, hop_length=None, input_data_format="channels_last",
                                 output_data_format="channels_last"), 2, 3000, 1),
        ("mel_512_h256_cf_db", dict(n_fft=512, sample_rate=22050, n_mels=40, mel_f_min=0.0,
                                    mel_f_max=8000, hop_length=256, return_decibel=True,
                                    db_amin=1e-5, db_dynamic_range=80.0,
                                    input_data_format="channels_first",
                                    output_data_format="channels_first"), 2, 3000, 2),
        ("mel_512_h256_cl_db_amin1e-3_dr30", dict(n_fft=512, sample_rate=22050, n_mels=40,
                                                 mel_f_min=0.0, mel_f_max=8000, hop_length=256,
                                                 return_decibel=True, db_amin=1e-3,
                                                 db_dynamic_range=30.0,
                                                 input_data_format="channels_last",
                                                 output_data_format="channels_last"), 2, 3000, 2),
        ("mel_2048_512_128", dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128,
                                  input_data_format="channels_last",
                                  output_data_format="channels_last"), 1, 6000, 2),
        ("mel_2048_1024_128_db_cf6", dict(n_fft=2048, hop_length=1024, sample_rate=44100,
                                          n_mels=128, return_decibel=True,
                                          input_data_format="channels_first",
                                          output_data_format="channels_first"), 6, 5000, 1),
        ("mel_1024_160_80_htk", dict(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=80,
                                     mel_htk=True, mel_norm=None, return_decibel=True,
                                     db_ref_value=0.5, input_data_format="channels_last",
                                     output_data_format="channels_first"), 1, 4000, 1),
    ]
    for name, kw, n_ch, length, batch in mel_cases:
        x = audio(n_ch, length, kw["input_data_format"], batch)
        model = composed.get_melspectrogram_layer(**kw)
        y = model(x.astype(np.float64))
        cfgs = [type(l).__name__ for l in model.layers]
        add(name, "melspectrogram", kw, x, y, {"layers": cfgs})

    # ---- ISTFT + perfect reconstruction (tests/test_time_frequency.py:447-534) ----
    for hop in (1024, 512, 256):
        for wf, sf in (("channels_last", "channels_last"), ("channels_first", "channels_last"),
                       ("channels_last", "channels_first")):
            if hop != 512 and (wf, sf) != ("channels_last", "channels_last"):
                continue
            stft, istft = composed.get_perfectly_reconstructing_stft_istft(
                n_fft=2048, hop_length=hop, waveform_data_format=wf, stft_data_format=sf)
            x = audio(1, 5000, wf, 2)
            s = stft(x.astype(np.float64))
            y = istft(s)
            kw = dict(n_fft=2048, hop_length=hop, waveform_data_format=wf, stft_data_format=sf)
            add("roundtrip_2048_%d_%s_%s" % (hop, wf[9:], sf[9:]), "roundtrip", kw, x, y,
                {"istft_config": istft.get_config()})
    kw = dict(n_fft=512, win_length=400, hop_length=100, forward_window_name="hamming_window",
              input_data_format="channels_first", output_data_format="channels_last")
    rng = np.random.default_rng(7)
    s = (rng.standard_normal((2, 2, 9, 257)) + 1j * rng.standard_normal((2, 2, 9, 257)))
    s = s.astype(np.complex64)
    add("istft_512_win400_hop100", "istft", kw, s, tfq.InverseSTFT(**kw)(s.astype(np.complex128)))

    # ---- ApplyFilterbank standalone, both layouts ----
    fbk = dict(sample_rate=22050, n_freq=257, n_mels=32, f_min=200.0, f_max=8000.0)
    for fmt in ("channels_last", "channels_first"):
        shp = (2, 7, 257, 3) if fmt == "channels_last" else (2, 3, 7, 257)
        x = rng.uniform(0, 3, shp).astype(np.float32)
        layer = tfq.ApplyFilterbank(type="mel", filterbank_kwargs=fbk, data_format=fmt)
        add("applyfb_mel_%s" % fmt[9:], "apply_filterbank",
            dict(type="mel", filterbank_kwargs=fbk, data_format=fmt), x, layer(x.astype(np.float64)))
    fbk = dict(sample_rate=22050, n_freq=257, n_bins=48, bins_per_octave=12)
    x = rng.uniform(0, 3, (1, 5, 257, 2)).astype(np.float32)
    layer = tfq.ApplyFilterbank(type="log", filterbank_kwargs=fbk, data_format="channels_last")
    add("applyfb_log_last", "apply_filterbank",
        dict(type="log", filterbank_kwargs=fbk, data_format="channels_last"), x,
        layer(x.astype(np.float64)))

    # ---- magnitude_to_decibel known-answer input (tests/test_backend.py:20-22) ----
    x = np.array([[1e-20, 1e-5, 1e-3, 5e-2], [0.3, 1.0, 20.5, 9999]], dtype=np.float32)
    for dr in (80.0, 120.0, 20.0):
        add("db_known_dr%d" % dr, "magnitude_to_decibel",
            dict(ref_value=1.0, amin=1e-5, dynamic_range=dr), x,
            backend.magnitude_to_decibel(x.astype(np.float64), ref_value=1.0, amin=1e-5,
                                         dynamic_range=dr))
    x = rng.uniform(0, 2, (3, 4, 5, 2)).astype(np.float32) ** 8
    add("db_4d", "magnitude_to_decibel", dict(ref_value=2.0, amin=1e-4, dynamic_range=25.0), x,
        backend.magnitude_to_decibel(x.astype(np.float64), ref_value=2.0, amin=1e-4,
                                     dynamic_range=25.0))
    x1 = rng.uniform(0, 2, (50,)).astype(np.float32) ** 6
    add("db_1d", "magnitude_to_decibel", dict(ref_value=1.0, amin=1e-5, dynamic_range=15.0), x1,
        backend.magnitude_to_decibel(x1.astype(np.float64), ref_value=1.0, amin=1e-5,
                                     dynamic_range=15.0))

    # ---- filterbank_mel through the reference's wrapper (tests/test_backend.py:43-75 params) ----
    for i, kw in enumerate([
        dict(sample_rate=22050, n_freq=257, n_mels=32, f_min=0.0, f_max=11025, htk=False,
             norm="slaney"),
        dict(sample_rate=44100, n_freq=257, n_mels=32, f_min=200, f_max=11025, htk=True,
             norm=None),
        dict(sample_rate=22050, n_freq=257, n_mels=32, f_min=200, f_max=5512, htk=False, norm=1.0),
    ]):
        fb = backend.filterbank_mel(**kw)
        assert fb.dtype == np.float32 and fb.shape == (257, 32)
        add("fbmel_%d" % i, "filterbank_mel", kw, np.zeros(0, np.float32), fb)

    # ---- consumers of the path: Frame / Energy / Delta / LogmelToMFCC ----
    # (reference params: tests/test_signal.py:11-106, tests/test_time_frequency.py:375-387)
    for i, (kw, n_ch, length, batch) in enumerate([
        (dict(frame_length=50, hop_length=25, pad_end=False, data_format="channels_last"), 1, 1000, 2),
        (dict(frame_length=32, hop_length=16, pad_end=False, data_format="channels_first"), 2, 1000, 1),
        (dict(frame_length=400, hop_length=160, pad_end=True, pad_value=0.25, data_format="channels_last"), 3, 2100, 2),
        (dict(frame_length=64, hop_length=64, pad_end=True, data_format="channels_first"), 1, 500, 1),
    ]):
        x = audio(n_ch, length, kw["data_format"], batch)
        layer = sig.Frame(**kw)
        add("frame_%d" % i, "frame", kw, x, layer(x.astype(np.float64)), {"config": layer.get_config()})
    for i, (kw, n_ch, length, batch) in enumerate([
        (dict(sample_rate=22050, ref_duration=0.1, frame_length=4, hop_length=2, data_format="channels_last"), 1, 8, 1),
        (dict(frame_length=2205, hop_length=1102, pad_end=True, data_format="channels_first"), 2, 9000, 2),
        (dict(sample_rate=16000, ref_duration=0.05, frame_length=400, hop_length=160, data_format="channels_last"), 2, 4000, 2),
    ]):
        x = audio(n_ch, length, kw["data_format"], batch)
        layer = sig.Energy(**kw)
        add("energy_%d" % i, "energy", kw, x, layer(x.astype(np.float64)), {"config": layer.get_config()})
    x = np.reshape(np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32), (1, -1, 1, 1))
    add("delta_known", "delta", dict(win_length=3, data_format="channels_last"), x,
        tfq.Delta(win_length=3, data_format="channels_last")(x.astype(np.float64)))
    for i, (kw, shp) in enumerate([
        (dict(win_length=5, mode="symmetric", data_format="channels_last"), (2, 9, 7, 1)),
        (dict(win_length=9, mode="reflect", data_format="channels_last"), (1, 20, 5, 1)),
        (dict(win_length=7, mode="constant", data_format="channels_first"), (2, 1, 12, 6)),
        (dict(win_length=3, mode="SYMMETRIC", data_format="channels_first"), (1, 1, 2, 3)),
    ]):
        x = rng.standard_normal(shp).astype(np.float32)
        layer = tfq.Delta(**kw)
        add("delta_%d" % i, "delta", kw, x, layer(x.astype(np.float64)), {"config": layer.get_config()})
    for i, (kw, shp) in enumerate([
        (dict(n_mfccs=20, data_format="channels_last"), (2, 11, 128, 1)),
        (dict(n_mfccs=1, data_format="channels_last"), (1, 5, 40, 2)),
        (dict(n_mfccs=40, data_format="channels_first"), (2, 2, 9, 40)),
        (dict(n_mfccs=13, data_format="channels_first"), (1, 3, 4, 80)),
    ]):
        x = (rng.standard_normal(shp) * 20 - 30).astype(np.float32)
        layer = sig.LogmelToMFCC(**kw)
        add("mfcc_%d" % i, "logmel_to_mfcc", kw, x, layer(x.astype(np.float64)), {"config": layer.get_config()})

    # ---- error behaviour recorded as (exception type name) ----
    errors = {}
    for label, fn in {
        "bad_data_format_value": lambda: tfq.STFT(input_data_format="weird"),
        "bad_data_format_type": lambda: tfq.STFT(output_data_format=3),
        "bad_window": lambda: tfq.STFT(window_name="bartlett"),
        "bad_db_ref": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), ref_value=0.0),
        "bad_db_amin": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), amin=-1.0),
        "bad_db_dr": lambda: backend.magnitude_to_decibel(np.ones((2, 2)), dynamic_range=0.0),
        "bad_log_fmax": lambda: backend.filterbank_log(sample_rate=8000, n_freq=257, n_bins=120),
        "delta_win_small": lambda: tfq.Delta(win_length=1),
        "delta_win_even": lambda: tfq.Delta(win_length=4),
        "delta_bad_mode": lambda: tfq.Delta(mode="wrap"),
        "frame_len_zero": lambda: sig.Frame(frame_length=0, hop_length=1),
        "frame_hop_zero": lambda: sig.Frame(frame_length=4, hop_length=0),
        "frame_hop_gt_len": lambda: sig.Frame(frame_length=4, hop_length=8),
        "frame_bad_format": lambda: sig.Frame(frame_length=4, hop_length=2, data_format="nope"),
    }.items():
        try:
            fn()
            errors[label] = None
        except Exception as e:  # noqa: BLE001
            errors[label] = type(e).__name__

    os.makedirs(OUT, exist_ok=True)
    arrays, meta = {}, {}
    for name, c in cases.items():
        arrays[name + "__x"] = c["x"]
        arrays[name + "__y"] = c["y"]
        meta[name] = dict(kind=c["kind"], kwargs=c["kwargs"], extra=c["extra"])
    np.savez_compressed(os.path.join(OUT, "kapre_ref_cases.npz"), **arrays)
    with open(os.path.join(OUT, "kapre_ref_cases.json"), "w") as f:
        json.dump(dict(cases=meta, errors=errors,
                       provenance="generated by oracle/make_golden.py from /root/reference "
                                  "(kapre 0.4.0) on numpy stand-ins for tensorflow/librosa"),
                  f, indent=1, sort_keys=True, default=str)
    sz = os.path.getsize(os.path.join(OUT, "kapre_ref_cases.npz"))
    print("wrote %d cases, %.1f KiB; errors=%s" % (len(cases), sz / 1024, errors))


if __name__ == "__main__":
    main()
