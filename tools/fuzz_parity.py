#!/usr/bin/env python3
"""Random configurations of the path against the float64 oracle (development aid; run on the GPU box):
    python tools/fuzz_parity.py [seconds] [seed]
STFT (complex / magnitude), InverseSTFT, the fused (log-)mel chain and the stand-alone layers over random n_fft / hop / window / padding / channel counts
/ layout pairs / launch sizes (from one frame to well past every dispatch threshold).  Prints every failing configuration."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import kapre_oracle as o
from kapre_amd import (STFT, InverseSTFT, Magnitude, Phase, Sequential, ApplyFilterbank, MagnitudeToDecibel, Frame, Energy, Delta,
                       LogmelToMFCC, composed, _ffi)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FMT = ("channels_last", "channels_first")
NFFT = (256, 512, 1024, 2048, 400, 320, 1000, 480, 4096, 384, 250)
bad = n = 0
t_end = time.time() + budget

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

while time.time() < t_end:
    n += 1
    n_fft = int(rng.choice(NFFT))
    hop = int(rng.choice([n_fft // 4, n_fft // 2, n_fft // 8, max(1, n_fft // 4 - 3), int(rng.integers(1, n_fft + 1))]))
    # (round 6: win_length > n_fft -- frames cut at win_length, cropped to n_fft -- in one draw of five, forward transforms only)
    win = int(rng.choice([n_fft, n_fft, max(2, n_fft - int(rng.integers(0, n_fft // 2))), max(2, n_fft - int(rng.integers(0, n_fft // 2))),
                          n_fft + int(rng.integers(1, n_fft))]))
    ch = int(rng.choice([1, 1, 2, 3, 4, 6]))
    fi, fo = FMT[rng.integers(2)], FMT[rng.integers(2)]
    frames = int(rng.choice([1, 3, 17, 60, 200, 700]))
    batch = int(rng.choice([1, 2, 5, 16, 48]))
    if frames * batch * ch * n_fft > 6e7:            # keeps the float64 oracle in seconds
        batch = max(1, int(6e7 // (frames * ch * n_fft)))
    pad_b, pad_e = bool(rng.integers(2)), bool(rng.integers(2))
    kind = rng.choice(["stft", "istft", "mel", "mel", "layers"])
    cfg = dict(kind=str(kind), n_fft=n_fft, hop=hop, win=win, ch=ch, fi=fi, fo=fo, frames=frames, batch=batch, pad=(pad_b, pad_e))
    try:
        if kind == "layers":                          # the stand-alone layers of the path and its consumers
            fmt = fi
            k = int(rng.choice([129, 257, 513, 1025, 201, 161, 481, 1001, 81]))      # round 6: k_fb_pw takes every (k - 1) % 4 == 0
            rows = int(rng.choice([1, 7, 60, 400, 3000]))
            b2 = max(1, min(batch, int(2e7 // (rows * k * ch))))
            xs = np.abs(rng.standard_normal((b2, rows, k, ch) if fmt == "channels_last" else (b2, ch, rows, k))).astype(np.float32) ** 3
            sr, nm = int(rng.choice([16000, 44100])), int(rng.choice([13, 40, 128]))
            fbl = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=sr, n_freq=k, n_mels=nm), data_format=fmt)
            e = rel(fbl(xs).cpu().numpy(), o.apply_filterbank(xs, o.filterbank_mel(sr, k, nm), fmt))
            got = MagnitudeToDecibel()(xs).cpu().numpy()
            e = max(e, rel(10.0 ** (got / 10.0), 10.0 ** (o.magnitude_to_decibel(xs) / 10.0)))
            w = rng.standard_normal((b2, rows * 37 + 50, ch) if fmt == "channels_last" else (b2, ch, rows * 37 + 50)).astype(np.float32)
            fl_ = int(rng.choice([64, 400, 1000]))
            fh = int(rng.choice([h for h in (32, 160, 333) if h <= fl_]))
            if w.shape[1 if fmt == "channels_last" else 2] >= fl_:
                e = max(e, rel(Frame(fl_, fh, data_format=fmt)(w).cpu().numpy(), o.kapre_frame(w, fl_, fh, data_format=fmt)))
                e = max(e, rel(Energy(frame_length=fl_, hop_length=fh, data_format=fmt)(w).cpu().numpy(),
                               o.kapre_energy(w, frame_length=fl_, hop_length=fh, data_format=fmt)))
            cfg = dict(kind="layers", k=k, rows=rows, batch=b2, ch=ch, fmt=fmt, n_mels=nm, frame=(fl_, fh))
        elif kind == "istft":
            win = min(win, n_fft + n_fft // 2)
            cfg["win"] = win
            if hop > win:
                continue
            k = n_fft // 2 + 1
            shape = (batch, frames, k, ch) if fi == "channels_last" else (batch, ch, frames, k)
            s = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
            kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, input_data_format=fi, output_data_format=fo)
            try:
                want = o.kapre_istft(s, **kw)
            except Exception:
                continue                              # window sums with zeros etc.: the oracle refuses, nothing to compare
            if not np.isfinite(want).all():
                continue
            got = InverseSTFT(**kw)(s).cpu().numpy()
            e = rel(got, want)
        else:
            t = max(win, n_fft) + (frames - 1) * hop - int(rng.integers(0, hop))
            t = max(t, 8)
            if not pad_e and t < max(win, n_fft) and kind == "mel":
                continue                              # no frame at all (tests cover the empty outputs)
            x = rng.standard_normal((batch, t, ch) if fi == "channels_last" else (batch, ch, t)).astype(np.float32)
            if kind == "stft":
                kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=pad_b, pad_end=pad_e, input_data_format=fi,
                          output_data_format=fo)
                want = o.kapre_stft(x, **kw)
                if want.size == 0:
                    continue
                e = max(rel(STFT(**kw)(x).cpu().numpy(), want),
                        rel(Sequential([STFT(**kw), Magnitude()])(x).cpu().numpy(), np.abs(want)))
            else:
                db = bool(rng.integers(2))
                kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_end=pad_e, sample_rate=int(rng.choice([16000, 22050, 44100])),
                          n_mels=int(rng.choice([40, 64, 80, 128])), return_decibel=db, input_data_format=fi, output_data_format=fo)
                want = o.kapre_melspectrogram(x, **kw)
                if want.size == 0:
                    continue
                got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
                if db:                                  # back to linear: a dB error far below the item maximum is not an error
                    got, want = 10.0 ** (got / 10.0), 10.0 ** (want / 10.0)
                e = rel(got, want)
        if not (e <= 2e-4):
            bad += 1
            print("FAIL rel err %.3g" % e, cfg, _ffi.last_launches(), flush=True)
    except Exception as ex:                           # an exception is a finding too
        bad += 1
        print("EXC %r" % (ex,), cfg, flush=True)
print("configurations %d, failures %d" % (n, bad))
